#!/bin/bash
# Samples GPU clock and power while the short bench runs (is the kernel mix power-limited?).
mkdir -p gpurun_out
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_watch.log &
W=$!
timeout -k 10 200 python -u bench.py --batch 8192 --T 100 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bv.json 2> gpurun_out/bv.err
wait $W
cut -c1-250 gpurun_out/clock_watch.log | uniq -c | head -40
