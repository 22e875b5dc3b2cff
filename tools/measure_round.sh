#!/bin/bash
# Full measurement on one GPU box: default bench, the same command under rocprofv3 --kernel-trace --stats,
# and the two HBM counter passes (FETCH_SIZE, WRITE_SIZE) on a short slice of the schedule.
# Outputs under gpurun_out/measure/.
set -o pipefail
out=gpurun_out/measure; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 -u bench.py > $out/bench.json 2> $out/bench.err || exit 1
echo "bench done"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o s -- python3 bench.py --no-cpu-baseline --fp32-lattices 0 > $out/bench_under_rocprof.json 2> $out/rocprof.err || exit 1
echo "trace done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o s --output-format csv -- python3 bench.py --batch 16384 --T 20 --no-cpu-baseline --no-profile --fp32-lattices 0 > $out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o s --output-format csv -- python3 bench.py --batch 16384 --T 20 --no-cpu-baseline --no-profile --fp32-lattices 0 > $out/pmc_write.log 2>&1 || exit 1
echo "pmc done"; ls $out $out/trace
