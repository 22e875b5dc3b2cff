#!/bin/bash
# Parity tests of the bf16 sampler, then the variant bench.
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/quick_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r2/quick_tests.log; [ $rc -eq 0 ] || { tail -40 gpurun_out/r2/quick_tests.log; exit 1; }
bash tools/bench_variants.sh
