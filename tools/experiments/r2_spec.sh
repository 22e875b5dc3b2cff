#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 600 python -u -m pytest tests/test_gpu_spectral.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/spec_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r2/spec_tests.log; [ $rc -eq 0 ] || { tail -50 gpurun_out/r2/spec_tests.log; exit 1; }
timeout -k 10 200 python3 tools/bench_spectral.py 4096 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/bench_spectral.py 16384 2>/dev/null || exit 1
