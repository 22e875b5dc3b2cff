#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 600 python -u -m pytest tests/test_xy.py -m gpu -x -q -s --timeout 300 --timeout-method thread > gpurun_out/r2/xy.log 2>&1; rc=$?
tail -25 gpurun_out/r2/xy.log; exit $rc
