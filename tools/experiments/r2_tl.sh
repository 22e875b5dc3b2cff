#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
TL_B=16384 ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/abl/lib_TIMELINE.so timeout -k 10 200 python -u tools/timeline.py > gpurun_out/r2/timeline.txt 2>&1 || { tail gpurun_out/r2/timeline.txt; exit 1; }
cat gpurun_out/r2/timeline.txt
