#!/bin/bash
# Records the micro-benchmark, the phase timelines (two / one workgroup per CU, packed / plain epilogue) and the relative
# phase of co-resident workgroups under gpurun_out/r2/.
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 5 60 ./build/ubench_overlap > gpurun_out/r2/ubench_overlap.txt || exit 1
for v in TIMELINE TIMELINE1 TLepisc; do
  TL_B=16384 ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/abl/lib_$v.so timeout -k 10 200 python -u tools/timeline.py 2>/dev/null > gpurun_out/r2/timeline_$v.txt || exit 1
done
ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/abl/lib_TIMELINE.so timeout -k 10 200 python -u tools/timeline2.py 2>/dev/null > gpurun_out/r2/timeline2.txt || exit 1
tail -3 gpurun_out/r2/timeline2.txt
