#!/bin/bash
# Phase timelines of the fused kernels: two workgroups per CU (product) and one (lone wavefront per SIMD).
set -o pipefail
mkdir -p gpurun_out/r2
for v in TIMELINE TIMELINE1; do
  echo "== $v"
  TL_B=16384 ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/abl/lib_$v.so timeout -k 10 200 python -u tools/timeline.py > gpurun_out/r2/timeline_$v.txt 2>&1 || { tail gpurun_out/r2/timeline_$v.txt; exit 1; }
  cat gpurun_out/r2/timeline_$v.txt
done
