#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
for v in ${TL_VARIANTS:-TIMELINE}; do
  echo "== $v"
  TL_VERBOSE=1 TL_B=16384 ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/abl/lib_$v.so timeout -k 10 200 python -u tools/timeline.py > gpurun_out/r2/timeline_$v.txt 2>&1 || { tail gpurun_out/r2/timeline_$v.txt; exit 1; }
  grep -A3 "^${TL_KERNEL:-B} wave0" gpurun_out/r2/timeline_$v.txt | cut -c1-200
  grep -A14 "^${TL_KERNEL:-B} wave0" gpurun_out/r2/timeline_$v.txt | grep -E "epilogue pieces|conv2:" | cut -c1-200
done
