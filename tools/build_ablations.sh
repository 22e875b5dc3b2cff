#!/bin/bash
# Builds timing-only variants of the fused kernels (wrong results on purpose) for ablation.
set -e
cd "$(dirname "$0")/../diffusion-model_amd/csrc"
mkdir -p ../isingdiff/abl
for v in NOMFMA NOSTAGE NOOUT "NOMFMA -DIDM_ABL_NOSTAGE -DIDM_ABL_NOOUT -DIDM_ABL_NOEPI" "NOW -DIDM_ABL_NOB" "NOW -DIDM_ABL_NOB -DIDM_ABL_NOEPI"; do
  name=$(echo "$v" | sed 's/ -DIDM_ABL_/_/g')
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DIDM_ABL_$v -c idm_fused.hip -o /tmp/abl_fused.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../isingdiff/abl/lib_$name.so idm_api.o idm_canon.o idm_fast.o /tmp/abl_fused.o
done
ls -la ../isingdiff/abl
