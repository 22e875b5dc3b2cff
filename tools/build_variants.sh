#!/bin/bash
# Builds tuning variants of the fused kernels: each argument is "name:-DIDM_CFG_X=... [-D...]".
# The libraries land in diffusion-model_amd/isingdiff/abl/lib_<name>.so (git-ignored).
set -e
cd "$(dirname "$0")/../diffusion-model_amd/csrc"
mkdir -p ../isingdiff/abl
find ../isingdiff/abl -name "lib_*.so" ! -name "lib_0base.so" -delete
n=0
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  n=$((n + 1)); [ $((n % 4)) -eq 0 ] && wait
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c idm_fused.hip -o /tmp/var_$name.o &&
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c idm_fused_split.hip -o /tmp/var_${name}_split.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../isingdiff/abl/lib_$name.so idm_api.o idm_canon.o idm_fast.o idm_strip.o idm_tileconv.o idm_bitflip.o idm_mlp.o idm_fft.o idm_comm.o idm_xy.o /tmp/var_$name.o /tmp/var_${name}_split.o || echo "FAILED $name" ) &
done
wait
ls ../isingdiff/abl
