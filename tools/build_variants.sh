#!/bin/bash
# Builds tuning variants of the fused and strip kernels: each argument is "name:-DIDM_CFG_X=... [-D...]".
# The libraries land in diffusion-model_amd/isingdiff/abl/lib_<name>.so (git-ignored).
set -e
cd "$(dirname "$0")/../diffusion-model_amd/csrc"
mkdir -p ../isingdiff/abl
find ../isingdiff/abl -name "lib_*.so" ! -name "lib_0base.so" -delete
n=0
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  n=$((n + 1)); [ $((n % 4)) -eq 0 ] && wait
  api=idm_api.o
  case "$flags" in *IDM_EXP_*|*IDM_SILU_SCALAR*)
    # timing experiments (wrong results): the whole variant is marked, and its idm_model_create refuses to run
    # without ISINGDIFF_ALLOW_EXPERIMENT=1
    flags="$flags -DIDM_EXPERIMENT"; api=/tmp/var_${name}_api.o
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DIDM_EXPERIMENT -c idm_api.hip -o $api ;;
  esac
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c idm_fused.hip -o /tmp/var_$name.o &&
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c idm_fused_split.hip -o /tmp/var_${name}_split.o &&
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c idm_strip.hip -o /tmp/var_${name}_strip.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../isingdiff/abl/lib_$name.so $api idm_canon.o idm_fast.o /tmp/var_${name}_strip.o idm_tileconv.o idm_bitflip.o idm_mlp.o idm_fft.o idm_comm.o idm_xy.o /tmp/var_$name.o /tmp/var_${name}_split.o || echo "FAILED $name" ) &
done
wait
ls ../isingdiff/abl
