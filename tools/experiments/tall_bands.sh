#!/bin/bash
# Experiment: fused kernels with whole-map / 16-row bands at one workgroup per CU (library built with taller
# prefetch registers) against the 8-row bands at two workgroups per CU.
mkdir -p gpurun_out
run() { timeout -k 10 200 python -u bench.py --batch 4096 --T 50 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/tb.json 2> gpurun_out/tb.err || { tail -5 gpurun_out/tb.err; exit 1; }
  python - "$1" <<'PY'
import json, sys
d = json.load(open("gpurun_out/tb.json"))
print(f"{sys.argv[1]:22s} {d['value']:9.1f} E={d['observables']['mean_E_per_site']:.6f} " + " ".join(f"{k['name'].split('(')[0][6:]}={k['avg_us']:.1f}" for k in d["kernels"]))
PY
}
export ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/libisingdiff.so; run base
export ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/abl/lib_tall.so
run tall_lib_8row
export IDM_EXP_BAND_LIMIT_KB=159
for k in 8 16 24; do export IDM_EXP_BAND_KINDS=$k; run tall_kinds_$k; done
unset IDM_EXP_BAND_LIMIT_KB IDM_EXP_BAND_KINDS
export ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/libisingdiff.so; run base
