#!/bin/bash
# Experiment: fused_B with 16-row bands at one workgroup per CU (libraries built with -DIDM_EXP_B16).
mkdir -p gpurun_out
run() { timeout -k 10 200 python -u bench.py --batch 4096 --T 50 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/tb.json 2> gpurun_out/tb.err || { tail -5 gpurun_out/tb.err; exit 1; }
  python - "$1" <<'PY'
import json, sys
d = json.load(open("gpurun_out/tb.json"))
print(f"{sys.argv[1]:22s} {d['value']:9.1f} E={d['observables']['mean_E_per_site']:.6f} " + " ".join(f"{k['name'].split('(')[0][6:]}={k['avg_us']:.1f}" for k in d["kernels"]))
PY
}
export ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/libisingdiff.so; run base
export IDM_EXP_B_ROWS=16
for v in b16a b16b b16c b16d; do export ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/abl/lib_$v.so; run $v; done
unset IDM_EXP_B_ROWS
export ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/libisingdiff.so; run base
