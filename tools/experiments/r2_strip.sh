#!/bin/bash
# Parity tests that cover the strip kernels, then the L=64 / L=128 bench slices for each library in abl/ and the product one.
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_spectral.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/strip_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r2/strip_tests.log; [ $rc -eq 0 ] || { tail -40 gpurun_out/r2/strip_tests.log; exit 1; }
for lib in diffusion-model_amd/isingdiff/abl/lib_*.so diffusion-model_amd/isingdiff/libisingdiff.so; do
  [ -f $lib ] || continue
  for cfg in "64 8192" "128 2048"; do
    set -- $cfg
    ISINGDIFF_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --L $1 --batch $2 --T 20 --steps 1 --warmup 1 --no-cpu-baseline --fp32-lattices 0 > gpurun_out/bv.json 2> gpurun_out/bv.err || { tail -5 gpurun_out/bv.err; exit 1; }
    python - "$lib" $1 <<'PY'
import json, sys
d = json.load(open("gpurun_out/bv.json"))
print(f"{sys.argv[1].split('/')[-1]:22s} L={sys.argv[2]:4s} {d['value']:9.1f} {d['whole_job_mfma_frac']:.4f} E={d['observables']['mean_E_per_site']:.6f} " + " ".join(f"{k['name'].split('(')[0].replace('strip_conv','c').replace('fused_','')}={k['avg_us']:.0f}" for k in d["kernels"]))
PY
  done
done
