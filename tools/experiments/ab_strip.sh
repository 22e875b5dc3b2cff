#!/bin/bash
# A/B of two libraries on the large-lattice slices and the spectral sampler
for lib in diffusion-model_amd/isingdiff/libisingdiff.so diffusion-model_amd/isingdiff/abl/lib_0prev.so diffusion-model_amd/isingdiff/libisingdiff.so; do
  for cfg in "64 8192" "128 2048"; do
    set -- $cfg
    ISINGDIFF_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --L $1 --batch $2 --T 20 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python - "$lib" $1 <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
print(f"{sys.argv[1].split('/')[-1]:24s} L={sys.argv[2]:4s} {d['value']:9.1f} frac={d['whole_job_mfma_frac']:.4f} " + " ".join(f"{k['name'].split('(')[0]}={k['avg_us']:.0f}" for k in d["kernels"]))
PY
  done
  ISINGDIFF_LIB=$PWD/$lib timeout -k 10 200 python -u tools/bench_spectral.py 2>&1 | tail -3
done
