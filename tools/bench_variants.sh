mkdir -p gpurun_out
export ISINGDIFF_ALLOW_EXPERIMENT=1   # variants built with IDM_EXP_* switches refuse to create handles without it
for lib in diffusion-model_amd/isingdiff/libisingdiff.so diffusion-model_amd/isingdiff/abl/lib_*.so diffusion-model_amd/isingdiff/libisingdiff.so; do
  ISINGDIFF_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --batch ${BV_BATCH:-8192} --T 50 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bv.json 2> gpurun_out/bv.err || { tail -5 gpurun_out/bv.err; exit 1; }
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bv.json"))
print(f"{sys.argv[1].split('/')[-1]:28s} {d['value']:9.1f} E={d['observables']['mean_E_per_site']:.6f} " + " ".join(f"{k['name'].split('(')[0][6:]}={k['avg_us']:.1f}" for k in d["kernels"]))
PY
done
