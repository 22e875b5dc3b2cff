#!/bin/bash
# fused_B with copy wavefronts (ISINGDIFF_B_SPLIT=1) against the two-workgroup schedule: bitwise comparison, then timing.
set -o pipefail
P=diffusion-model_amd/isingdiff/libisingdiff.so
for b in ${CMP_BATCHES:-1 5 777 4096}; do
  timeout -k 10 200 python -u tools/experiments/compare_libs_batch.py $P $P $b 3 ISINGDIFF_B_SPLIT=1 2>/dev/null || exit 1
done
for v in 0 1 0 1; do
  ISINGDIFF_B_SPLIT=$v timeout -k 10 200 python -u bench.py --batch 8192 --T 50 --steps 1 --warmup 1 --no-cpu-baseline --fp32-lattices 0 > gpurun_out/bv.json 2> gpurun_out/bv.err || { tail -5 gpurun_out/bv.err; exit 1; }
  python - $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/bv.json"))
print(f"B_SPLIT={sys.argv[1]} {d['value']:9.1f} E={d['observables']['mean_E_per_site']:.6f} " + " ".join(f"{k['name'].split('(')[0][6:]}={k['avg_us']:.1f}" for k in d["kernels"]))
PY
done
