#!/bin/bash
# Same-box A/B of two settings of an environment switch of the library (read once per process), interleaved:
#   tools/ab_env.sh VAR valueA valueB [rounds] [bench.py arguments...]
# Each round runs bench.py once per value (short schedule by default); prints the value and the per-kernel averages.
var=$1; a=$2; b=$3; rounds=${4:-3}; shift 4
args=${@:---batch 16384 --T 50 --steps 1 --warmup 1 --no-cpu-baseline --fp32-lattices 0}
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for v in "$a" "$b"; do
    env $var=$v timeout -k 10 200 python -u bench.py $args > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python - "$var=$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
print(f"{sys.argv[1]:28s} {d['value']:9.1f} E={d['observables']['mean_E_per_site']:.6f} " + " ".join(f"{k['name'].split('(')[0][6:]}={k['avg_us']:.1f}" for k in d["kernels"]))
PY
  done
done
