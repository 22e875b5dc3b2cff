#!/bin/bash
# Same-box A/B of two builds of the library, interleaved:  tools/ab_lib.sh libA.so libB.so [rounds] [bench.py arguments...]
# (paths relative to the repository root).  Each round runs bench.py once per library; prints the value and the per-kernel
# averages.  Libraries built with IDM_EXP_* switches need the allowance below.
la=$1; lb=$2; rounds=${3:-3}; shift 3
args=${@:---batch 16384 --T 50 --steps 1 --warmup 1 --no-cpu-baseline --fp32-lattices 0}
export ISINGDIFF_ALLOW_EXPERIMENT=1
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for lib in "$la" "$lb"; do
    ISINGDIFF_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py $args > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python - "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
print(f"{sys.argv[1].split('/')[-1]:24s} {d['value']:9.1f} " + " ".join(f"{k['name'].replace('fused_','').replace('strip_','').split('(')[0]}={k['avg_us']:.1f}" for k in d["kernels"]))
PY
  done
done
