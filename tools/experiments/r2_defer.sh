#!/bin/bash
# fused_B with deferred epilogues against the undeferred copy-wavefront kernel: bit identity, then a short bench of each.
set -o pipefail
mkdir -p gpurun_out/defer
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/defer/parity.log 2>&1 || { tail -30 gpurun_out/defer/parity.log; exit 1; }
tail -2 gpurun_out/defer/parity.log
for d in 0 1 0 1; do
  env ${DEFER_VAR:-ISINGDIFF_B_DEFER}=$d timeout -k 10 300 python3 -u bench.py --batch 16384 --T 50 --steps 1 --warmup 1 --no-cpu-baseline --fp32-lattices 0 > gpurun_out/defer/bench_$d.json 2> gpurun_out/defer/bench_$d.err || { tail -5 gpurun_out/defer/bench_$d.err; exit 1; }
  python3 - $d <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/defer/bench_{sys.argv[1]}.json"))
print("defer", sys.argv[1], round(d["value"], 1), [(k["name"].split("(")[0], round(k["avg_us"], 1)) for k in d["kernels"]])
PY
done
