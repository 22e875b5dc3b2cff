#!/bin/bash
# Full GPU suite + short bench (tuning aid).
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/gpu_all.log 2>&1; rc=$?
tail -3 gpurun_out/r2/gpu_all.log; [ $rc -eq 0 ] || { tail -60 gpurun_out/r2/gpu_all.log; exit 1; }
timeout -k 10 300 python -u bench.py --batch 16384 --T 50 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2/bench_small.json 2> gpurun_out/r2/bench_small.err || { tail -5 gpurun_out/r2/bench_small.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2/bench_small.json"))
print(round(d["value"], 1), round(d["whole_job_mfma_frac"], 4), [(k["name"].split("(")[0], round(k["avg_us"], 1), round(k.get("tflops", 0))) for k in d["kernels"]])
PY
