#!/bin/bash
# Short GPU check used while tuning kernels: bf16 parity tests, then a B=4096, T=50 bench line.
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread -k "bf16 or shard" > gpurun_out/t1.log 2>&1
rc=$?
tail -1 gpurun_out/t1.log | cut -c1-300
[ $rc -eq 0 ] || { tail -30 gpurun_out/t1.log | cut -c1-400; exit 1; }
timeout -k 10 300 python -u bench.py --batch 4096 --T 50 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_small.json 2> gpurun_out/bench_small.err || { tail -5 gpurun_out/bench_small.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_small.json"))
print(round(d["value"], 1), round(d["whole_job_mfma_frac"], 4), [(k["name"].split("(")[0], round(k["avg_us"], 1)) for k in d["kernels"]])
PY
