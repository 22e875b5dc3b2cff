#!/bin/bash
# One-call GPU check: allocation-free entry point, FFT kernel variants, full GPU suite, short bench.
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2/noalloc.log 2>&1; rc=$?
tail -3 gpurun_out/r2/noalloc.log; [ $rc -eq 0 ] || { tail -40 gpurun_out/r2/noalloc.log; exit 1; }
timeout -k 10 300 python -u -m pytest tests/test_gpu_spectral.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r2/spectral.log 2>&1; rc=$?
tail -3 gpurun_out/r2/spectral.log; [ $rc -eq 0 ] || { tail -40 gpurun_out/r2/spectral.log; exit 1; }
for v in 0 1 2 3 4; do
  IDM_FFT32=$v timeout -k 10 120 python -u tools/bench_elementwise.py > gpurun_out/r2/elem_$v.json 2> gpurun_out/r2/elem_$v.err || { tail -5 gpurun_out/r2/elem_$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r2/elem_{sys.argv[1]}.json"))
print("fft variant", sys.argv[1], d["kernels"]["irfft2_sign (spectrum in, int8 out)"])
PY
done
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/gpu_all.log 2>&1; rc=$?
tail -3 gpurun_out/r2/gpu_all.log; [ $rc -eq 0 ] || { tail -40 gpurun_out/r2/gpu_all.log; exit 1; }
timeout -k 10 300 python -u bench.py --batch 16384 --T 50 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2/bench_small.json 2> gpurun_out/r2/bench_small.err || { tail -5 gpurun_out/r2/bench_small.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2/bench_small.json"))
print(round(d["value"], 1), round(d["whole_job_mfma_frac"], 4), [(k["name"].split("(")[0], round(k["avg_us"], 1), round(k.get("tflops", 0))) for k in d["kernels"]])
PY
