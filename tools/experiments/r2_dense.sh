#!/bin/bash
# Dense spectral sampler: bench, then rocprofv3 kernel stats of the same command.
set -o pipefail
mkdir -p gpurun_out/dense
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_dense.py -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tail -3 && timeout -k 10 300 python3 -u tools/bench_dense.py 65536 > gpurun_out/dense/bench.json 2> gpurun_out/dense/bench.err || { tail -5 gpurun_out/dense/bench.err; exit 1; }
cat gpurun_out/dense/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dense/trace -o s -- python3 tools/bench_dense.py 65536 > gpurun_out/dense/trace.log 2>&1 || { tail -5 gpurun_out/dense/trace.log; exit 1; }
python3 - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/dense/trace/s_kernel_stats.csv')))[:8]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/1e6:9.1f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
