#!/bin/bash
# MNIST bit-flip sampler: its GPU tests, the throughput bench and a rocprofv3 kernel-stats run of the bench.
set -o pipefail
mkdir -p gpurun_out/bitflip
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_bitflip.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/bitflip/tests.log 2>&1 || { tail -40 gpurun_out/bitflip/tests.log; exit 1; }
tail -2 gpurun_out/bitflip/tests.log
timeout -k 10 300 python3 -u tools/bench_bitflip.py 4096 > gpurun_out/bitflip/bench.json 2> gpurun_out/bitflip/bench.err || { tail -5 gpurun_out/bitflip/bench.err; exit 1; }
cat gpurun_out/bitflip/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bitflip/trace -o s -- python3 tools/bench_bitflip.py 1024 bf16 > gpurun_out/bitflip/trace.log 2>&1 || { tail -5 gpurun_out/bitflip/trace.log; exit 1; }
python3 - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/bitflip/trace/s_kernel_stats.csv')))[:24]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/1e6:9.1f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
