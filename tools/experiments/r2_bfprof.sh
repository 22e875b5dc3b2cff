#!/bin/bash
set -o pipefail
out=gpurun_out/r2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bf_trace -o s -- python3 tools/bench_bitflip.py 1024 > $out/bf_trace.log 2>&1 || { tail -5 $out/bf_trace.log; exit 1; }
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r2/bf_trace/s_kernel_stats.csv")))
for r in rows[:24]: print(f"{float(r['Percentage']):6.2f}% {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.1f}us  {r['Name'][:110]}")
PY
