#!/bin/bash
# Kernel shares of the spectral UNet sampler (strip-kernel path) under rocprofv3.
set -o pipefail
out=gpurun_out/specprof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 150 python3 tools/bench_spectral.py 16384 > $out/bench_16384.json 2>$out/bench.err || { tail $out/bench.err; exit 1; }
cat $out/bench_16384.json
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out/trace -o s --output-format csv -- python3 tools/bench_spectral.py 4096 > $out/prof.log 2>&1 || { tail -5 $out/prof.log; exit 1; }
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/specprof/trace/s_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f} %  calls {r['Calls']:>6s}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
