"""Effective shader clock per kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE run:
clock ~= GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration."""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"].startswith("GRBM_GUI_ACTIVE"):
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            rows[r["Kernel_Name"][:60]].append((float(r["Counter_Value"]), dur))
for k, v in rows.items():
    c = sum(a for a, _ in v); d = sum(b for _, b in v)
    print(f"{k:62s} n={len(v):4d} avg_dur_us={1e6*d/len(v):9.1f} clock_GHz={c/8/d/1e9:6.3f}")
