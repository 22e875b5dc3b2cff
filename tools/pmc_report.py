#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc passes (tools/pmc_round.sh): for every kernel, each counter summed over
dispatches, and the derived shares the design notes quote.

Units (MI355X): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts;
SQ_BUSY_CYCLES counts per shader engine; SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is
busy, summed over SIMDs (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32); GRBM_GUI_ACTIVE is summed over the
8 XCDs, so clock = GRBM_GUI_ACTIVE / 8 / duration."""
import collections
import csv
import sys

N_SIMD = 256 * 4
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
cnt = collections.Counter()
for path in sys.argv[1:]:
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            k = k[5:] if k.startswith("void ") else k
            k = k[:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (path, k, r["Dispatch_Id"])
            if key not in seen and r["Counter_Name"].startswith("GRBM_GUI_ACTIVE"):
                seen.add(key)
                dur[(path, k)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                cnt[(path, k)] += 1
for k, c in acc.items():
    paths = [p for (p, kk) in dur if kk == k]
    if not paths or not any(w in k for w in ("fused", "strip", "irfft", "conv", "mlp", "bf_")):
        continue
    p0 = paths[0]
    d, n = dur[(p0, k)], cnt[(p0, k)]
    print(f"{k}\n   dispatches {n}  avg {1e6 * d / n:.1f} us")
    for name, v in sorted(c.items()):
        print(f"   {name:28s} {v / n / (2 if name.startswith('GRBM') and len(paths) > 1 else 1):16.0f} per dispatch")
    g = c.get("GRBM_GUI_ACTIVE", 0) / len(paths)
    if g and d:
        clk = g / 8 / d
        cyc = clk * d / n          # cycles per dispatch
        print(f"   clock {clk / 1e9:.3f} GHz")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            print(f"   MFMA pipe busy            {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / n / (cyc * N_SIMD):5.1f} % of SIMD-cycles")
        if "SQ_ACTIVE_INST_VALU" in c:
            print(f"   vector-ALU issue (x4)     {100 * 4 * c['SQ_ACTIVE_INST_VALU'] / n / (cyc * N_SIMD):5.1f} % of SIMD-cycles")
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            w = c["SQ_WAVE_CYCLES"]
            print(f"   of wavefront time: waiting (s_waitcnt / barrier) {100 * c['SQ_WAIT_ANY'] / w:4.1f} %, issue stalls {100 * c['SQ_WAIT_INST_ANY'] / w:4.1f} %, "
                  f"issuing {100 * c['SQ_ACTIVE_INST_ANY'] / w:4.1f} %")
        if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
            print(f"   LDS bank conflicts        {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:5.1f} % of LDS-active cycles")
        if "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"]:
            print(f"   vector-ALU instructions per MFMA {c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA'] - 1:5.2f}  (SQ_INSTS_VALU includes the MFMAs)")
