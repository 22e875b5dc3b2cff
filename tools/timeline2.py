#!/usr/bin/env python3
"""Relative phase of the fused_B workgroups that share a CU (library built with -DIDM_TIMELINE).

    ISINGDIFF_LIB=.../lib_TIMELINE.so python tools/timeline2.py
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-model_amd"))
import torch
from isingdiff import load_model, sample, beta_schedule
from isingdiff import _lib

model = load_model(os.path.join(ROOT, "weights", "b_diffusion_model_2d_ising_realspace_unet.keras"), device=0, precision="bf16")
B = int(os.environ.get("TL_B", "16384"))
sample(model, B, 32, T=2, betas=beta_schedule(2), seed=1)
torch.cuda.synchronize()
lib = _lib.load()
t = np.zeros((1024, 16, 16), dtype=np.int64)
hw = np.zeros((1024, 2), dtype=np.uint32)
lib.idm_debug_timeline2.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert lib.idm_debug_timeline2(t.ctypes.data, hw.ctypes.data) == 0
# HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (se is 3 bits on gfx94x: 15:13)
key = {}
for b in range(1024):
    if t[b, 0, 0] == 0:
        continue
    h = int(hw[b, 0]); x = int(hw[b, 1]) & 0xF
    k = (x, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 0xF)
    key.setdefault(k, []).append(b)
groups = [v for v in key.values() if len(v) >= 2]
print(f"{len(key)} CUs seen, {len(groups)} with two or more recorded workgroups")
names = ["stage-in", "B1", "conv8 start", "conv8 done", "B2", "conv9 done", "B3", "out done", "end"]
shown = 0
for g in groups:
    a, b = g[0], g[1]
    ta, tb = t[a], t[b]
    base = ta[0, 0]
    # phase of workgroup b's band starts relative to workgroup a's, in units of a's period
    per = np.diff(ta[:, 0]).mean()
    off = ((tb[:, 0] - ta[:, 0]) % per) / per
    if shown < 6:
        print(f"workgroups {a},{b}: period {per:.0f} cycles; start of b's bands at phase {np.round(off, 2)} of a's period")
        # where in a's band does b's conv8 MFMA loop (stamps 10..11) fall, and vice versa
        for nm, (p, q) in (("a", (ta, tb)), ("b", (tb, ta))):
            m0, m1 = p[4:12, 10], p[4:12, 11]      # MFMA loop of conv8
            e1 = p[4:12, 12]
            print(f"   {nm}: band start {p[4,0]-base:>8d}  conv8 mfma {m0[0]-base:>8d}..{m1[0]-base:>8d}  epilogue ..{e1[0]-base:>8d}  B2 {p[4,4]-base:>8d}  conv9 ..{p[4,5]-base:>8d}  B3 {p[4,6]-base:>8d}  out ..{p[4,7]-base:>8d}")
        shown += 1
offs = []
for g in groups:
    ta, tb = t[g[0]], t[g[1]]
    per = np.diff(ta[:, 0]).mean()
    offs.append((((tb[4:12, 0] - ta[4:12, 0]) % per) / per).mean())
print("distribution of the phase offset between co-resident workgroups (fraction of a band period):")
print(np.round(np.histogram(offs, bins=10, range=(0, 1))[0] / max(len(offs), 1), 2))
