#!/usr/bin/env python3
"""Slot times of the copy-wavefront fused_B kernel (library built with -DIDM_TIMELINE, run with ISINGDIFF_B_SPLIT=1)."""
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
t = np.zeros((2, 64, 16), dtype=np.int64)
lib.idm_debug_timeline3.argtypes = [ctypes.c_void_p]
assert lib.idm_debug_timeline3(t.ctypes.data) == 0
n = int((t[0, :, 0] > 0).sum())
sl = slice(4, n - 1)
per = np.diff(t[0, 4:n, 0]).mean()
print(f"iterations recorded {n}; period per pair of bands {per:.0f} cycles")
names = ["conv8 X | update Y', carry", "conv8 Y | stage-in X', prefetch Y'", "conv9 X | stage-in Y', prefetch X''", "conv9 Y | update X, carry"]
if os.environ.get("ISINGDIFF_B_PAIR") == "1":   # the merged-stream kernel: three barriers per pair of bands
    names = ["conv8 (X, Y) loop + heads | carry, x fetch", "activation of conv8 (X, Y) | update, padding rows", "conv9 (X, Y) | stage-in, prefetch"]
NS = len(names)
for role, rn in ((0, "MFMA wavefront 0"), (1, "other wavefront 4")):
    a = t[role, sl].astype(np.float64)
    prev_depart = np.concatenate([t[role, 3:n - 2, 2 * NS - 1:2 * NS].astype(np.float64), a[:, 1:2 * NS:2][:, :NS - 1]], axis=1)   # departure from the previous barrier
    work = a[:, 0:2 * NS:2] - prev_depart
    wait = a[:, 1:2 * NS:2] - a[:, 0:2 * NS:2]
    print(rn)
    for k in range(NS):
        print(f"   slot {k + 1} ({names[k]}): work {work[:, k].mean():7.0f}   wait at barrier {wait[:, k].mean():7.0f}")

# inside the MFMA wavefront's slots of stream X (stamps 8-13: operands addressed, last MFMA issued, sinks done)
if NS == 4:
    a = t[0, sl].astype(np.float64)
    dep_prev = t[0, 3:n - 2, 7].astype(np.float64)          # departure from barrier 4 of the previous iteration
    for name, k0, start in (("conv8 X (slot 1)", 8, dep_prev), ("conv9 X (slot 3)", 11, a[:, 3])):
        ok = a[:, k0] > 0
        bar = a[:, 0] if k0 == 8 else a[:, 4]
        print(f"   {name}: set-up {np.mean((a[:, k0] - start)[ok]):6.0f}   MFMA loop {np.mean((a[:, k0 + 1] - a[:, k0])[ok]):6.0f}"
              f"   sinks {np.mean((a[:, k0 + 2] - a[:, k0 + 1])[ok]):6.0f}   re-arm to barrier {np.mean((bar - a[:, k0 + 2])[ok]):6.0f}")
