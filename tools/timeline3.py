#!/usr/bin/env python3
"""Slot times of the split-role fused_B kernel (library built with -DIDM_TIMELINE -DIDM_FUSED_B_SPLIT)."""
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
names = ["conv8 X | epi9+update Y'", "hand-over X | stage-in Y", "conv8 Y | epi8 X + prefetch", "hand-over Y | -", "conv9 X | epi8 Y",
         "hand-over X | -", "conv9 Y | epi9+update X", "hand-over Y | stage-in X' + prefetch"]
for role, rn in ((0, "MFMA wavefront 0"), (1, "other wavefront 4")):
    a = t[role, sl].astype(np.float64)
    prev_depart = np.concatenate([t[role, 3:n - 2, 15:16].astype(np.float64), a[:, 1::2][:, :7]], axis=1)   # departure from the previous barrier
    work = a[:, 0::2] - prev_depart
    wait = a[:, 1::2] - a[:, 0::2]
    print(rn)
    for k in range(8):
        print(f"   slot {k + 1} ({names[k]}): work {work[:, k].mean():7.0f}   wait at barrier {wait[:, k].mean():7.0f}")
