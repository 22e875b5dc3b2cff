#!/usr/bin/env python3
"""Per-phase cycle counts of the fused band kernels (library built with -DIDM_TIMELINE).

    ISINGDIFF_LIB=.../lib_TIMELINE.so python tools/timeline.py
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-model_amd"))
import torch
from isingdiff import load_model, sample, beta_schedule
from isingdiff import _lib

model = load_model(os.path.join(ROOT, "weights", "b_diffusion_model_2d_ising_realspace_unet.keras"), device=0, precision="bf16")
B = int(os.environ.get("TL_B", "4096"))
sample(model, B, 32, T=4, betas=beta_schedule(4), seed=1)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((5, 2, 64, 10), dtype=np.int64)
lib.idm_debug_timeline.argtypes = [ctypes.c_void_p]
rc = lib.idm_debug_timeline(buf.ctypes.data)
assert rc == 0, rc
names = ["A", "M1", "M2", "M3", "B"]
phases = ["stage-in", "bar0", "conv1", "bar1", "conv2", "pf+bar2", "out", "bar3"]
for k in range(5):
    for w in range(2):
        t = buf[k, w]
        n = int((t[:, 0] > 0).sum())
        if n < 6:
            continue
        sl = t[2:n - 1]
        d = np.diff(sl[:, :9], axis=1).astype(np.float64)
        period = np.diff(t[2:n, 0]).mean()
        wall = (t[n - 1, 9] - t[2, 9]) / 100e6
        cyc = (t[n - 1, 0] - t[2, 0])
        print(f"{names[k]} wave{0 if w == 0 else 3}: bands={n} period={period:.0f} cyc  ({cyc / max(wall, 1e-12) / 1e9:.2f} GHz counter)")
        print("   " + "  ".join(f"{p}={v:.0f}" for p, v in zip(phases, d.mean(axis=0))))
