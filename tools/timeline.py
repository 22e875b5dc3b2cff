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
buf = np.zeros((5, 2, 64, 32), dtype=np.int64)
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
        # inside the convolutions (last piece of the band): operand addressing, MFMA loop, epilogue
        c1 = sl[:, [2, 10, 11, 12, 3]].astype(np.float64)
        c2 = sl[:, [4, 18, 19, 20, 5]].astype(np.float64)
        for nm, c in (("conv1", c1), ("conv2", c2)):
            dd = np.diff(c, axis=1).mean(axis=0)
            print(f"   {nm}: before={dd[0]:.0f}  mfma-loop={dd[1]:.0f}  epilogue={dd[2]:.0f}  after={dd[3]:.0f}")
            if os.environ.get("TL_VERBOSE") and nm == "conv1" and k != 0:
                per = np.diff(c, axis=1)
                print("      per band mfma-loop:", " ".join(f"{v:.0f}" for v in per[:, 1]))
                print("      per band epilogue: ", " ".join(f"{v:.0f}" for v in per[:, 2]))
                print("      wait for constants (LDS):", " ".join(f"{v:.0f}" for v in (sl[:, 16] - sl[:, 11]).astype(np.float64)))
                print("      wait for the last MFMAs: ", " ".join(f"{v:.0f}" for v in (sl[:, 17] - sl[:, 16]).astype(np.float64)))
                seq = [11, 16, 17, 24, 25, 26, 27, 13, 14, 12]
                lab = ["consts", "last-mfma", "g0", "g1", "g2", "g3", "tile1-end", "tile2", "end"]
                dd2 = np.diff(sl[:, seq].astype(np.float64), axis=1)
                ok = np.abs(dd2).max(axis=1) < 1e6
                for r in range(4):
                    sel = ok & (np.arange(len(sl)) % 4 == r)
                    if sel.any():
                        print(f"      epilogue pieces, bands {r} mod 4: " + "  ".join(f"{l}={v:.0f}" for l, v in zip(lab, dd2[sel].mean(axis=0))))
                for g in range(0):
                    prev = sl[:, 17] if g == 0 else sl[:, 24 + g - 1]
                    print(f"      first tile, channel group {g}:", " ".join(f"{v:.0f}" for v in (sl[:, 24 + g] - prev).astype(np.float64)[:24]))
                t0 = (sl[:, 13] - sl[:, 17]).astype(np.float64)
                t1 = (sl[:, 14] - sl[:, 13]).astype(np.float64)
                print("      first tile's epilogue:", " ".join(f"{v:.0f}" for v in t0))
                print("      second tile's:        ", " ".join(f"{v:.0f}" for v in t1))
