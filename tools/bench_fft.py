#!/usr/bin/env python3
"""Throughput of the LDS-resident irfft2 + sign kernel (idm_irfft2_sign) against the HBM roof: algorithmic bytes
8*L*(L/2+1) in + L*L out per lattice.  Prints one JSON line.   python tools/bench_fft.py [L ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-model_amd"))
import torch
from isingdiff.spectral import irfft2_sign

sizes = [int(a) for a in sys.argv[1:]] or [20, 32, 64, 128]
out = {}
for L in sizes:
    B = max(1024, (65536 * 32 * 32) // (L * L))
    spec = torch.randn(B, L, L // 2 + 1, 2, device="cuda")
    irfft2_sign(spec); torch.cuda.synchronize()
    reps = 20
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        irfft2_sign(spec)
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / reps * 1e-3
    nbytes = spec.numel() * 4 + B * L * L
    out[f"L={L}"] = {"lattices": B, "us": round(t * 1e6, 1), "GBps": round(nbytes / t / 1e9, 1),
                     "frac_of_8TBps": round(nbytes / t / 8e12, 3)}
print(json.dumps({"kernel": "irfft2_sign", "sizes": out}))
