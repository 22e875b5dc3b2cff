#!/usr/bin/env python3
"""Throughput of the spectral (FFT-UNet) sampler, bf16: B lattices x T steps; strip kernels vs per-layer kernels."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-model_amd"))
import torch
from isingdiff import load_model, sample, beta_schedule

B, L, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 32, 50
model = load_model(os.path.join(ROOT, "weights", "b_diffusion_model_2d_ising_unet.keras"), device=0, precision="bf16")
flop = 2.0 * 9 * (2 * 32 + 32 * 32 * 2 + 0) * 0  # filled below from the layer table
hp, wp = 32, 20
macs = 9 * (2 * 32 + 32 * 32) * hp * wp + 9 * (32 * 64 + 64 * 64) * hp * wp / 4 + 9 * (64 * 128 + 128 * 128) * hp * wp / 16 \
    + 9 * (192 * 64 + 64 * 64) * hp * wp / 4 + 9 * (96 * 32 + 32 * 32) * hp * wp + 32 * 2 * hp * wp
for fused in (1, 0):
    model.set_option("fused", fused)
    sample(model, 256, L, T=4, betas=beta_schedule(1000)[-4:], seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = sample(model, B, L, T=T, betas=beta_schedule(1000)[-T:], seed=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"path": "strip kernels" if fused else "per-layer kernels", "lattices": B, "T": T,
                      "lattice_steps_per_s": B * T / dt, "tflops": 2 * macs * B * T / dt / 1e12,
                      "frac_of_bf16_peak": 2 * macs * B * T / dt / 2.5e15}))
