#!/usr/bin/env python3
"""Run-to-run determinism of the other samplers at bench size: the spectral UNet (bf16, fused pairs), the bit-flip UNet (bf16
mode, half activations) and the dense model (bf16) run a few steps several times with the same seed; every run must give
bitwise the same state.  python tools/determinism_soak_workloads.py [runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-model_amd"))
import numpy as np, torch
from isingdiff import load_model, sample, beta_schedule, bitflip

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
W = os.path.join(ROOT, "weights")


def check(name, fn):
    first = None
    for i in range(runs):
        out = fn()
        if first is None:
            first = out.clone()
        elif not torch.equal(first, out):
            print(f"{name}: run {i} differs in {(first != out).sum().item()} elements")
            sys.exit(1)
    print(f"{name}: {runs} runs bitwise identical")


spec = load_model(os.path.join(W, "b_diffusion_model_2d_ising_unet.keras"), device=0, precision="bf16")
b4 = beta_schedule(1000)[:4]
check("spectral UNet, 65 536 spectra, 4 steps",
      lambda: sample(spec, 65536, 32, T=4, betas=b4, seed=3, precision="bf16", keep_state=True, observables=False).x_final)
for L, n in ((64, 8192), (128, 2048)):
    real = load_model(os.path.join(W, "b_diffusion_model_2d_ising_realspace_unet.keras"), device=0, precision="bf16")
    check(f"real-space UNet, L = {L}, {n} lattices, 4 steps",
          lambda: sample(real, n, L, T=4, betas=b4, seed=3, precision="bf16", keep_state=True, observables=False).x_final)
bf = bitflip.BitflipModel(bitflip.synthetic_weights(0), device=0, time_step=100, precision="bf16")
x0 = np.random.RandomState(0).randint(2, size=(16384, 28, 28, 1)).astype(np.uint8)
check("bit-flip UNet, 16 384 images, 100 steps", lambda: bitflip.diffusion_backward(x0, bf, 100, seed=5, return_trajectory=False))
