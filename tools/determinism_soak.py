#!/usr/bin/env python3
"""Runs the bench batch (65 536 lattices, bf16 fused kernels) for a few steps many times and checks that every
run gives bitwise the same state: a synchronisation slip between wavefronts would show as run-to-run noise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-model_amd"))
import torch
from isingdiff import load_model, sample, beta_schedule

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = 65536 if L == 32 else 65536 * 1024 // (L * L)
model = load_model(os.path.join(ROOT, "weights", "b_diffusion_model_2d_ising_realspace_unet.keras"), device=0, precision="bf16")
betas = beta_schedule(1000)[:4]
first = None
for i in range(runs):
    r = sample(model, n, L, T=4, betas=betas, seed=3, precision="bf16", keep_state=True, observables=False)
    if first is None:
        first = r.x_final.clone()
    elif not torch.equal(first, r.x_final):
        bad = (first != r.x_final).nonzero()
        print(f"run {i}: {bad.shape[0]} elements differ, first at {bad[0].tolist()}")
        sys.exit(1)
print(f"{runs} runs of {n} lattices (L={L}) bitwise identical")
