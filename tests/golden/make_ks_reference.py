#!/usr/bin/env python3
"""Generates tests/golden/ks_reference_L{20,32}.npz: per-lattice energy per site and magnetisation of
lattices drawn by the PyTorch-CPU restatement of the reference sampler (oracle/ref_torch.py:
create_unet_model + diffusion_backward + sign projection + ising_energy / ising_magnetization,
b_diffusion_model_unet_2d_ising.py:22-98,:203-223) with numpy noise, full T = 1000 schedule.

These are outputs of the CPU restatement, not of Keras (which cannot be imported here): they give the
GPU distribution tests a large CPU-side sample without paying for it on the GPU box.  Run in the build
container (minutes of CPU time):

    python tests/golden/make_ks_reference.py [--threads 8]
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "diffusion-model_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--only", type=int, default=0, help="generate one lattice size only")
    args = ap.parse_args()
    import torch
    from isingdiff.keras_file import load_unet_spec
    from oracle.ref_torch import (TorchUNet, beta_schedule, diffusion_backward, ising_energy, ising_magnetization,
                                  sign_project)
    torch.set_num_threads(args.threads)
    spec = load_unet_spec(os.path.join(ROOT, "weights", "b_diffusion_model_2d_ising_realspace_unet.keras"))
    model = TorchUNet(spec)
    T = 1000
    betas = beta_schedule(T)
    for L, n, seed, batch in ((20, 2048, 20200, 256), (32, 1024, 32200, 128)):
        if args.only and args.only != L:
            continue
        rs = np.random.RandomState(seed)
        e_all, m_all = [], []
        t0 = time.time()
        for b0 in range(0, n, batch):
            x = rs.standard_normal((batch, L, L, 1)).astype(np.float32)
            xf = diffusion_backward(model, x, betas, T, lambda t: rs.standard_normal(x.shape).astype(np.float32))
            spins = sign_project(xf)[..., 0]
            e_all.append(ising_energy(spins, 1.0, 0.0) / L ** 2)
            m_all.append(ising_magnetization(spins))
            print(f"L={L}: {b0 + batch}/{n} lattices, {time.time() - t0:.0f} s", flush=True)
        out = os.path.join(HERE, f"ks_reference_L{L}.npz")
        np.savez_compressed(out, energy_per_site=np.concatenate(e_all), magnetisation=np.concatenate(m_all),
                            L=L, T=T, seed=seed, sampler="oracle/ref_torch.py (PyTorch-CPU float32), numpy RandomState noise")
        print("wrote", out)


if __name__ == "__main__":
    main()
