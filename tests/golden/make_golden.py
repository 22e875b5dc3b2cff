"""Generates tests/golden/realspace_B4_L20_seed0.npz and realspace_B2_L32_seed0.npz (the headline lattice size)
from the C oracle (oracle/ref_cpu.c).

The reference holds no recorded outputs for this path and Keras is not installed, so these
vectors pin the ORACLE (regression) and the GPU float32 path against it; they are not Keras
outputs.  Inputs: x_T and per-step noise from np.random.RandomState(0) (legacy stream, frozen),
drawn in that order; weights weights/b_diffusion_model_2d_ising_realspace_unet.keras.
Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-model_amd")]

from isingdiff.keras_file import load_unet_spec  # noqa: E402
from oracle import ref_cpu  # noqa: E402


def make(ref, B, L, T, name):
    rs = np.random.RandomState(0)
    x_T = rs.standard_normal((B, L, L, 1)).astype(np.float32)
    z = rs.standard_normal((T, B, L, L, 1)).astype(np.float32)
    betas = ref_cpu.beta_schedule(T)
    x, scores = ref.sample(x_T, betas, T, noise_z=z, record_scores=True)
    spins = ref_cpu.sign_project(x)[..., 0]
    M, S = ref_cpu.observe(spins)
    out = {f"score_t{t}": scores[t].reshape(B, -1) for t in (999, 900, 500, 0)}
    out.update(spins=spins, M=M, S=S, x_final=x.reshape(B, -1), time_add_t500=ref.time_add(np.float32(500) / np.float32(1000)))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name), **out)
    print(name, "E/L^2 =", (-S / L ** 2).mean(), " <m> =", (M / L ** 2).mean(), " max|x| =", np.abs(x).max())


def main():
    spec = load_unet_spec(os.path.join(ROOT, "weights", "b_diffusion_model_2d_ising_realspace_unet.keras"))
    ref = ref_cpu.RefModel(spec)
    make(ref, 4, 20, 1000, "realspace_B4_L20_seed0.npz")
    make(ref, 2, 32, 1000, "realspace_B2_L32_seed0.npz")


if __name__ == "__main__":
    main()
