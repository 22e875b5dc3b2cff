#!/usr/bin/env python3
"""Throughput of the dense spectral sampler: B samples x T steps, one JSON line per precision
(float32 canonical kernel, bf16 MFMA kernel)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-model_amd"))
import numpy as np
import torch
from isingdiff import dense

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 200
betas = np.linspace(1e-4, 0.02, 1000).astype(np.float32)[-T:]
x0 = np.random.RandomState(0).standard_normal((B, 440)).astype(np.float32)
for precision, peak, label in (("fp32", 157.3e12, "frac_of_fp32_vector_peak"), ("bf16", 2500e12, "frac_of_bf16_mfma_peak")):
    model = dense.load_dense_model(os.path.join(ROOT, "weights", "b_diffusion_model_2d_ising.keras"), device=0, precision=precision)
    dense.diffusion_backward(x0[:256], betas, model, 8, return_trajectory=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = dense.diffusion_backward(x0, betas, model, T, return_trajectory=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flop = 2.0 * (441 * 440 + 4 * 440 * 440) * B * T
    print(json.dumps({"workload": f"dense spectral sampler, n=440, {precision}", "samples": B, "T": T,
                      "sample_steps_per_s": B * T / dt, "samples_per_s_full_schedule": B * T / dt / 1000,
                      "tflops": flop / dt / 1e12, label: flop / dt / peak}))
