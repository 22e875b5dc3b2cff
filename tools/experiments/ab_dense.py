#!/usr/bin/env python3
"""A/B of two builds of the library on the dense bf16 sampler: the same bits (final state after 12 steps, Philox noise,
ragged batch) and the throughput of each.  usage: ab_dense.py LIB_A LIB_B (each run in a child process)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = f"""
import os, sys, time, json, hashlib, numpy as np, torch
sys.path.insert(0, {os.path.join(ROOT, 'diffusion-model_amd')!r})
from isingdiff import dense
m = dense.load_dense_model({os.path.join(ROOT, 'weights', 'b_diffusion_model_2d_ising.keras')!r}, device=0, precision='bf16')
betas = np.linspace(1e-4, 0.02, 1000).astype(np.float32)
rs = np.random.RandomState(3)
h = hashlib.sha256()
for B in (1, 70, 129, 1000):
    x0 = rs.standard_normal((B, 440)).astype(np.float32)
    h.update(dense.diffusion_backward(x0, betas[:12], m, 12, seed=4, first_global_index=7, return_trajectory=False).cpu().numpy().tobytes())
    z = rs.standard_normal((3, B, 440)).astype(np.float32)
    h.update(dense.diffusion_backward(x0, betas[:3], m, 3, noise=z, return_trajectory=False).cpu().numpy().tobytes())
    h.update(m.score(x0, 0.5).cpu().numpy().tobytes())
B, T = 65536, 200
x0 = rs.standard_normal((B, 440)).astype(np.float32)
dense.diffusion_backward(x0[:256], betas[-8:], m, 8, return_trajectory=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
dense.diffusion_backward(x0, betas[-T:], m, T, return_trajectory=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({{"lib": os.path.basename(os.environ.get("ISINGDIFF_LIB", "default")), "sha": h.hexdigest()[:16], "M_sample_steps_per_s": B * T / dt / 1e6}}))
"""
for lib in sys.argv[1:] + sys.argv[1:]:
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, ISINGDIFF_LIB=os.path.abspath(lib)), timeout=300)
