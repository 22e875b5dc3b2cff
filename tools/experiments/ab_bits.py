#!/usr/bin/env python3
"""Two builds of the library must give the same bits: final float32 state of the bf16 L=32 sampler (Philox noise) for a few batch
sizes, hashed in a child process per library.  usage: ab_bits.py LIB_A LIB_B"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = f"""
import os, sys, hashlib, numpy as np
sys.path.insert(0, {os.path.join(ROOT, 'diffusion-model_amd')!r})
from isingdiff import load_model, sample, beta_schedule
m = load_model({os.path.join(ROOT, 'weights', 'b_diffusion_model_2d_ising_realspace_unet.keras')!r}, device=0, precision='bf16')
h = hashlib.sha256()
for B in (1, 5, 777, 2049, 8192):
    r = sample(m, B, 32, T=4, betas=beta_schedule(1000)[:4], seed=5, precision='bf16', keep_state=True)
    h.update(r.x_final.cpu().numpy().tobytes())
print(os.path.basename(os.environ['ISINGDIFF_LIB']), h.hexdigest()[:24])
"""
outs = []
for lib in sys.argv[1:3]:
    r = subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, ISINGDIFF_LIB=os.path.abspath(lib)), timeout=300, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1])
    outs.append(r.stdout.strip().splitlines()[-1].split()[-1])
print("identical" if outs[0] == outs[1] else "DIFFERENT")
sys.exit(0 if outs[0] == outs[1] else 1)
