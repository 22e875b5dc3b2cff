"""Bitwise comparison of the bf16 sampler's final state between two builds of the library.
    python tools/experiments/compare_libs_batch.py LIB_A LIB_B BATCH [T] [VAR=VALUE ...]   (the variables are set for LIB_A only)
"""
import os, sys, subprocess, numpy as np
ROOT = os.getcwd()
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "diffusion-model_amd"))
from isingdiff import load_model, sample, beta_schedule
B, T = int(sys.argv[2]), int(sys.argv[3])
m = load_model("weights/b_diffusion_model_2d_ising_realspace_unet.keras", device=0, precision="bf16")
r = sample(m, B, 32, T=T, betas=beta_schedule(1000)[:T], seed=5, precision="bf16", keep_state=True)
torch.cuda.synchronize()
np.save(sys.argv[1], r.x_final.cpu().numpy())
'''
A, B_, batch = sys.argv[1], sys.argv[2], sys.argv[3]
T = sys.argv[4] if len(sys.argv) > 4 else "3"
extra = dict(v.split("=", 1) for v in sys.argv[5:])
outs = []
for name, lib in (("a", A), ("b", B_)):
    env = dict(os.environ, ISINGDIFF_LIB=os.path.join(ROOT, lib))
    if name == "a":
        env.update(extra)
    out = f"gpurun_out/cmp_{name}.npy"
    subprocess.run([sys.executable, "-c", code, out, batch, T], check=True, env=env, timeout=90)
    outs.append(np.load(out))
a, b = outs
d = a != b
print(f"batch {batch} T {T}: differing elements:", int(d.sum()), "of", d.size, " finite:", bool(np.isfinite(a).all()))
if d.any():
    idx = np.nonzero(d.reshape(d.shape[0], 32, 32))
    print("lattices:", np.unique(idx[0])[:20], "rows:", np.unique(idx[1]), "cols:", np.unique(idx[2])[:40])
    print("max abs diff", np.nanmax(np.abs(a - b)))
    sys.exit(1)
