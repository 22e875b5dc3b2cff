import os, sys, subprocess, numpy as np
ROOT = os.getcwd()
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "diffusion-model_amd"))
from isingdiff import load_model, sample, beta_schedule
m = load_model("weights/b_diffusion_model_2d_ising_realspace_unet.keras", device=0, precision="bf16")
r = sample(m, 64, 32, T=2, betas=beta_schedule(1000)[:2], seed=5, precision="bf16", keep_state=True)
np.save(sys.argv[1], r.x_final.cpu().numpy())
'''
outs = []
A, B_ = (sys.argv[1:3] + ["diffusion-model_amd/isingdiff/libisingdiff.so", "diffusion-model_amd/isingdiff/abl/lib_0base.so"])[:2] if len(sys.argv) >= 3 else ("diffusion-model_amd/isingdiff/libisingdiff.so", "diffusion-model_amd/isingdiff/abl/lib_0base.so")
for name, lib in (("a", A), ("b", B_)):
    env = dict(os.environ, ISINGDIFF_LIB=os.path.join(ROOT, lib))
    out = f"gpurun_out/cmp_{name}.npy"
    subprocess.run([sys.executable, "-c", code, out], check=True, env=env)
    outs.append(np.load(out))
a, b = outs
d = a != b
print("differing elements:", int(d.sum()), "of", d.size)
if d.any():
    lat, row, col, _ = np.nonzero(d)
    print("lattices:", np.unique(lat)[:20], "rows:", np.unique(row), "cols:", np.unique(col)[:40])
    print("max abs diff", np.abs(a - b).max())
