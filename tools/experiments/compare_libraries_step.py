"""Two builds of the library on one reverse step: injected noise, in-kernel noise, no update."""
import os, sys, subprocess, numpy as np
ROOT = os.getcwd()
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "diffusion-model_amd"))
from isingdiff import load_model, sample, beta_schedule, diffusion_backward
m = load_model("weights/b_diffusion_model_2d_ising_realspace_unet.keras", device=0, precision="bf16")
rs = np.random.RandomState(0)
x = rs.standard_normal((64, 32, 32, 1)).astype(np.float32)
z = rs.standard_normal((1, 64, 32, 32, 1)).astype(np.float32)
betas = beta_schedule(1000)[:1]
inj = diffusion_backward(x, betas, m, 1, noise=z, precision="bf16")
r1 = sample(m, 64, 32, T=1, betas=betas, seed=5, precision="bf16", keep_state=True, x_T=x)
tr, scs = diffusion_backward(x, betas, m, 1, noise=z, precision="bf16", return_scores=True)
np.savez(sys.argv[1], score_in_step=np.asarray(scs[0].cpu() if hasattr(scs[0], "cpu") else scs[0]), injected=np.asarray(inj[0].cpu() if hasattr(inj[0], "cpu") else inj[0]), philox=r1.x_final.cpu().numpy())
'''
A, B_ = sys.argv[1:3]
outs = []
for name, lib in (("a", A), ("b", B_)):
    env = dict(os.environ, ISINGDIFF_LIB=os.path.join(ROOT, lib))
    out = f"gpurun_out/cmps_{name}.npz"
    subprocess.run([sys.executable, "-c", code, out], check=True, env=env)
    outs.append(np.load(out))
a, b = outs
for k in a.files:
    d = a[k] != b[k]
    msg = f"{k}: differing {int(d.sum())} of {d.size}"
    if d.any():
        idx = np.nonzero(d)
        msg += f"  rows {np.unique(idx[-3])[:40]}  max abs {np.abs(a[k][d] - b[k][d]).max():.3e}"
    print(msg)
