"""Per-layer comparison of two builds of the library on the same input (bf16 fused path, L=32)."""
import os, sys, subprocess, numpy as np
ROOT = os.getcwd()
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "diffusion-model_amd"))
from isingdiff import load_model
m = load_model("weights/b_diffusion_model_2d_ising_realspace_unet.keras", device=0, precision="bf16")
x = np.random.RandomState(0).standard_normal((8, 32, 32, 1)).astype(np.float32)
score, taps = m.forward_taps(x, 0.5, precision="bf16")
np.savez(sys.argv[1], score=score.cpu().numpy(), **{f"t{i}": t.cpu().numpy() for i, t in enumerate(taps)})
'''
A, B_ = sys.argv[1:3]
outs = []
for name, lib in (("a", A), ("b", B_)):
    env = dict(os.environ, ISINGDIFF_LIB=os.path.join(ROOT, lib))
    out = f"gpurun_out/cmpt_{name}.npz"
    subprocess.run([sys.executable, "-c", code, out], check=True, env=env)
    outs.append(np.load(out))
a, b = outs
for k in a.files:
    x, y = a[k], b[k]
    ok = np.isfinite(y)
    d = (x != y) & ok & np.isfinite(x)
    msg = f"{k}: differing {int(d.sum())} of {d.size}"
    if d.any():
        idx = np.nonzero(d)
        msg += f"  rows {np.unique(idx[1])[:40]}  max abs {np.abs(x[d] - y[d]).max():.3e}"
    print(msg)
