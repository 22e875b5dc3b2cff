"""Errors of the bit-flip UNet's bf16 mode against the float32 oracle (max abs of the flip probabilities, relative L2 of the
ResBlock outputs), for the activation storage the environment selects.  python tools/experiments/bitflip_errors.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diffusion-model_amd")); sys.path.insert(0, ROOT)
import numpy as np
from isingdiff import bitflip
from oracle import ref_cpu
w = bitflip.synthetic_weights(0)
m = bitflip.BitflipModel(w, device=0, time_step=100, precision="bf16")
ref = ref_cpu.RefBitflip(w, bitflip.TENSOR_NAMES)
emb = bitflip.sinusoidal_table(100)
for H, W in ((28, 28), (12, 20)):
    x = (np.random.RandomState(11).rand(6, H, W, 1) < 0.5).astype(np.uint8)
    for t in (99, 30, 0):
        p, taps = m([x, np.full(6, np.float32(t) / np.float32(100), np.float32)], taps=True)
        pr, tr = ref.forward(x[..., 0], emb[t], taps=True)
        d = np.abs(p.cpu().numpy()[..., 0] - pr)
        print(H, W, t, "proba max abs %.4f rms %.5f" % (d.max(), np.sqrt((d ** 2).mean())),
              "taps rel L2", " ".join("%.4f" % (np.linalg.norm(a.cpu().numpy() - b) / np.linalg.norm(b)) for a, b in zip(taps, tr)))
