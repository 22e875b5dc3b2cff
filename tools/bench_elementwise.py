#!/usr/bin/env python3
"""Achieved HBM bandwidth of the stand-alone scheduler / observable kernels (the float32 path's update, the
sign projection, the observables, the Philox draw and the spectral irfft2+sign), against the 8 TB/s peak.
Algorithmic bytes only (every operand read or written once).  Prints one JSON line."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-model_amd"))
import torch
from isingdiff import _lib
from isingdiff.observables import observe, sign_project
from isingdiff.spectral import irfft2_sign

lib = _lib.load()
B, L = 65536, 32
n = B * L * L
dev = torch.device("cuda:0")
x = torch.randn(n, device=dev)
score = torch.randn(n, device=dev)
z = torch.randn(n, device=dev)
spec = torch.randn(B, L, L // 2 + 1, 2, device=dev)
spins = sign_project(x.view(B, L, L)).contiguous()
stream = None


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


out = {}
def rec(name, nbytes, fn):
    t = timed(fn)
    out[name] = {"us": round(t * 1e6, 1), "GBps": round(nbytes / t / 1e9, 1), "frac_of_8TBps": round(nbytes / t / 8e12, 3)}

rec("update (injected noise: x, score, z in; x out)", 16 * n,
    lambda: _lib.check(lib.idm_update(0, x.data_ptr(), score.data_ptr(), z.data_ptr(), 0.01, n, stream)))
rec("update (no noise: x, score in; x out)", 12 * n,
    lambda: _lib.check(lib.idm_update(0, x.data_ptr(), score.data_ptr(), None, 0.01, n, stream)))
rec("philox_normal (out only)", 4 * n,
    lambda: _lib.check(lib.idm_philox_normal(0, z.data_ptr(), B, L * L, 42, 0, 7, stream)))
rec("sign_project (f32 in, int8 out)", 5 * n, lambda: sign_project(x.view(B, L, L)))
rec("observe (int8 in)", 1 * n, lambda: observe(spins))
rec("irfft2_sign (spectrum in, int8 out)", spec.numel() * 4 + n, lambda: irfft2_sign(spec))
print(json.dumps({"lattices": B, "L": L, "kernels": out}))
