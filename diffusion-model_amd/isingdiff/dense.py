"""Dense spectral score model of ``b_diffusion_model_fft_2d_ising.py`` on the GPU.

Mirrors the reference's callables for this model:

* ``score_model(xt)`` with ``xt = concat([x, t_norm], axis=1)`` (``:28-30``; Sequential of
  Dense(440, tanh) x4 + Dense(440), ``:105-114``)
* ``diffusion_backward(x_asym, beta_lst, score_model, T)`` (``:19-38``)
* the post-processing ``reshape -> Re | Im -> ising_ifft -> sign`` (``:142-146``)

Weights come from the saved ``models/b_diffusion_model_2d_ising.keras`` (read without Keras).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import hashlib
import json
import zipfile
from typing import List

import numpy as np

from . import _lib
from .hdf5_min import read_datasets


@dataclasses.dataclass
class DenseSpec:
    path: str
    sha256: str
    kernels: List[np.ndarray]      # (in, out) float32
    biases: List[np.ndarray]
    activations: List[str]         # "tanh" | "linear"

    @property
    def n(self) -> int:
        return int(self.kernels[-1].shape[1])


def load_dense_spec(path: str) -> DenseSpec:
    """Parse a Keras-3 ``.keras`` archive holding a Sequential of Dense layers on ``[x | t_norm]``."""
    raw = open(path, "rb").read()
    with zipfile.ZipFile(path) as z:
        cfg = json.loads(z.read("config.json"))
        data = read_datasets(z.read("model.weights.h5"))
    if cfg.get("class_name") != "Sequential":
        raise ValueError(f"{path}: expected a Sequential model, found {cfg.get('class_name')}")
    kernels, biases, acts = [], [], []
    for layer in cfg["config"]["layers"]:
        if layer["class_name"] == "InputLayer":
            continue
        if layer["class_name"] != "Dense":
            raise ValueError(f"{path}: unsupported layer {layer['class_name']}")
        name, act = layer["config"]["name"], layer["config"]["activation"]
        if act not in ("tanh", "linear"):
            raise ValueError(f"{path}: unsupported activation {act}")
        k = np.ascontiguousarray(data[f"/layers/{name}/vars/0"], np.float32)
        b = np.ascontiguousarray(data[f"/layers/{name}/vars/1"], np.float32)
        if kernels and k.shape[0] != kernels[-1].shape[1]:
            raise ValueError(f"{path}: layer {name} does not chain")
        kernels.append(k); biases.append(b); acts.append(act)
    if not kernels or kernels[0].shape[0] != kernels[-1].shape[1] + 1:
        raise ValueError(f"{path}: the model must map n + 1 inputs ([x | t_norm]) to n outputs")
    return DenseSpec(path=path, sha256=hashlib.sha256(raw).hexdigest(), kernels=kernels, biases=biases, activations=acts)


class DenseScoreModel:
    """``score_model(xt)`` of the reference: xt (B, n+1) = [x | t_norm] with one shared t_norm per call
    (as ``diffusion_backward`` builds it); returns the score (B, n) float32 (torch, on the GPU).
    ``precision``: "fp32" (canonical arithmetic, bit-exact against the CPU restatement) or "bf16"
    (Dense layers on bf16 MFMA, float32 accumulation and state)."""

    def __init__(self, spec: DenseSpec, device: int = 0, precision: str = "fp32"):
        import torch
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.torch, self.spec, self.device, self.n = torch, spec, device, spec.n
        self.precision = _lib.IDM_BF16 if precision == "bf16" else _lib.IDM_FP32
        d = _lib.MlpDesc()
        d.n_layers = len(spec.kernels)
        if d.n_layers > _lib.MLP_MAXLAYERS:
            raise ValueError("too many layers")
        for i, (k, b, a) in enumerate(zip(spec.kernels, spec.biases, spec.activations)):
            d.in_dim[i], d.out_dim[i], d.act[i] = k.shape[0], k.shape[1], int(a == "tanh")
            d.w[i], d.b[i] = k.ctypes.data, b.ctypes.data
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.load().idm_mlp_create(C.byref(h), device, C.byref(d)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:   # (module globals may be gone at exit)
            _lib._lib.idm_mlp_destroy(h)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, a):
        torch = self.torch
        a = torch.as_tensor(np.asarray(a, np.float32) if not torch.is_tensor(a) else a)
        return a.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous()

    def score(self, x, t_norm: float):
        x = self._f32(x)
        if x.dim() != 2 or x.shape[1] != self.n:
            raise ValueError(f"x must be (B, {self.n})")
        out = self.torch.empty_like(x)
        with self.torch.cuda.device(self.device):
            _lib.check(_lib.load().idm_mlp_forward_precision(self._h, x.data_ptr(), float(np.float32(t_norm)), out.data_ptr(),
                                                             x.shape[0], self.precision, self._stream()))
        return out

    def __call__(self, xt):
        xt = self._f32(xt)
        if xt.dim() != 2 or xt.shape[1] != self.n + 1:
            raise ValueError(f"xt must be (B, {self.n + 1}) = [x | t_norm]")
        t = xt[:, self.n]
        if bool((t != t[0]).any()):
            raise ValueError("every row of a call shares one t_norm (as in diffusion_backward)")
        return self.score(xt[:, :self.n].contiguous(), float(t[0]))


def load_dense_model(path: str, device: int = 0, precision: str = "fp32") -> DenseScoreModel:
    return DenseScoreModel(load_dense_spec(path), device, precision)


def diffusion_backward(x_asym, beta_lst, score_model: DenseScoreModel, T: int, *, noise=None, seed: int = 42,
                       first_global_index: int = 0, return_trajectory: bool = True):
    """``diffusion_backward(x_asym, beta_lst, score_model, T)`` (``b_diffusion_model_fft_2d_ising.py:19-38``).

    Returns (T+1, B, n) with ``[0]`` = final state (the reference's reversed stack), or the final state
    alone.  ``noise``: None = in-kernel Philox normals; an array (T, B, n) of standard normals indexed by
    t (deterministic mode); False = no noise."""
    torch = score_model.torch
    x = score_model._f32(x_asym).clone()
    B, n = x.shape
    betas = np.ascontiguousarray(np.asarray(beta_lst.cpu() if torch.is_tensor(beta_lst) else beta_lst, np.float32).reshape(-1))
    if T <= 0 or betas.shape[0] < T:
        raise ValueError(f"beta_lst must have at least T={T} entries (the library reads betas[0..T-1])")
    betas = np.ascontiguousarray(betas[:T])
    if x.dim() != 2 or n != score_model.n:
        raise ValueError(f"x_asym must be (B, {score_model.n})")
    z = None
    if noise is not None and noise is not False:
        z = score_model._f32(noise)
        if tuple(z.shape) != (T, B, n):
            raise ValueError("noise must be (T, B, n)")
    traj = torch.empty((T + 1, B, n), dtype=torch.float32, device=x.device) if return_trajectory else None
    cfg = _lib.SampleCfg(T=T, precision=score_model.precision, betas_host=betas.ctypes.data, seed=seed,
                         first_global_index=first_global_index, init_state=0, add_noise=int(noise is not False), chunk=0)
    with torch.cuda.device(score_model.device):
        _lib.check(_lib.load().idm_mlp_sample(score_model._h, C.byref(cfg), x.data_ptr(), None if z is None else z.data_ptr(),
                                              None if traj is None else traj.data_ptr(), B, score_model._stream()))
    return traj if return_trajectory else x


def spectral_to_spins(x, L: int):
    """(B, L*2*(L/2+1)) rows [Re(ky, 0..L/2) | Im(ky, 0..L/2)] per ky (``:142-146``) -> ±1 spins (B, L, L) int8
    through irfft2 + sign on the GPU."""
    from .spectral import irfft2_sign
    import torch
    x = torch.as_tensor(x)
    B, half = x.shape[0], L // 2 + 1
    spec = x.reshape(B, L, 2, half).permute(0, 1, 3, 2).contiguous()      # (B, L, L/2+1, [Re, Im])
    return irfft2_sign(spec)
