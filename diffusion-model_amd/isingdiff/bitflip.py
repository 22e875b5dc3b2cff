"""MNIST bit-flip diffusion sampler of ``mnist_diffusion/mnist_diffusion.py`` on the GPU.

Mirrors the reference's two callables for this model:

* ``model([X, t_tensor]) -> flip_proba``   (``create_improved_diffusion_unet`` ``:162-240``)
* ``diffusion_backward(X_b, model, time_step)``   (``:43-52``): ``X ^= uniform < flip_proba``

The reference ships no weights for this model, so :func:`synthetic_weights` draws a Keras-style
initialisation (Glorot-uniform kernels) with non-trivial biases and GroupNorm affine parameters;
a dict with the same tensor names built from a trained model works the same way.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

from . import _lib

NBLK = 5
BLOCK_CIN = (32, 32, 64, 192, 96)
BLOCK_WIDTH = (32, 64, 128, 64, 32)
EMB_DIM = 32
TEMB = 128

# Order of the weight tensors at the C boundary (Keras layouts: conv (kh,kw,Cin,Cout), dense (in,out)).
TENSOR_NAMES = (["in_w", "in_b", "d1_w", "d1_b", "d2_w", "d2_b"]
                + [f"b{i}_{n}" for i in range(NBLK)
                   for n in ("t_w", "t_b", "g1_g", "g1_b", "c1_w", "c1_b", "g2_g", "g2_b", "c2_w", "c2_b", "s_w", "s_b")]
                + ["out_w", "out_b"])


def tensor_shapes() -> dict:
    s = {"in_w": (1, 1, 1, 32), "in_b": (32,), "d1_w": (EMB_DIM, TEMB), "d1_b": (TEMB,), "d2_w": (TEMB, TEMB),
         "d2_b": (TEMB,), "out_w": (1, 1, 32, 1), "out_b": (1,)}
    for i, (ci, w) in enumerate(zip(BLOCK_CIN, BLOCK_WIDTH)):
        s[f"b{i}_t_w"], s[f"b{i}_t_b"] = (TEMB, w), (w,)
        s[f"b{i}_g1_g"], s[f"b{i}_g1_b"] = (ci,), (ci,)
        s[f"b{i}_c1_w"], s[f"b{i}_c1_b"] = (3, 3, ci, w), (w,)
        s[f"b{i}_g2_g"], s[f"b{i}_g2_b"] = (w,), (w,)
        s[f"b{i}_c2_w"], s[f"b{i}_c2_b"] = (3, 3, w, w), (w,)
        if ci != w:
            s[f"b{i}_s_w"], s[f"b{i}_s_b"] = (1, 1, ci, w), (w,)
    return s


def synthetic_weights(seed: int = 0) -> dict:
    """Glorot-uniform kernels (Keras default initialiser); biases and GroupNorm beta ~ U(-0.1, 0.1),
    gamma ~ U(0.8, 1.2) so that every term of the forward pass is exercised."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in tensor_shapes().items():
        if name.endswith("_w"):
            fan_in = int(np.prod(shape[:-1]))
            fan_out = int(shape[-1]) * (int(np.prod(shape[:-2])) if len(shape) == 4 else 1)
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            out[name] = rs.uniform(-lim, lim, shape).astype(np.float32)
        elif name.endswith("_g"):
            out[name] = rs.uniform(0.8, 1.2, shape).astype(np.float32)
        else:
            out[name] = rs.uniform(-0.1, 0.1, shape).astype(np.float32)
    return out


TIME_INPUTS = ("float", "int64")


def sinusoidal_table(time_step: int, dim: int = EMB_DIM, max_positions: int = 10000, time_input: str = "float") -> np.ndarray:
    """Rows t = 0..time_step-1 of SinusoidalPositionalEmbedding (``:67-99``) evaluated at the
    sampler's input ``float32(t) / float32(time_step)`` (``:47``), in float32:
    ``freqs = exp(-log(max_positions) * arange(half) / (half - 1))``, ``[cos(args) | sin(args)]``.

    ``time_input`` settles an ambiguity of the reference that cannot be resolved without running Keras: the model
    declares ``time_input = Input(shape=(), dtype=tf.int64)`` (``:197``) but the sampler feeds ``float32(t / time_step)``
    in [0, 1) (``:47``).  ``"float"`` (default) embeds the float value, which is what the layer's own
    ``tf.cast(inputs, tf.float32)`` (``:79``) and the training loop's inputs suggest was meant; ``"int64"`` applies the
    cast to the declared input dtype first, as a Keras functional model does when it standardises its inputs: every
    step then sees position int64(t / time_step) = 0 and the model loses its time dependence.  Parity with Keras is
    unpinned either way (no weights, no recorded outputs, Keras not importable)."""
    if time_input not in TIME_INPUTS:
        raise ValueError(f"time_input must be one of {TIME_INPUTS}")
    half = dim // 2
    pos = (np.arange(time_step, dtype=np.float32) / np.float32(time_step)).astype(np.float32)
    if time_input == "int64":
        pos = pos.astype(np.int64).astype(np.float32)      # truncation toward zero, then the layer's cast to float32
    freqs = np.exp((np.float32(-np.log(np.float32(max_positions))) * np.arange(half, dtype=np.float32)
                    / np.float32(half - 1)).astype(np.float32)).astype(np.float32)
    args = (pos[:, None] * freqs[None, :]).astype(np.float32)
    emb = np.concatenate([np.cos(args), np.sin(args)], axis=-1).astype(np.float32)
    if dim % 2:
        emb = np.concatenate([emb, np.zeros_like(emb[:, :1])], axis=-1)
    return np.ascontiguousarray(emb)


assert len(TENSOR_NAMES) == _lib.BITFLIP_NTENSOR
BitflipDesc, BitflipCfg = _lib.BitflipDesc, _lib.BitflipCfg


class BitflipModel:
    """``model([X, t_tensor])`` of the reference (``:48``): X (B,H,W,1) bool / 0-1, t_tensor (B,) with one
    shared value ``t / time_step``; returns the flip probabilities (B,H,W,1) float32 (torch, on the GPU)."""

    def __init__(self, weights: dict, device: int = 0, time_step: int = 100, precision: str = "fp32",
                 time_input: str = "float"):
        import torch
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.torch = torch
        self.device = device
        self.time_step = time_step
        self.weights = weights
        shapes = tensor_shapes()
        keep, desc = [], BitflipDesc()
        desc.n_tensors = len(TENSOR_NAMES)
        for i, name in enumerate(TENSOR_NAMES):
            if name in shapes:
                a = np.ascontiguousarray(weights[name], np.float32)
                if a.shape != shapes[name]:
                    raise ValueError(f"{name}: expected shape {shapes[name]}, got {a.shape}")
                keep.append(a)
                desc.tensors[i] = a.ctypes.data
            else:
                desc.tensors[i] = None
        self._keep = keep
        self.time_input = time_input
        self._emb = sinusoidal_table(time_step, time_input=time_input)
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.load().idm_bitflip_create(C.byref(h), device, C.byref(desc)))
        self._h = h
        if precision == "bf16":     # 3x3 convolutions on bf16 MFMA; "fp32" is the bit-exact parity mode
            _lib.check(_lib.load().idm_bitflip_set_precision(h, _lib.IDM_BF16))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:   # (module globals may be gone at exit)
            _lib._lib.idm_bitflip_destroy(h)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _to_u8(self, X):
        torch = self.torch
        X = torch.as_tensor(np.asarray(X) if not torch.is_tensor(X) else X)
        if X.dim() == 4:
            if X.shape[-1] != 1:
                raise ValueError("X must be (B,H,W,1) or (B,H,W)")
            X = X[..., 0]
        return X.to(device=f"cuda:{self.device}", dtype=torch.uint8).contiguous()

    def time_vectors(self, t_index: int) -> np.ndarray:
        """The five per-block time vectors (320 floats) for step ``t_index`` (tests)."""
        out = self.torch.empty(sum(BLOCK_WIDTH), dtype=self.torch.float32, device=f"cuda:{self.device}")
        row = np.ascontiguousarray(self._emb[t_index])
        with self.torch.cuda.device(self.device):
            _lib.check(_lib.load().idm_bitflip_time(self._h, row.ctypes.data, out.data_ptr()))
        return out.cpu().numpy()

    def embedding_row(self, t_norm: float) -> np.ndarray:
        t = int(round(float(t_norm) * self.time_step))
        if not 0 <= t < self.time_step or np.float32(t) / np.float32(self.time_step) != np.float32(t_norm):
            raise ValueError("t must be k / time_step for an integer step k (as diffusion_backward passes it)")
        return np.ascontiguousarray(self._emb[t])

    def __call__(self, inputs, taps: bool = False):
        X, t = inputs
        torch = self.torch
        x8 = self._to_u8(X)
        B, H, W = x8.shape
        t_arr = np.asarray(t.cpu() if torch.is_tensor(t) else t, np.float32).reshape(-1)
        if t_arr.size not in (1, B) or np.any(t_arr != t_arr[0]):
            raise ValueError("every sample of a call shares one t (as in diffusion_backward)")
        row = self.embedding_row(t_arr[0])
        proba = torch.empty((B, H, W, 1), dtype=torch.float32, device=x8.device)
        tap_t, tap_p = None, None
        if taps:
            shp = [(H, W, 32), (H // 2, W // 2, 64), (H // 4, W // 4, 128), (H // 2, W // 2, 64), (H, W, 32)]
            tap_t = [torch.empty((B,) + s, dtype=torch.float32, device=x8.device) for s in shp]
            tap_p = (C.c_void_p * NBLK)(*[t_.data_ptr() for t_ in tap_t])
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().idm_bitflip_forward(self._h, x8.data_ptr(), row.ctypes.data, proba.data_ptr(), B, H, W,
                                                       tap_p, self._stream()))
        return (proba, tap_t) if taps else proba


def diffusion_backward(X_b, model: BitflipModel, time_step: int | None = None, *, uniforms=None, seed: int = 42,
                       first_global_index: int = 0, return_trajectory: bool = True, chunk: int = 0):
    """``diffusion_backward(X_b, model, time_step)`` (``mnist_diffusion.py:43-52``).

    Returns the stack of states in the reference's order (``[0]`` = input, ``[-1]`` = final) as a
    uint8 torch tensor (time_step+1, B, H, W, 1), or only the final state with
    ``return_trajectory=False``.  ``uniforms`` (time_step, B, H, W) float32, indexed by t, replaces
    the in-kernel Philox uniforms (deterministic mode for parity tests)."""
    torch = model.torch
    T = model.time_step if time_step is None else int(time_step)
    if T != model.time_step:
        raise ValueError("time_step differs from the model's embedding table")
    x = model._to_u8(X_b).clone()
    B, H, W = x.shape
    u = None
    if uniforms is not None:
        u = torch.as_tensor(np.asarray(uniforms, np.float32) if not torch.is_tensor(uniforms) else uniforms)
        u = u.to(device=x.device, dtype=torch.float32).contiguous()
        if tuple(u.shape) != (T, B, H, W):
            raise ValueError("uniforms must be (time_step, B, H, W)")
    traj = torch.empty((T + 1, B, H, W), dtype=torch.uint8, device=x.device) if return_trajectory else None
    cfg = BitflipCfg(T=T, emb_table_host=model._emb.ctypes.data, seed=seed, first_global_index=first_global_index,
                     chunk=chunk)
    with torch.cuda.device(model.device):
        _lib.check(_lib.load().idm_bitflip_sample(model._h, C.byref(cfg), x.data_ptr(),
                                                  None if u is None else u.data_ptr(),
                                                  None if traj is None else traj.data_ptr(), B, H, W, model._stream()))
    return traj[..., None] if return_trajectory else x[..., None]
