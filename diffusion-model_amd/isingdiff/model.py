"""``ScoreModel``: the drop-in for the Keras model object used as ``score_model([x, t_norm])``
(b_diffusion_model_unet_2d_ising.py:88) and ``load_model`` for
``keras.models.load_model(path)`` (:199-201).  The forward runs in libisingdiff.so on the GPU;
there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .keras_file import UNetSpec, load_unet_spec

PRECISIONS = {"fp32": _lib.IDM_FP32, "float32": _lib.IDM_FP32, "bf16": _lib.IDM_BF16, "bfloat16": _lib.IDM_BF16}


def _precision_code(p) -> int:
    if isinstance(p, int):
        return p
    try:
        return PRECISIONS[p]
    except KeyError:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}") from None


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class ScoreModel:
    """UNet score model resident on one GPU.

    ``model([x, t_norm])`` mirrors Keras ``Model.__call__``: ``x`` is ``(B,H,W,C)`` float32,
    ``t_norm`` is ``(B,1)`` (or a scalar); as in the reference's sampler every row of
    ``t_norm`` must hold the same value.  Returns the same kind of object as ``x``
    (numpy in → numpy out, torch CUDA tensor in → torch CUDA tensor out).
    """

    def __init__(self, spec: UNetSpec, device: int = 0, precision="bf16"):
        if not torch.cuda.is_available():
            raise _lib.IdmError("no GPU visible: isingdiff has no CPU fallback")
        self.spec = spec
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self.precision = _precision_code(precision)
        self.channels = spec.channels
        lib = _lib.load()
        desc = _lib.ModelDesc()
        desc.channels = spec.channels
        desc.base_filters = spec.base_filters
        desc.batchnorm = int(spec.batchnorm)
        self._keep = []

        def ptr(a: Optional[np.ndarray]):
            if a is None:
                return None
            a = np.ascontiguousarray(a, np.float32)
            self._keep.append(a)
            return a.ctypes.data

        for i, c in enumerate(spec.convs):
            desc.conv_w[i] = ptr(c.kernel)
            desc.conv_b[i] = ptr(c.bias)
            desc.bn_scale[i] = ptr(c.bn_scale)
            desc.bn_shift[i] = ptr(c.bn_shift)
        for i, d in enumerate(spec.dense):
            desc.dense_w[i] = ptr(d.kernel)
            desc.dense_b[i] = ptr(d.bias)
        h = C.c_void_p()
        _lib.check(lib.idm_model_create(C.byref(h), self.device_index, C.byref(desc)))
        self._h = h
        self._keep = []

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.load().idm_model_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    @property
    def handle(self):
        if not self._h:
            raise _lib.IdmError("model is closed")
        return self._h

    def _t_scalar(self, t_norm) -> float:
        if isinstance(t_norm, torch.Tensor):
            t = t_norm.detach().float().reshape(-1).cpu().numpy()
        else:
            t = np.asarray(t_norm, np.float32).reshape(-1)
        if t.size == 0:
            raise ValueError("t_norm is empty")
        if not np.all(t == t[0]):
            raise ValueError("every lattice of a call must share one t_norm (as in diffusion_backward)")
        return float(t[0])

    def _to_device(self, x):
        if isinstance(x, torch.Tensor):
            return x.to(self.device, torch.float32).contiguous(), True
        return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(self.device), False

    def __call__(self, inputs: Sequence, precision=None):
        x, t_norm = inputs
        xd, is_torch = self._to_device(x)
        if xd.dim() != 4 or xd.shape[3] != self.channels:
            raise ValueError(f"x must be (B,H,W,{self.channels}), got {tuple(xd.shape)}")
        B, H, W, _ = xd.shape
        if B == 0:
            out = torch.empty_like(xd)
            return out if is_torch else out.cpu().numpy()
        prec = self.precision if precision is None else _precision_code(precision)
        score = torch.empty_like(xd)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().idm_forward(self.handle, xd.data_ptr(), C.c_float(self._t_scalar(t_norm)),
                                               score.data_ptr(), B, H, W, prec, _stream_ptr(self.device)))
        return score if is_torch else score.cpu().numpy()

    def predict(self, inputs: Sequence, batch_size: Optional[int] = None, precision=None) -> np.ndarray:
        """numpy-returning equivalent (Keras ``Model.predict``); ``batch_size`` bounds device memory."""
        x, t_norm = inputs
        n = len(x)
        if batch_size is None or batch_size >= n:
            out = self([x, t_norm], precision)
            return out.cpu().numpy() if isinstance(out, torch.Tensor) else out
        t = self._t_scalar(t_norm)
        parts = []
        for i in range(0, n, batch_size):
            o = self([x[i:i + batch_size], t], precision)
            parts.append(o.cpu().numpy() if isinstance(o, torch.Tensor) else o)
        return np.concatenate(parts, 0)

    def forward_taps(self, x, t_norm, precision=None):
        """Score plus the 11 per-layer activations as float32 CUDA tensors (test hook)."""
        xd, _ = self._to_device(x)
        B, H, W, _ = xd.shape
        hp, wp = H + (4 - H % 4) % 4, W + (4 - W % 4) % 4
        f, c = self.spec.base_filters, self.channels
        shapes = [(hp, wp, f), (hp, wp, f), (hp // 2, wp // 2, 2 * f), (hp // 2, wp // 2, 2 * f),
                  (hp // 4, wp // 4, 4 * f), (hp // 4, wp // 4, 4 * f), (hp // 2, wp // 2, 2 * f),
                  (hp // 2, wp // 2, 2 * f), (hp, wp, f), (hp, wp, f)]
        taps = [torch.empty((B,) + s, dtype=torch.float32, device=self.device) for s in shapes]
        arr = (C.c_void_p * 11)(*[t.data_ptr() for t in taps], None)
        prec = self.precision if precision is None else _precision_code(precision)
        score = torch.empty_like(xd)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().idm_forward_taps(self.handle, xd.data_ptr(), C.c_float(self._t_scalar(t_norm)),
                                                    score.data_ptr(), B, H, W, prec,
                                                    C.cast(arr, C.POINTER(C.c_void_p)), _stream_ptr(self.device)))
        return score, taps

    def set_option(self, name: str, value: int):
        """Tuning switch, e.g. ``set_option("fused", 0)`` forces one kernel per layer in bf16."""
        _lib.check(_lib.load().idm_set_option(self.handle, name.encode(), int(value)))

    def plan_chunk(self, B: int, H: int, W: int, precision=None):
        """How a batch of ``B`` lattices will be cut: ``(lattices per chunk, workspace bytes per lattice)``."""
        prec = self.precision if precision is None else _precision_code(precision)
        chunk, per = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.load().idm_plan_chunk(self.handle, int(B), int(H), int(W), prec, C.byref(chunk), C.byref(per)))
        return chunk.value, per.value

    def profile_begin(self):
        """Start per-kernel event timing of this model's launches (see idm_profile_begin)."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().idm_profile_begin(self.handle))

    def profile_end(self):
        """Stop timing; returns [{name, launches, ms, flop, bytes}] per kernel class."""
        ents = (_lib.ProfileEntry * 32)()
        n = C.c_int(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().idm_profile_end(self.handle, ents, 32, C.byref(n)))
        return [dict(name=e.name.decode(), launches=e.launches, ms=e.ms, flop=e.flop, bytes=e.bytes)
                for e in ents[:min(n.value, 32)]]

    def time_table(self, T: int) -> np.ndarray:
        out = np.empty((T, 4 * self.spec.base_filters), np.float32)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().idm_time_table(self.handle, T, out.ctypes.data))
        return out


def load_model(path: str, device: int = 0, precision="bf16") -> ScoreModel:
    """Counterpart of ``keras.models.load_model(path)`` for the reference's UNet ``.keras`` files."""
    return ScoreModel(load_unet_spec(path), device=device, precision=precision)
