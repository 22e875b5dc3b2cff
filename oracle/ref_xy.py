"""TEST INFRASTRUCTURE - ctypes binding of oracle/_ref/libxy_ref.so, which is the REFERENCE's own
XY_model/xymetoropolis.c compiled by `make -C oracle ref` (gcc -O3, the flags of the reference's Makefile) from
where the source lies under /root/reference.  Nothing of it is copied into this repository; the built library is
git-ignored and travels to the GPU box with the tree.  Only tests/, smoke() and bench baselines may import this.

Functions follow the reference's header (XY_model/xymetoropolis.h:38-43): field2d is passed by value."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libxy_ref.so")


class Field2d(C.Structure):          # xymetoropolis.h:15-20
    _fields_ = [("f", C.POINTER(C.c_double)), ("nx", C.c_int), ("ny", C.c_int), ("iox", C.c_int), ("ioy", C.c_int)]


class VortexCount(C.Structure):      # :23-27
    _fields_ = [("positive", C.c_int), ("negative", C.c_int)]


class SamplingStats(C.Structure):    # :30-36
    _fields_ = [("energy", C.c_double), ("magnetization", C.c_double), ("vortex_positive", C.c_double),
                ("vortex_negative", C.c_double)]


_lib = None


def available() -> bool:
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        L.xyenergy.restype = C.c_double
        L.xyenergy.argtypes = [Field2d]
        L.xymagnetization.restype = C.c_double
        L.xymagnetization.argtypes = [Field2d]
        L.xyvortex.restype = VortexCount
        L.xyvortex.argtypes = [Field2d]
        L.xy_sampling.restype = SamplingStats
        L.xy_sampling.argtypes = [C.POINTER(Field2d), C.c_double, C.c_int, C.c_int]
        _lib = L
    return _lib


def _field(a: np.ndarray) -> Field2d:
    assert a.dtype == np.float64 and a.ndim == 2 and a.flags.c_contiguous
    return Field2d(a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0], a.shape[1], 0, 0)


def observe(theta: np.ndarray):
    """(energy per site, magnetisation, +1 vortices, -1 vortices) of each (L, L) lattice of a (B, L, L) array."""
    theta = np.ascontiguousarray(theta, np.float64)
    out = []
    for a in theta:
        v = lib().xyvortex(_field(a))
        out.append((lib().xyenergy(_field(a)), lib().xymagnetization(_field(a)), v.positive, v.negative))
    e, m, p, n = zip(*out)
    return np.array(e), np.array(m), np.array(p, np.int32), np.array(n, np.int32)


def sampling(theta0: np.ndarray, beta: float, nsamples: int, seed: int = 42):
    """xy_sampling on a copy of theta0 (L, L) after srand(seed); returns (stats dict, final field)."""
    a = np.array(theta0, np.float64, order="C")
    C.CDLL(None).srand(C.c_uint(seed))
    f = _field(a)
    s = lib().xy_sampling(C.byref(f), C.c_double(beta), int(nsamples), 0)
    return dict(energy=s.energy, magnetization=s.magnetization, vortex_positive=s.vortex_positive,
                vortex_negative=s.vortex_negative), a
