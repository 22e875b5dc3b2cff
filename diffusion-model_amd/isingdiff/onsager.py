"""Exact 2-D Ising results used as validation constants (the reference's onsager.py:7-50).

Written around the arithmetic-geometric mean so that no elliptic-integral library is needed:
K(m) = π / (2·AGM(1, √(1−m))),  E(m) = K(m)·(1 − Σ 2^{n−1} c_n²)  with parameter m = k².
"""
from __future__ import annotations

import math

theoretical_critical_kT = 2.2691


def _ellip_ke(m: float):
    a, b, c = 1.0, math.sqrt(1.0 - m), math.sqrt(m)
    s, p = 0.5 * c * c, 0.5
    for _ in range(64):
        if abs(c) < 1e-17:
            break
        a, b, c = 0.5 * (a + b), math.sqrt(a * b), 0.5 * (a - b)
        p *= 2.0
        s += p * c * c
    k = math.pi / (2.0 * a)
    return k, k * (1.0 - s)


def onsager_energy(kT: float, J: float) -> float:
    """Energy per site (onsager.py:7-19)."""
    x = 2.0 * J / kT
    k = 2.0 * math.sinh(x) / math.cosh(x) ** 2
    K, _ = _ellip_ke(k * k)
    th = math.tanh(x)
    return -J / th * (1.0 + (2.0 / math.pi) * (2.0 * th * th - 1.0) * K)


def onsager_magnetization(kT: float, J: float) -> float:
    """Spontaneous magnetisation per site, zero above kT_c = 2J/asinh(1) (onsager.py:22-31)."""
    if kT >= 2.0 * J / math.asinh(1.0):
        return 0.0
    return (1.0 - math.sinh(2.0 * J / kT) ** -4) ** 0.125


def onsager_heat_capacity(kT: float, J: float) -> float:
    """Heat capacity per site in units of k_B (onsager.py:34-50)."""
    x = 2.0 * J / kT
    k = 2.0 * math.sinh(x) / math.cosh(x) ** 2
    K, E = _ellip_ke(k * k)
    kp = 2.0 * math.tanh(x) ** 2 - 1.0
    return (2.0 / math.pi) * (x / math.tanh(x)) ** 2 * (2.0 * K - 2.0 * E - (1.0 - kp) * (math.pi / 2.0 + kp * K))
