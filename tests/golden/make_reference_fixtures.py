#!/usr/bin/env python3
"""Generates fixtures from the REFERENCE's own functions (build container only; the reference tree
does not travel to the GPU box, the small .npz files written here do):

  reference_onsager.npz      onsager_energy / onsager_magnetization / onsager_heat_capacity of
                             /root/reference/onsager.py:7-50 on a kT grid (J = 1 and J = 0.5)
  reference_ifft_L{L}.npz    random (non-Hermitian) half spectra and ising_ifft(spec, L) of
                             /root/reference/b_fft.py:12-17, plus ising_fft of +-1 lattices (:4-9)

The two modules need only numpy / scipy and define functions without side effects at import.

    python tests/golden/make_reference_fixtures.py [--reference /root/reference]
"""
import argparse
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    ons = load(os.path.join(args.reference, "onsager.py"), "reference_onsager")
    fft = load(os.path.join(args.reference, "b_fft.py"), "reference_b_fft")

    kT = np.concatenate([np.linspace(0.5, 4.0, 71), [2.0, 2.2691, 2.5, float(ons.theoretical_critical_kT)]])
    out = {"kT": kT, "critical_kT": np.float64(ons.theoretical_critical_kT)}
    for tag, J in (("J1", 1.0), ("J05", 0.5)):
        out[f"energy_{tag}"] = np.asarray(ons.onsager_energy(kT, J), np.float64)
        out[f"magnetization_{tag}"] = np.asarray(ons.onsager_magnetization(kT, J), np.float64)
        out[f"heat_capacity_{tag}"] = np.asarray(ons.onsager_heat_capacity(kT, J), np.float64)
    np.savez_compressed(os.path.join(HERE, "reference_onsager.npz"), **out)
    print("wrote reference_onsager.npz", {k: v.shape for k, v in out.items()})

    for L, B in ((4, 8), (20, 8), (32, 6), (64, 3), (128, 2)):
        rs = np.random.RandomState(1000 + L)
        # (B, L, L/2+1, 2) float32 Re/Im exactly as the spectral sampler's state holds them
        spec = (rs.standard_normal((B, L, L // 2 + 1, 2)) * 8.0).astype(np.float32)
        z = spec[..., 0].astype(np.float64) + 1j * spec[..., 1].astype(np.float64)
        real = fft.ising_ifft(z, L)
        spins = (rs.randint(0, 2, size=(B, L, L)) * 2 - 1).astype(np.int8)
        fwd = fft.ising_fft(spins, L)
        np.savez_compressed(os.path.join(HERE, f"reference_ifft_L{L}.npz"), spectrum=spec, ising_ifft=real,
                            spins=spins, ising_fft_re=fwd.real, ising_fft_im=fwd.imag)
        print(f"wrote reference_ifft_L{L}.npz", spec.shape, real.shape, real.dtype)


if __name__ == "__main__":
    main()
