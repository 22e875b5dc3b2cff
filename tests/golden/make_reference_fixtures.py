#!/usr/bin/env python3
"""Generates fixtures from the REFERENCE's own functions (build container only; the reference tree
does not travel to the GPU box, the small .npz files written here do):

  reference_onsager.npz      onsager_energy / onsager_magnetization / onsager_heat_capacity of
                             /root/reference/onsager.py:7-50 on a kT grid (J = 1 and J = 0.5)
  reference_ifft_L{L}.npz    random (non-Hermitian) half spectra and ising_ifft(spec, L) of
                             /root/reference/b_fft.py:12-17, plus ising_fft of +-1 lattices (:4-9)

  reference_observables.npz  ising_energy / ising_magnetization of
                             /root/reference/b_pure_metoropolis_2d_ising.py:42-60 on random and hand-made +-1 lattices

onsager.py and b_fft.py need only numpy / scipy and define functions without side effects at import.
b_pure_metoropolis_2d_ising.py cannot be imported (its line 4 imports a plotting package that is not installed, and
the module runs a temperature sweep at import), so the two function definitions - pure numpy - are taken out of its
syntax tree and compiled alone, with numpy as their only global; nothing else of the file runs.

    python tests/golden/make_reference_fixtures.py [--reference /root/reference]
"""
import argparse
import ast
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def extract_functions(path, names):
    """Compile only the named top-level function definitions of a Python source file (numpy as `np` in scope)."""
    with open(path, encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names), [n.name for n in picked]
    ns = {"np": np}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def observables_fixture(reference):
    src = os.path.join(reference, "b_pure_metoropolis_2d_ising.py")
    ising_energy, ising_magnetization = extract_functions(src, ["ising_energy", "ising_magnetization"])
    out = {}
    for L, B in ((4, 64), (20, 48), (32, 40), (64, 12), (128, 4)):
        rs = np.random.RandomState(2000 + L)
        p_up = rs.uniform(0.05, 0.95, size=(B, 1, 1))           # from nearly ordered to disordered lattices
        spins = np.where(rs.uniform(size=(B, L, L)) < p_up, 1, -1).astype(np.int8)
        yy, xx = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
        spins[0] = 1
        spins[1] = -1
        spins[2] = np.where((yy + xx) % 2 == 0, 1, -1)          # checkerboard: E = +2 L^2
        spins[3] = np.where(yy % 2 == 0, 1, -1)                 # stripes: vertical bonds -, horizontal bonds +
        x = spins.astype(np.float64)
        out[f"spins_L{L}"] = spins
        out[f"energy_J1_h0_L{L}"] = np.asarray(ising_energy(x, 1.0, 0.0), np.float64)
        out[f"energy_J05_h03_L{L}"] = np.asarray(ising_energy(x, 0.5, 0.3), np.float64)
        out[f"energy_int8_J1_h0_L{L}"] = np.asarray(ising_energy(spins.astype(np.int64), 1, 0), np.float64)
        out[f"magnetization_L{L}"] = np.asarray(ising_magnetization(x), np.float64)
    np.savez_compressed(os.path.join(HERE, "reference_observables.npz"), **out)
    print("wrote reference_observables.npz", {k: v.shape for k, v in out.items() if k.startswith("energy_J1")})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    observables_fixture(args.reference)
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
