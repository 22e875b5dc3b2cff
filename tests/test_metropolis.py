"""Checkerboard Metropolis generator: CPU oracle statistics against Onsager, GPU bit-exact against the oracle.

The reference's generator (b_pure_metoropolis_2d_ising.py:20-105) draws from numpy's global RNG one site at a
time; parity with it is statistical (exact 2-D Ising results of onsager.py), not bitwise."""
import numpy as np
import pytest

from isingdiff.onsager import onsager_energy, onsager_magnetization
from oracle import ref_cpu


def table(beta, J=1.0, h=0.0):
    s = np.array([-1.0, 1.0])[:, None]
    nn = np.array([-4.0, -2.0, 0.0, 2.0, 4.0])[None, :]
    return np.exp(-beta * 2.0 * s * (J * nn + h)).astype(np.float32)


def test_oracle_reaches_onsager_values():
    L, B, kT = 32, 64, 2.0
    x = ref_cpu.metropolis(np.ones((B, L, L), np.int8), table(1 / kT), 600, seed=1)
    M, S = ref_cpu.observe(x)
    e = (-S / L ** 2).mean()
    m = (np.abs(M) / L ** 2).mean()
    assert abs(e - onsager_energy(kT, 1.0)) < 0.01, e            # -1.7456
    assert abs(m - onsager_magnetization(kT, 1.0)) < 0.01, m     # 0.9113
    # detailed balance sanity: a hot start relaxes to the same energy
    rs = np.random.RandomState(0)
    y = ref_cpu.metropolis(rs.choice([-1, 1], size=(B, L, L)).astype(np.int8), table(1 / 3.0), 600, seed=2)
    _, S3 = ref_cpu.observe(y)
    assert abs((-S3 / L ** 2).mean() - onsager_energy(3.0, 1.0)) < 0.015


def test_oracle_chains_are_reproducible_and_continuable():
    x0 = np.ones((3, 8, 8), np.int8)
    a = ref_cpu.metropolis(x0, table(0.4), 10, seed=7)
    b = ref_cpu.metropolis(ref_cpu.metropolis(x0, table(0.4), 4, seed=7), table(0.4), 6, seed=7, sweep0=4)
    assert np.array_equal(a, b)
    assert not np.array_equal(a[0], a[1])                        # chains differ by their global index
    c = ref_cpu.metropolis(x0[1:], table(0.4), 10, seed=7, first_global_index=1)
    assert np.array_equal(a[1:], c)


@pytest.mark.gpu
@pytest.mark.parametrize("L,h", [(20, 0.0), (32, 0.3), (6, 0.0)])
def test_gpu_bit_exact_against_oracle(L, h):
    import torch
    from isingdiff import metropolis
    rs = np.random.RandomState(L)
    x0 = rs.choice([-1, 1], size=(9, L, L)).astype(np.int8)
    tab = metropolis.acceptance_table(0.45, 1.0, h)
    want = ref_cpu.metropolis(x0, tab, 25, seed=11, first_global_index=5, sweep0=3)
    got = metropolis.sweep(torch.as_tensor(x0, device="cuda:0").clone(), 0.45, 1.0, h, 25, seed=11, first_global_index=5,
                           sweep0=3).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_gpu_pure_metropolis_signature_and_statistics():
    from isingdiff import metropolis
    L, kT = 32, 2.0
    samples, energy, mag = metropolis.pure_metropolis(np.ones((L, L)), 1 / kT, 1.0, 0.0, n_mcs_equilibration=500,
                                                      n_mcs_measurement=20480, measurement_interval_mcs=10)
    assert samples.shape == (2048, L, L) and energy.shape == (2048,) and mag.shape == (2048,)
    assert set(np.unique(samples)) == {-1, 1}
    assert abs(energy.mean() / L ** 2 - onsager_energy(kT, 1.0)) < 0.004
    assert abs(np.abs(mag).mean() - onsager_magnetization(kT, 1.0)) < 0.004
