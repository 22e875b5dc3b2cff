"""Distribution-level parity: energies and magnetisations of lattices sampled on the GPU (Philox noise,
float32 and bf16) against lattices sampled by the PyTorch-CPU restatement of the Keras sampler (numpy
noise), compared with two-sample Kolmogorov-Smirnov tests.

The CPU side is tests/golden/ks_reference_L{20,32}.npz: per-lattice E/L^2 and m of 2 048 (L = 20) and 1 024
(L = 32) lattices over the full T = 1000 schedule, written by tests/golden/make_ks_reference.py in the build
container (outputs of oracle/ref_torch.py, not of Keras).  L = 32 in bf16 is the bench configuration: the
fused band kernels, whose free-running output cannot be compared bit for bit with anything.

The samplers do not share noise, so this is a statistical statement ("indistinguishable at the 1 % level"),
complementary to the bit-exact injected-noise tests in test_gpu_parity.py.  Seeds are fixed and the kernels
are deterministic, so a pass is reproducible."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
T = 1000
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cpu_reference(L):
    g = np.load(os.path.join(GOLDEN, f"ks_reference_L{L}.npz"))
    assert int(g["L"]) == L and int(g["T"]) == T
    return g["energy_per_site"], g["magnetisation"]


def gpu_sample(model, n, L, seed, precision):
    from isingdiff import sample
    from isingdiff.observables import observe
    obs = observe(sample(model, n, L, T=T, seed=seed, precision=precision).spins)
    return -obs.S.cpu().numpy().astype(np.float64) / L ** 2, obs.M.cpu().numpy().astype(np.float64) / L ** 2


@pytest.mark.parametrize("L,precision,n", [(20, "fp32", 4096), (20, "bf16", 4096), (32, "bf16", 4096), (32, "fp32", 1024)])
def test_energy_and_magnetisation_distributions_match_cpu_sampler(gpu_real, L, precision, n):
    from scipy.stats import ks_2samp
    e_cpu, m_cpu = cpu_reference(L)
    assert len(e_cpu) >= (2048 if L == 20 else 1024)
    e_gpu, m_gpu = gpu_sample(gpu_real, n, L, 2024 + L, precision)
    p_e = ks_2samp(e_cpu, e_gpu).pvalue
    p_m = ks_2samp(np.abs(m_cpu), np.abs(m_gpu)).pvalue
    p_s = ks_2samp(m_cpu, m_gpu).pvalue
    assert min(p_e, p_m, p_s) > 0.01, (L, precision, p_e, p_m, p_s, e_cpu.mean(), e_gpu.mean(), np.abs(m_cpu).mean(),
                                       np.abs(m_gpu).mean())


@pytest.mark.parametrize("L", [20, 32])
def test_bf16_and_fp32_gpu_distributions_match(gpu_real, L):
    from scipy.stats import ks_2samp
    n = 4096 if L == 20 else 2048
    e32, m32 = gpu_sample(gpu_real, n, L, 7, "fp32")
    e16, m16 = gpu_sample(gpu_real, n, L, 8, "bf16")
    p_e = ks_2samp(e32, e16).pvalue
    p_m = ks_2samp(np.abs(m32), np.abs(m16)).pvalue
    assert min(p_e, p_m) > 0.01, (L, p_e, p_m, e32.mean(), e16.mean())
