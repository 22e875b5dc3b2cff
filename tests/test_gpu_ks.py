"""Distribution-level parity: energies and magnetisations of lattices sampled on the GPU (Philox noise,
float32 and bf16) against lattices sampled by the PyTorch-CPU restatement of the Keras sampler (numpy
noise), compared with two-sample Kolmogorov-Smirnov tests.

The samplers do not share noise, so this is a statistical statement ("indistinguishable at the 0.1 %
level"), complementary to the bit-exact injected-noise tests in test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
L, T = 20, 1000


@pytest.fixture(scope="module")
def cpu_samples(real_spec):
    import torch
    from oracle.ref_torch import TorchUNet, diffusion_backward, sign_project, ising_energy, ising_magnetization
    from oracle.ref_cpu import beta_schedule
    torch.set_num_threads(min(16, torch.get_num_threads()))
    rs = np.random.RandomState(1234)
    B = 192
    model = TorchUNet(real_spec)
    x = rs.standard_normal((B, L, L, 1)).astype(np.float32)
    xf = diffusion_backward(model, x, beta_schedule(T), T, lambda t: rs.standard_normal(x.shape).astype(np.float32))
    spins = sign_project(xf)[..., 0]
    return ising_energy(spins, 1.0, 0.0) / L ** 2, ising_magnetization(spins)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_energy_and_magnetisation_distributions_match_cpu_sampler(cpu_samples, gpu_real, precision):
    from scipy.stats import ks_2samp
    from isingdiff import sample
    from isingdiff.observables import observe
    e_cpu, m_cpu = cpu_samples
    n = 2048
    res = sample(gpu_real, n, L, T=T, seed=2024, precision=precision)
    obs = observe(res.spins)
    e_gpu = -obs.S.cpu().numpy().astype(np.float64) / L ** 2
    m_gpu = obs.M.cpu().numpy().astype(np.float64) / L ** 2
    p_e = ks_2samp(e_cpu, e_gpu).pvalue
    p_m = ks_2samp(np.abs(m_cpu), np.abs(m_gpu)).pvalue
    p_s = ks_2samp(m_cpu, m_gpu).pvalue
    assert min(p_e, p_m, p_s) > 1e-3, (precision, p_e, p_m, p_s, e_cpu.mean(), e_gpu.mean(), np.abs(m_cpu).mean(),
                                       np.abs(m_gpu).mean())


def test_bf16_and_fp32_gpu_distributions_match(gpu_real):
    from scipy.stats import ks_2samp
    from isingdiff import sample
    from isingdiff.observables import observe
    out = {}
    for prec, seed in (("fp32", 7), ("bf16", 8)):
        obs = observe(sample(gpu_real, 4096, L, T=T, seed=seed, precision=prec).spins)
        out[prec] = (-obs.S.cpu().numpy() / L ** 2, np.abs(obs.M.cpu().numpy()) / L ** 2)
    p_e = ks_2samp(out["fp32"][0], out["bf16"][0]).pvalue
    p_m = ks_2samp(out["fp32"][1], out["bf16"][1]).pvalue
    assert min(p_e, p_m) > 1e-3, (p_e, p_m, out["fp32"][0].mean(), out["bf16"][0].mean())
