"""isingdiff — MI355X-native reverse-diffusion sampler for the reference's 2-D Ising UNet score
models.  Public surface mirrors the reference script's callables:

    load_model(path)                          keras.models.load_model
    model([x, t_norm]) / model.predict(...)   score_model([x, t_norm])
    diffusion_backward(x, betas, model, T)    diffusion_backward
    sample(model, n, L)                       the script's sampling + sign projection + observables
    ising_energy / ising_magnetization        b_pure_metoropolis_2d_ising helpers
    onsager_energy …                          onsager.py
    bitflip.BitflipModel / bitflip.diffusion_backward
                                              mnist_diffusion.py: model([X, t]) and its XOR sampler
"""
from .keras_file import UNetSpec, load_unet_spec
from .onsager import onsager_energy, onsager_heat_capacity, onsager_magnetization, theoretical_critical_kT

__all__ = ["UNetSpec", "load_unet_spec", "onsager_energy", "onsager_heat_capacity", "onsager_magnetization",
           "theoretical_critical_kT", "load_model", "ScoreModel", "diffusion_backward", "sample", "beta_schedule",
           "observe", "sign_project", "ising_energy", "ising_magnetization", "Observables", "shard_range",
           "allreduce_observables", "irfft2_sign"]

_LAZY = {
    "load_model": "model", "ScoreModel": "model",
    "diffusion_backward": "sampler", "sample": "sampler", "beta_schedule": "sampler", "SampleResult": "sampler",
    "observe": "observables", "sign_project": "observables", "ising_energy": "observables",
    "ising_magnetization": "observables", "Observables": "observables",
    "irfft2_sign": "spectral",
    "shard_range": "distributed", "allreduce_observables": "distributed",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module("." + _LAZY[name], __name__), name)
    raise AttributeError(name)
