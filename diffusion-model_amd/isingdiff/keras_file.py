"""Read the reference's saved ``.keras`` score models without Keras or TensorFlow.

A ``.keras`` archive (Keras 3 format, written by ``score_model.save`` at
b_diffusion_model_unet_2d_ising.py:197 and
b_diffusion_model_fft_unet_2d_ising.py:201) is a zip of ``metadata.json``,
``config.json`` (the Functional graph) and ``model.weights.h5``.  Variables live
at ``/layers/<layer name>/vars/<i>``: Conv2D/Dense ``0`` = kernel
``(kh, kw, Cin, Cout)`` / ``(in, out)``, ``1`` = bias; BatchNormalization
``0..3`` = gamma, beta, moving_mean, moving_variance.

``load_unet_spec`` checks that the graph is the UNet of ``create_unet_model``
(b_diffusion_model_unet_2d_ising.py:22-78, BatchNorm variant
b_diffusion_model_fft_unet_2d_ising.py:25-92) and returns its tensors in a
layout-neutral ``UNetSpec``.
"""
from __future__ import annotations

import dataclasses
import hashlib
import io
import json
import zipfile
from typing import Dict, List, Optional, Tuple

import numpy as np

from .hdf5_min import read_datasets

BN_EPSILON_DEFAULT = 1e-3


@dataclasses.dataclass
class ConvLayer:
    name: str
    kernel: np.ndarray          # (kh, kw, Cin, Cout) float32, Keras order (cross-correlation)
    bias: np.ndarray            # (Cout,) float32
    silu: bool                  # swish applied after (bias, optional BN)
    bn_scale: Optional[np.ndarray] = None   # (Cout,) gamma / sqrt(var + eps), float32
    bn_shift: Optional[np.ndarray] = None   # (Cout,) beta - mean * scale, float32


@dataclasses.dataclass
class DenseLayer:
    name: str
    kernel: np.ndarray          # (in, out) float32
    bias: np.ndarray            # (out,) float32


@dataclasses.dataclass
class UNetSpec:
    """The score UNet as plain arrays.

    ``convs`` holds the 11 Conv2D layers in forward order (conv2d … conv2d_10),
    ``dense`` the three time-embedding Dense layers (dense, dense_1 = embedding,
    dense_2 = projection added after the first bottleneck conv).
    """
    path: str
    sha256: str
    keras_version: str
    input_hw: Tuple[int, int]           # (H, W) the Functional model was built for
    channels: int                       # C of the state tensor (1 real-space, 2 spectral)
    base_filters: int
    batchnorm: bool
    convs: List[ConvLayer]
    dense: List[DenseLayer]
    n_params: int

    def pad_for(self, h: int, w: int) -> Tuple[int, int]:
        """Bottom/right zero padding to a multiple of 4 (…unet_2d_ising.py:29-34)."""
        return (4 - h % 4) % 4, (4 - w % 4) % 4


def _layer_index(config: dict) -> List[dict]:
    if config.get("class_name") != "Functional":
        raise ValueError("expected a Keras Functional model")
    return config["config"]["layers"]


def load_unet_spec(path: str) -> UNetSpec:
    raw = open(path, "rb").read()
    sha = hashlib.sha256(raw).hexdigest()
    with zipfile.ZipFile(io.BytesIO(raw)) as z:
        config = json.loads(z.read("config.json"))
        meta = json.loads(z.read("metadata.json"))
        tensors = read_datasets(z.read("model.weights.h5"))
    layers = _layer_index(config)
    by_class: Dict[str, List[dict]] = {}
    for l in layers:
        by_class.setdefault(l["class_name"], []).append(l)

    inputs = by_class.get("InputLayer", [])
    img = [l for l in inputs if len(l["config"]["batch_shape"]) == 4]
    tim = [l for l in inputs if len(l["config"]["batch_shape"]) == 2]
    if len(img) != 1 or len(tim) != 1:
        raise ValueError("expected one image input and one time input")
    _, h, w, c = img[0]["config"]["batch_shape"]

    convs_cfg = by_class.get("Conv2D", [])
    dense_cfg = by_class.get("Dense", [])
    bn_cfg = by_class.get("BatchNormalization", [])
    if len(convs_cfg) != 11 or len(dense_cfg) != 3 or len(bn_cfg) not in (0, 10):
        raise ValueError(
            f"not the reference UNet: {len(convs_cfg)} Conv2D, {len(dense_cfg)} Dense, {len(bn_cfg)} BatchNorm")
    for cls, n in (("AveragePooling2D", 2), ("UpSampling2D", 2), ("Concatenate", 2), ("Add", 1)):
        if len(by_class.get(cls, [])) != n:
            raise ValueError(f"not the reference UNet: expected {n} {cls} layers")
    for l in by_class["AveragePooling2D"]:
        if list(l["config"]["pool_size"]) != [2, 2] or l["config"]["padding"] != "valid":
            raise ValueError("AveragePooling2D must be 2x2 valid")
    for l in by_class["UpSampling2D"]:
        if list(l["config"]["size"]) != [2, 2] or l["config"].get("interpolation", "nearest") != "nearest":
            raise ValueError("UpSampling2D must be 2x nearest")

    def var(name: str, i: int) -> np.ndarray:
        key = f"/layers/{name}/vars/{i}"
        if key not in tensors:
            raise KeyError(f"{path}: missing {key}")
        return np.ascontiguousarray(tensors[key], dtype=np.float32)

    batchnorm = len(bn_cfg) == 10
    convs: List[ConvLayer] = []
    for i, l in enumerate(convs_cfg):
        cfg = l["config"]
        ks = list(cfg["kernel_size"])
        want = [1, 1] if i == 10 else [3, 3]
        if ks != want or cfg["padding"] != "same" or list(cfg.get("strides", [1, 1])) != [1, 1]:
            raise ValueError(f"{l['name']}: unexpected conv geometry")
        act = cfg["activation"]
        if act not in ("silu", "swish", "linear"):
            raise ValueError(f"{l['name']}: activation {act} not supported")
        silu = act in ("silu", "swish")
        layer = ConvLayer(l["name"], var(l["name"], 0), var(l["name"], 1), silu)
        if batchnorm and i < 10:
            b = bn_cfg[i]
            eps = np.float32(b["config"].get("epsilon", BN_EPSILON_DEFAULT))
            gamma, beta, mean, varr = (var(b["name"], k) for k in range(4))
            scale = (gamma / np.sqrt(varr + eps)).astype(np.float32)
            layer.bn_scale = scale
            layer.bn_shift = (beta - mean * scale).astype(np.float32)
            layer.silu = True  # Conv(linear) -> BN -> Activation("swish")
        convs.append(layer)

    f = convs[0].kernel.shape[3]
    expect = [(c, f), (f, f), (f, 2 * f), (2 * f, 2 * f), (2 * f, 4 * f), (4 * f, 4 * f),
              (6 * f, 2 * f), (2 * f, 2 * f), (3 * f, f), (f, f), (f, c)]
    for layer, (ci, co) in zip(convs, expect):
        if layer.kernel.shape[2:] != (ci, co):
            raise ValueError(f"{layer.name}: kernel {layer.kernel.shape} != (*,*,{ci},{co})")
    if convs[10].silu:
        raise ValueError("head conv must be linear")
    dense = [DenseLayer(l["name"], var(l["name"], 0), var(l["name"], 1)) for l in dense_cfg]
    if [d.kernel.shape for d in dense] != [(1, 4 * f), (4 * f, 4 * f), (4 * f, 4 * f)]:
        raise ValueError("unexpected time-embedding shapes")
    n_params = int(sum(t.size for t in tensors.values()))
    return UNetSpec(path, sha, meta.get("keras_version", "?"), (h, w), c, f, batchnorm, convs, dense, n_params)
