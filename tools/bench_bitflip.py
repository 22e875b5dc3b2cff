#!/usr/bin/env python3
"""Throughput of the MNIST bit-flip sampler (float32 parity-mode kernels, then the bf16 convolution mode): B images through the
whole 100-step schedule.  Prints one JSON line per mode (not the repository's headline bench).
python tools/bench_bitflip.py [images] [fp32|bf16] [images per chunk, 0 = the library's default]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-model_amd"))
import numpy as np
import torch
from isingdiff import bitflip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0 = np.random.RandomState(0).randint(2, size=(B, 28, 28, 1)).astype(np.uint8)
flop = 199321600.0 * 100 * B     # conv MACs*2 of the improved UNet per forward x steps x images
only = sys.argv[2] if len(sys.argv) > 2 else None
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for precision, peak, label in (("fp32", 157.3e12, "frac_of_fp32_mfma_peak"), ("bf16", 2500e12, "frac_of_bf16_mfma_peak")):
    if only and precision != only:
        continue
    model = bitflip.BitflipModel(bitflip.synthetic_weights(0), device=0, time_step=100, precision=precision)
    bitflip.diffusion_backward(x0[:64], model, 100, return_trajectory=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = bitflip.diffusion_backward(x0, model, 100, return_trajectory=False, chunk=chunk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": f"mnist bit-flip sampler, 28x28, T=100, {precision} convolutions", "images": B, "chunk": chunk,
                      "images_per_s": B / dt, "tflops": flop / dt / 1e12, label: flop / dt / peak,
                      "mean_final_pixel": float(out.float().mean())}))
