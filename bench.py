#!/usr/bin/env python3
"""Benchmark of the reverse-diffusion samplers.  Default (the headline metric): denoised Ising lattice samples/s at
L=32 over the full 1000-step reverse schedule of the real-space UNet.

One "step" = one pass of the hot path over one batch: B units per GPU drawn from Philox noise, T reverse steps
through the score model (bf16 MFMA, float32 state), post-processing (sign projection / irfft2) and observables, and
(N > 1) one all-reduce of the observable histograms over RCCL.

    python bench.py --gpus 1 --steps K --warmup W [--workload realspace|spectral|bitflip|dense]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--workload   realspace  UNet of b_diffusion_model_unet_2d_ising.py, L x L lattices (--L 32 / 64 / 128)
             spectral   FFT-UNet of b_diffusion_model_fft_unet_2d_ising.py on the rfft2 spectrum, then irfft2 + sign
             bitflip    improved UNet + XOR sampler of mnist_diffusion/mnist_diffusion.py, 28 x 28, T = 100, synthetic weights
             dense      dense spectral model of b_diffusion_model_fft_2d_ising.py (n = 440 = 20 x 11 x 2)
--T k        runs a k-step schedule of ITS OWN: betas = float32(linspace(1e-4, 0.02, k)) and t / k as the time input.
             It is not a slice of the 1000-step schedule (state magnitudes differ); only --T 1000 (bitflip: 100) is the
             reference's workload, and the metric name says which one ran.

Prints ONE JSON line on rank 0.  `roofline` is for the kernel class with the largest share of GPU time, timed with
HIP events on the launch stream inside the timed region (bitflip / dense: all kernels of the sampler call as one
class); `cpu_baseline` is the PyTorch-CPU restatement of the Keras path (oracle/ref_torch.py) on a bounded sample,
rank 0, N=1 (UNet workloads).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "diffusion-model_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOP_PER_SITE_FORWARD = 221824.0      # 2*MACs of the 11 Conv2D layers per lattice site per forward (real-space UNet)
BITFLIP_FLOP_PER_FORWARD = 199321600.0  # 2*MACs of the improved UNet's convolutions per 28 x 28 image
PEAK_BF16_TFLOPS = 2500.0             # MI355X dense bf16 MFMA
PEAK_FP32_TFLOPS = 157.3              # MI355X fp32 MFMA
PEAK_HBM_GBS = 8000.0
# HBM bytes per lattice per launch of the dominant kernel at L=32 (x, skip1 and the conv7 map in, x out), from the
# rocprofv3 FETCH_SIZE / WRITE_SIZE passes kept under profiles/ (FETCH_SIZE doubled: gfx950 counts 16-byte-per-lane
# streaming reads at half their bytes).  Counters cannot be read inside this script; tools/measure_round.sh collects
# them with the same library.
PMC_HBM_BYTES_PER_LATTICE = {"fused_B(conv8,9,head,step)": 104.9 * 1024}
PMC_SOURCE = "profiles/r03/pmc_hbm.txt"


def spectral_flop_per_forward(L):
    """2*MACs of the FFT-UNet on the padded (L, pad4(L/2+1)) spectrum map (two state channels)."""
    hp, wp = L + (4 - L % 4) % 4, (L // 2 + 1) + (4 - (L // 2 + 1) % 4) % 4
    n1 = hp * wp
    macs = 9 * (2 * 32 + 32 * 32) * n1 + 9 * (32 * 64 + 64 * 64) * n1 / 4 + 9 * (64 * 128 + 128 * 128) * n1 / 16 \
        + 9 * (192 * 64 + 64 * 64) * n1 / 4 + 9 * (96 * 32 + 32 * 32) * n1 + 32 * 2 * n1
    return 2.0 * macs


def cpu_baseline(spec, shape, T, betas, flop_per_sample, budget_s=20.0):
    """Time the CPU restatement on a bounded sample of the same workload (same state shape, same weights, same update
    rule).  A short probe sizes the sample to about `budget_s` seconds: the whole schedule on as many lattices as fit,
    or, on a slow host, 32 lattices for part of the schedule (then scaled to whole-schedule lattices/s)."""
    from oracle.ref_torch import TorchUNet, diffusion_backward
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    model = TorchUNet(spec)
    rs = np.random.RandomState(42)
    B_p, T_probe = 32, 4
    xp = rs.standard_normal((B_p,) + shape).astype(np.float32)
    draw = lambda shp: (lambda t: rs.standard_normal(shp).astype(np.float32))
    diffusion_backward(model, xp, betas[-T_probe:], T_probe, draw(xp.shape))            # warm-up
    t0 = time.perf_counter()
    diffusion_backward(model, xp, betas[-T_probe:], T_probe, draw(xp.shape))
    per_lattice_step = (time.perf_counter() - t0) / (T_probe * B_p)
    if per_lattice_step * B_p * T <= budget_s:
        T_s = T
        B_s = int(min(1024, max(B_p, budget_s / (per_lattice_step * T)))) // 16 * 16
    else:
        B_s = B_p
        T_s = int(max(8, budget_s / (per_lattice_step * B_p)))
    x = rs.standard_normal((B_s,) + shape).astype(np.float32)
    t0 = time.perf_counter()
    diffusion_backward(model, x, betas[-T_s:], T_s, draw(x.shape))
    dt = time.perf_counter() - t0
    value = B_s / (dt * T / T_s)
    return {"value": value, "unit": "lattices/s", "cores": threads, "kind": "port",
            "sample": f"PyTorch-CPU fp32 restatement of the Keras sampler (oracle/ref_torch.py), B={B_s} lattices, "
                      f"state {'x'.join(str(v) for v in shape)}, {T_s} of {T} steps in {dt:.1f} s, scaled to the full schedule "
                      f"(BASELINE config 1 names L=20, B=256; the sample keeps this run's lattice size instead)",
            "gflops": value * flop_per_sample / 1e9}


def allreduce_i64(t):
    """Sum an int64 CUDA tensor over the ranks (RCCL; through host memory for a gloo rehearsal group)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    if dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        return h.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="realspace", choices=["realspace", "spectral", "bitflip", "dense"])
    ap.add_argument("--batch", type=int, default=65536, help="lattices / images / samples per GPU per step")
    ap.add_argument("--L", type=int, default=0, help="lattice side (realspace / spectral: default 32; dense: 20, fixed by the saved model)")
    ap.add_argument("--T", type=int, default=0, help="steps of the schedule (default: the reference's 1000; bitflip 100); see the module docstring")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel event timing")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=INT", help="library tuning switch (idm_set_option), e.g. strip_below_rows=0")
    ap.add_argument("--fp32-lattices", type=int, default=2048,
                    help="lattices of the float32 (bit-exact parity mode) companion measurement, realspace, N=1 only; 0 skips it")
    ap.add_argument("--dump-spins", default="", metavar="PREFIX",
                    help="write the last step's spins / final pixels of each rank to PREFIX.rank<r>.npy (rehearsal / tests)")
    args = ap.parse_args()

    if os.environ.get("ISINGDIFF_BENCH_TRACE"):      # debugging aid: dump every thread's stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ISINGDIFF_BENCH_TRACE"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # one process per GPU; ISINGDIFF_BENCH_BACKEND=gloo lets the N > 1 control flow be rehearsed on a box
    # with fewer GPUs than ranks (ranks then share devices round-robin; not a measurement)
    backend = os.environ.get("ISINGDIFF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from isingdiff import load_model, sample, beta_schedule, allreduce_observables, observe

    wl = args.workload
    B = args.batch
    first = rank * B
    T = args.T or (100 if wl == "bitflip" else 1000)
    full_T = T == (100 if wl == "bitflip" else 1000)
    peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
    model = None          # UNet workloads: the handle whose launches are profiled per kernel class
    last = {}

    if wl in ("realspace", "spectral"):
        L = args.L or 32
        fname = "b_diffusion_model_2d_ising_realspace_unet.keras" if wl == "realspace" else "b_diffusion_model_2d_ising_unet.keras"
        model = load_model(os.path.join(ROOT, "weights", fname), device=local_rank, precision=args.precision)
        for opt in args.option:
            name, _, val = opt.partition("=")
            model.set_option(name, int(val))
        betas = beta_schedule(T)
        flop_per_sample = (FLOP_PER_SITE_FORWARD * L * L if wl == "realspace" else spectral_flop_per_forward(L)) * T
        unit = "lattices/s"
        what = "real-space UNet" if wl == "realspace" else "FFT-UNet on the rfft2 spectrum (BatchNorm folded), irfft2 + sign"
        state_shape = (L, L, 1) if wl == "realspace" else (L, L // 2 + 1, 2)

        def one_step(i):
            res = sample(model, B, L, T=T, betas=betas, seed=42 + i, first_global_index=first, chunk=args.chunk)
            obs = allreduce_observables(res.observables)
            last["spins"] = res.spins
            return {"mean_E_per_site": obs.mean_energy_per_site(), "units_in_histogram": int(obs.moments[0]),
                    "lattices_in_histogram": int(obs.moments[0])}
    elif wl == "bitflip":
        from isingdiff import bitflip
        L = 28
        bmodel = bitflip.BitflipModel(bitflip.synthetic_weights(0), device=local_rank, time_step=T, precision=args.precision)
        flop_per_sample = BITFLIP_FLOP_PER_FORWARD * T
        unit = "images/s"
        what = "MNIST bit-flip UNet (synthetic Glorot weights, seed 0), XOR sampler"
        state_shape = (28, 28, 1)

        def one_step(i):
            # X_b: fair coin flips that depend on (global image index, pixel) only, so that the batch is the same however
            # it is sharded (a multiplicative hash of the index; the sampler's uniforms are Philox on the same index)
            g = torch.arange(first, first + B, dtype=torch.int64, device="cuda").view(B, 1)
            px = torch.arange(L * L, dtype=torch.int64, device="cuda").view(1, L * L)
            hsh = (g * 2654435761 + px * 40503 + (i + 7) * 97) & 0x7FFFFFFF
            x0 = (((hsh * 1103515245 + 12345) >> 16) & 1).to(torch.uint8).view(B, L, L, 1)
            out = bitflip.diffusion_backward(x0, bmodel, T, seed=42 + i, first_global_index=first, return_trajectory=False,
                                             chunk=args.chunk)
            count = out.view(B, -1).sum(1, dtype=torch.int64)
            hist = allreduce_i64(torch.bincount(count, minlength=L * L + 1))
            last["spins"] = out
            k = torch.arange(L * L + 1, device=hist.device, dtype=torch.float64)
            return {"mean_final_pixel": float((hist.double() * k).sum() / (hist.sum() * L * L)),
                    "units_in_histogram": int(hist.sum())}
    else:
        from isingdiff import dense, _lib
        L = 20
        dmodel = dense.load_dense_model(os.path.join(ROOT, "weights", "b_diffusion_model_2d_ising.keras"), device=local_rank,
                                        precision=args.precision)
        n = dmodel.n
        betas = beta_schedule(T)
        dims = [(k.shape[0], k.shape[1]) for k in dmodel.spec.kernels]
        flop_per_sample = 2.0 * sum(a * b for a, b in dims) * T
        unit = "samples/s"
        what = f"dense spectral score model (n = {n}), irfft2 + sign"
        state_shape = (n,)

        def one_step(i):
            x = torch.empty((B, n), dtype=torch.float32, device="cuda")
            _lib.check(_lib.load().idm_philox_normal(local_rank, x.data_ptr(), B, n, 42 + i, first, T,
                                                     torch.cuda.current_stream().cuda_stream))
            xf = dense.diffusion_backward(x, betas, dmodel, T, seed=42 + i, first_global_index=first, return_trajectory=False)
            spins = dense.spectral_to_spins(xf, L)
            obs = allreduce_observables(observe(spins))
            last["spins"] = spins
            return {"mean_E_per_site": obs.mean_energy_per_site(), "units_in_histogram": int(obs.moments[0])}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(-1 - i)
    barrier()
    profile = not args.no_profile
    ev = None
    if profile and model is not None:
        model.profile_begin()
    elif profile:
        # the library launches on torch's current stream: these events bracket every kernel of the timed calls
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    t0 = time.perf_counter()
    obs = None
    for i in range(args.steps):
        obs = one_step(i)
    if ev:
        ev[1].record()
    barrier()
    dt = time.perf_counter() - t0
    if profile and model is not None:
        prof = model.profile_end()
    elif profile:
        prof = [{"name": f"{wl} sampler (all kernels of the timed calls)", "launches": args.steps, "ms": ev[0].elapsed_time(ev[1]),
                 "flop": flop_per_sample * B * args.steps, "bytes": 0.0}]
    else:
        prof = []
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if args.dump_spins:
        np.save(f"{args.dump_spins}.rank{rank}.npy", last["spins"].cpu().numpy())
    fp32 = None
    if rank == 0 and world == 1 and wl == "realspace" and args.precision == "bf16" and args.fp32_lattices > 0:
        # companion record: the float32 parity mode (canonical-order kernels, bit-exact against the CPU oracle) on the
        # same lattice size and schedule, outside the timed region of the headline number
        nb = args.fp32_lattices
        sample(model, min(nb, 256), L, T=min(T, 20), betas=betas[:min(T, 20)], seed=1, precision="fp32")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sample(model, nb, L, T=T, betas=betas, seed=7, precision="fp32")
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t1
        fp32 = {"value": nb / d1, "unit": "lattices/s", "lattices": nb, "dtype": "fp32", "seconds": d1,
                "whole_job_mfma_frac": nb / d1 * flop_per_sample / (PEAK_FP32_TFLOPS * 1e12),
                "note": "v_mfma_f32_32x32x2_f32 kernels in canonical summation order (parity mode), one kernel per layer"}
    if rank == 0:
        total = world * B * args.steps
        value = total / dt
        roof = None
        kernels = []
        if prof:
            tot_ms = sum(e["ms"] for e in prof) or 1.0
            for e in prof:
                k = {"name": e["name"], "launches": e["launches"], "avg_us": 1e3 * e["ms"] / max(e["launches"], 1),
                     "share": e["ms"] / tot_ms}
                if e["flop"] > 0:
                    k["tflops"] = e["flop"] / (e["ms"] * 1e-3) / 1e12 if e["ms"] > 0 else 0.0
                k["gbs"] = e["bytes"] / (e["ms"] * 1e-3) / 1e9 if e["ms"] > 0 else 0.0
                kernels.append(k)
            dom = max(prof, key=lambda e: e["ms"])
            if dom["flop"] > 0:
                ach = dom["flop"] / (dom["ms"] * 1e-3) / 1e12
                per_lat = PMC_HBM_BYTES_PER_LATTICE.get(dom["name"]) if (wl, L) == ("realspace", 32) else None
                lat_per_launch = B * args.steps * 1.0 / (dom["launches"] / T) if model is not None else B
                roof = {"bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                        "frac": ach / peak, "traffic": per_lat * lat_per_launch if per_lat else None,
                        "traffic_unit": f"HBM bytes per launch (PMC, {PMC_SOURCE})" if per_lat else None,
                        "flop_per_launch": dom["flop"] / dom["launches"], "avg_launch_us": 1e3 * dom["ms"] / dom["launches"],
                        "share_of_gpu_time": dom["ms"] / tot_ms}
            else:
                ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
                roof = {"bound": "hbm", "kernel": dom["name"], "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": ach / PEAK_HBM_GBS, "traffic": None}
        sched = "full schedule" if full_T else f"{T}-step schedule of its own (not a slice of the reference's)"
        if wl == "realspace":
            metric = f"denoised Ising lattice samples/sec (whole node) at L={L}, {sched}"
        elif wl == "spectral":
            metric = f"denoised Ising lattice samples/sec (whole node), FFT-UNet on the spectrum, L={L}, {sched}"
        elif wl == "bitflip":
            metric = f"MNIST bit-flip diffusion images/sec (whole node), 28x28, {sched}"
        else:
            metric = f"dense spectral score model samples/sec (whole node), n={state_shape[0]}, {sched}"
        out = {
            "metric": metric,
            "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic (Philox N(0,1) x_T and per-step noise; saved .keras weights)" if wl != "bitflip"
            else "synthetic (hashed coin-flip X_b, Philox uniforms per step, Glorot-uniform weights seed 0: the reference ships none)",
            "config": {"workload": f"reverse sampling, {what}, L={L}, T={T}, {B} per GPU per step",
                       "L": L, "T": T, "lattices_per_gpu": B, "parallelism": f"batch-sharded x{world}"},
            "whole_job_mfma_frac": value * flop_per_sample / (world * peak * 1e12),
            "roofline": roof,
            "kernels": kernels,
            "fp32_companion": fp32,
            "observables": obs,
        }
        if world == 1 and not args.no_cpu_baseline and model is not None:
            out["cpu_baseline"] = cpu_baseline(model.spec, state_shape, T, betas, flop_per_sample)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
