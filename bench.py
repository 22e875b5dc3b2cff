#!/usr/bin/env python3
"""Benchmark: denoised Ising lattice samples/s at L=32 over the full 1000-step reverse schedule.

One "step" = one pass of the hot path over one batch: B lattices per GPU drawn from Philox noise,
T reverse steps through the UNet (bf16 MFMA convolutions, float32 state), sign projection,
observables and (N > 1) one RCCL all-reduce of the observable histograms.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is for the kernel class with the largest share of GPU
time, timed with HIP events on its own stream inside the timed region; `cpu_baseline` is the
PyTorch-CPU restatement of the Keras path (oracle/ref_torch.py) on a bounded sample, rank 0, N=1.
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

FLOP_PER_SITE_FORWARD = 221824.0      # 2*MACs of the 11 Conv2D layers per lattice site per forward
PEAK_BF16_TFLOPS = 2500.0             # MI355X dense bf16 MFMA
PEAK_FP32_TFLOPS = 157.3              # MI355X fp32 MFMA
PEAK_HBM_GBS = 8000.0
# HBM bytes per lattice per launch of the dominant kernel at L=32 (x, skip1 and the conv7 map in,
# x out), from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes kept in profiles/r02/pmc_hbm.txt
# (FETCH_SIZE doubled: gfx950 counts 16-byte-per-lane streaming reads at half their bytes).
# Counters cannot be read inside this script; tools/measure_round.sh collects them with the same library.
PMC_HBM_BYTES_PER_LATTICE = {"fused_B(conv8,9,head,step)": 104.1 * 1024}


def cpu_baseline(spec, L, T, betas, budget_s=20.0):
    """Time the CPU restatement on a bounded sample of the same workload (same L, same weights,
    same update rule).  A short probe sizes the sample to about `budget_s` seconds: the whole
    schedule on as many lattices as fit, or, on a slow host, 32 lattices for part of the schedule
    (then scaled to whole-schedule lattices/s)."""
    from oracle.ref_torch import TorchUNet, diffusion_backward
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    model = TorchUNet(spec)
    rs = np.random.RandomState(42)
    B_p, T_probe = 32, 4
    xp = rs.standard_normal((B_p, L, L, spec.channels)).astype(np.float32)
    draw = lambda shape: (lambda t: rs.standard_normal(shape).astype(np.float32))
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
    x = rs.standard_normal((B_s, L, L, spec.channels)).astype(np.float32)
    t0 = time.perf_counter()
    diffusion_backward(model, x, betas[-T_s:], T_s, draw(x.shape))
    dt = time.perf_counter() - t0
    value = B_s / (dt * T / T_s)
    return {"value": value, "unit": "lattices/s", "cores": threads, "kind": "port",
            "sample": f"PyTorch-CPU fp32 restatement of the Keras sampler (oracle/ref_torch.py), B={B_s} lattices, "
                      f"L={L}, {T_s} of {T} steps in {dt:.1f} s, scaled to the full schedule",
            "gflops": value * FLOP_PER_SITE_FORWARD * L * L * T / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="lattices per GPU per step")
    ap.add_argument("--L", type=int, default=32)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel event timing")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=INT", help="library tuning switch (idm_set_option), e.g. strip_below_rows=0")
    ap.add_argument("--fp32-lattices", type=int, default=2048,
                    help="lattices of the float32 (bit-exact parity mode) companion measurement, N=1 only; 0 skips it")
    ap.add_argument("--dump-spins", default="", metavar="PREFIX",
                    help="write the last step's spins of each rank to PREFIX.rank<r>.npy (rehearsal / tests)")
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

    from isingdiff import load_model, sample, beta_schedule, allreduce_observables

    path = os.path.join(ROOT, "weights", "b_diffusion_model_2d_ising_realspace_unet.keras")
    model = load_model(path, device=local_rank, precision=args.precision)
    for opt in args.option:
        name, _, val = opt.partition("=")
        model.set_option(name, int(val))
    betas = beta_schedule(args.T)
    B, L, T = args.batch, args.L, args.T
    first = rank * B

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    last = {}

    def one_step(i):
        res = sample(model, B, L, T=T, betas=betas, seed=42 + i, first_global_index=first, chunk=args.chunk)
        obs = allreduce_observables(res.observables)
        last["spins"] = res.spins
        return obs

    for i in range(args.warmup):
        one_step(-1 - i)
    barrier()
    profile = not args.no_profile
    if profile:
        model.profile_begin()
    t0 = time.perf_counter()
    obs = None
    for i in range(args.steps):
        obs = one_step(i)
    barrier()
    dt = time.perf_counter() - t0
    prof = model.profile_end() if profile else []
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if args.dump_spins:
        np.save(f"{args.dump_spins}.rank{rank}.npy", last["spins"].cpu().numpy())
    fp32 = None
    if rank == 0 and world == 1 and args.precision == "bf16" and args.fp32_lattices > 0:
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
                "whole_job_mfma_frac": nb / d1 * FLOP_PER_SITE_FORWARD * L * L * T / (PEAK_FP32_TFLOPS * 1e12),
                "note": "v_mfma_f32_32x32x2_f32 kernels in canonical summation order (parity mode), one kernel per layer"}
    if rank == 0:
        total = world * B * args.steps
        value = total / dt
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
        flop_per_sample = FLOP_PER_SITE_FORWARD * L * L * T
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
                per_lat = PMC_HBM_BYTES_PER_LATTICE.get(dom["name"]) if L == 32 else None
                lat_per_launch = B * args.steps * 1.0 / (dom["launches"] / T)
                roof = {"bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                        "frac": ach / peak, "traffic": per_lat * lat_per_launch if per_lat else None,
                        "traffic_unit": "HBM bytes per launch (PMC, profiles/r02/pmc_hbm.txt)",
                        "flop_per_launch": dom["flop"] / dom["launches"], "avg_launch_us": 1e3 * dom["ms"] / dom["launches"],
                        "share_of_gpu_time": dom["ms"] / tot_ms}
            else:
                ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
                roof = {"bound": "hbm", "kernel": dom["name"], "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": ach / PEAK_HBM_GBS, "traffic": None}
        out = {
            "metric": "denoised Ising lattice samples/sec (whole node) at L=32, full schedule" if (L, T) == (32, 1000)
            else f"denoised Ising lattice samples/sec (whole node) at L={L}, {T}-step slice",
            "value": value, "unit": "lattices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic (Philox N(0,1) x_T and per-step noise; saved .keras weights)",
            "config": {"workload": f"reverse sampling, real-space UNet, L={L}, T={T}, {B} lattices per GPU per step",
                       "L": L, "T": T, "lattices_per_gpu": B, "parallelism": f"batch-sharded x{world}"},
            "whole_job_mfma_frac": value * flop_per_sample / (world * peak * 1e12),
            "roofline": roof,
            "kernels": kernels,
            "fp32_companion": fp32,
            "observables": {"mean_E_per_site": obs.mean_energy_per_site() if obs is not None else None,
                            "lattices_in_histogram": int(obs.moments[0]) if obs is not None else 0},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model.spec, L, T, betas)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
