#!/usr/bin/env python3
"""bench.py — Mpixel-iterations/s of the fused CSV level-set update on MI355X.

A "step" is one pass of the hot path over the resident batch: one CSV iteration
(curvature + region terms + delta map + update + the c1/c2 sums) of every image this rank
holds.  N=1 runs BASELINE.json configs[1] (4096x4096, 1 channel, checkerboard init,
500 iterations, tol 0 so exactly K iterations execute); N>1 shards independent images, one
process per GPU, `--images-per-gpu` each (weak scaling: per-GPU work fixed), no data-path
collective — torch.distributed (RCCL) is used only for the barriers and the max-over-ranks
of the elapsed time.  Inputs (image planes, level set) are resident in HBM before the timed
region starts.

Prints ONE JSON line on rank 0 (see the driver's contract) with two extra objects:
  roofline      algorithmic bytes (2*8 + C bytes per pixel-iteration) per launch
                / average launch duration from HIP events on the kernel's stream
  cpu_baseline  the CPU oracle (reference-faithful pass structure, oracle/cv_oracle.c) timed
                on this host's cores on a bounded sample of the same workload (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)   # untimed; the first ~100 launches of a cold process run slower (clock ramp)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--channels", type=int, default=1, choices=[1, 3])
    ap.add_argument("--images-per-gpu", type=int, default=1)
    ap.add_argument("--math", default="default", choices=["default", "strict", "fast"])
    ap.add_argument("--finalize", type=int, default=0, choices=[0, 1])
    ap.add_argument("--opt", action="append", default=[], help="context option key=value (repeatable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=5)
    return ap.parse_args()


def main():
    args = parse()
    from chan_vese_amd import batch
    # torch (if any) is imported inside init_distributed, BEFORE the HIP library is loaded, so
    # that both share torch's bundled HIP runtime.  CHANVESE_DIST_BACKEND=gloo rehearses the
    # multi-rank path on one GPU (all ranks on device 0).
    dist, rank, world, local_rank = batch.init_distributed(os.environ.get("CHANVESE_DIST_BACKEND"))
    torch = None
    if dist is not None and dist.get_backend() == "nccl":
        import torch

    import numpy as np
    from chan_vese_amd import capi, synth

    n, C = args.size, args.channels
    device = local_rank % max(capi.device_count(), 1)
    math_mode = {"default": 0, "strict": 1, "fast": 2}[args.math]

    # ---- resident inputs: image b of this rank, checkerboard level set
    ctxs = []
    u0 = capi.checkerboard_host(n, n)
    for b in range(args.images_per_gpu):
        gb = rank * args.images_per_gpu + b
        if world == 1 and args.images_per_gpu == 1:
            planes = synth.config_planes("C2" if C == 1 else "C3", n)
        elif C == 1:
            planes = [synth.batch_image(gb, n)]
        else:
            planes = synth.config_planes("C3", n)
        p = capi.make_params(tol=0.0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1]) if C == 3 else capi.make_params(tol=0.0)
        ctx = capi.Context(n, n, C, p, device=device)
        ctx.set_option("math_mode", math_mode)
        ctx.set_option("finalize", args.finalize)
        for kv in args.opt:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        ctx.set_image(planes)
        ctx.set_levelset(u0)
        ctxs.append(ctx)

    def run_steps(k):
        # interleave the images' streams in chunks so their kernels overlap on the GPU
        chunk = 8 if len(ctxs) > 1 else 16   # 16 = one hipGraph of the library (api.hip, kGraphSteps)
        done = 0
        while done < k:
            c = min(chunk, k - done)
            for ctx in ctxs:
                ctx.enqueue_steps(c)
            done += c

    def sync_all():
        out = [ctx.sync() for ctx in ctxs]
        if torch is not None:
            torch.cuda.synchronize()
        return out

    def barrier():
        batch.barrier(dist)

    run_steps(args.warmup)
    sync_all()
    barrier()
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.steps)
    res = sync_all()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = [ctx.last_run_ms() for ctx in ctxs]  # HIP events on each image's stream
    assert all(r[0] == args.warmup + args.steps and not r[2] for r in res), res

    # end-of-run gather of the per-rank records (the only collective besides barriers/max)
    records = batch.gather_records(dist, [args.images_per_gpu, float(n) * n * args.steps * args.images_per_gpu, elapsed])
    elapsed = batch.max_over_ranks(dist, elapsed)
    total_images = world * args.images_per_gpu
    value = batch.aggregate_throughput(records, elapsed)

    out = None
    if rank == 0:
        bytes_per_launch = (2 * 8 + C) * float(n) * n      # read u, write u, read C planes
        avg_launch_s = (sum(kernel_ms) / len(kernel_ms)) / 1e3 / args.steps
        if args.images_per_gpu > 1:
            avg_launch_s /= args.images_per_gpu            # streams overlap: per-launch share of the span
        achieved = bytes_per_launch / avg_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"csv_step_{n}x{n}x{C}")
            except Exception:
                traffic = None
        workload = (f"{n}x{n} {C}-channel synthetic disk, checkerboard init, {args.steps} CSV iterations, tol 0"
                    + (" (BASELINE configs[1])" if (n, C, args.steps, total_images) == (4096, 1, 500, 1) else ""))
        kopt = dict(kv.split("=") for kv in (args.opt or [])).get("kernel", "-1")
        two_px = C == 1 and n % 16 == 0 and n >= 144 and (kopt == "3" or (kopt == "-1" and n * n <= 40000000))   # api.hip resolve_geometry
        kernel_name = {"0": "csv_step_kernel (tile)", "1": "csv_strip_kernel"}.get(kopt, "csv_wave2_kernel" if two_px else "csv_wave_kernel")
        out = {
            "metric": "Mpixel-iterations/s (CSV u-update)",
            "value": value,
            "unit": "Mpixel-iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "images_per_gpu": args.images_per_gpu,
                       "state": "fp64", "math": args.math, "parallelism": f"batch-shard x{world}",
                       "per_rank_mpx_it_s": [r[1] / r[2] / 1e6 for r in records]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel_name, "avg_launch_us": avg_launch_s * 1e6,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, C, args.cpu_iters)

    for ctx in ctxs:
        ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


def cpu_baseline(n, C, iters):
    """The CPU oracle (kind 'port': same pass structure and OpenMP team sizes as the
    reference, src/main.cpp:963-1001) on `iters` iterations of the same image."""
    import numpy as np
    from chan_vese_amd import synth
    from oracle import cv_oracle as O
    planes = synth.config_planes("C2" if C == 1 else "C3", n)
    u = O.checkerboard(n, n)
    p = O.make_params(tol=0.0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1]) if C == 3 else O.make_params(tol=0.0)
    O.csv_step(planes, u, p)  # warm-up (page faults, thread pool)
    t0 = time.perf_counter()
    for _ in range(iters):
        O.csv_step(planes, u, p)
    dt = time.perf_counter() - t0
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": float(n) * n * iters / dt / 1e6, "unit": "Mpixel-iterations/s", "cores": threads,
            "kind": "port",
            "sample": f"{iters} CSV iterations of the same {n}x{n}x{C} image after 1 warm-up iteration "
                      f"({dt:.1f} s); OpenMP teams as the reference: 3 threads in curvature(), "
                      f"{threads} for the delta map, c1/c2 sweeps serial"}


if __name__ == "__main__":
    main()
