#!/usr/bin/env python3
"""bench.py — Mpixel-iterations/s of the fused CSV level-set update on MI355X.

A "step" is one pass of the hot path over the resident batch: one CSV iteration (curvature +
region terms + delta map + update + the c1/c2 sums) of every image this rank holds.

Workloads (`--config`, BASELINE.json `configs`):
  C2 (default at N=1)  4096x4096, 1 channel, clean disk, checkerboard init, 500 iterations, tol 0
  C3                   4096x4096, 3 channels (per-channel lambda), 300 iterations
  C4                   2048x2048 noisy disk: Perona-Malik 1000 steps (K=30, L=0.25, T=250) timed as its
                       own phase (`pm` object, 16 B per plane-pixel-step), then 200 CSV iterations (the plane fits
                       the chip's LDS: the library runs them in its resident flow, one cooperative launch per chunk)
  C5 (default at N>1)  the per-GPU share of the 64-image batch: 8 noisy 4096x4096 images per GPU
                       (images 8r..8r+7 on rank r), interleaved on their own streams
  C4-image, C5-image   one C4 / C5 image (noise 32 at 2048^2 / noise 16 at 4096^2), CSV only
  near                 C2's image with dt = 0.001 (the reference README's second example run): |u| < 32 eps everywhere for the
                       whole run, every pixel takes the table form of H_eps; checked by that very property
`--gpus N` (N>1) without a launcher (WORLD_SIZE unset) starts N child ranks itself — fresh processes,
one per GPU, before this process touches the GPU — and relays rank 0's line.  Under
`python -m torch.distributed.run` the ranks are used as launched.  There is no data-path collective:
torch.distributed (RCCL) carries the barriers, the max-over-ranks of the elapsed time and the end-of-run
gather of per-rank records.  Inputs (image planes, level set) are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (the driver's contract) with extra objects:
  roofline      algorithmic bytes ((2*8 + C) per pixel-iteration) per launch / average launch
                duration from HIP events on the kernel's stream
  cpu_baseline  the CPU oracle (reference-faithful pass structure, oracle/cv_oracle.c) on this host's
                cores, bounded sample of the same workload (rank 0, N=1)
  phases        (N=1) us per iteration over iterations 1-16 / 17-100 / 101-500 of a fresh run from the
                checkerboard, one sync per segment, measured after the timed region
  pm            (C4) the Perona-Malik phase with its own roofline
  checked       true once every resident image's result passed verify_result() AFTER the timed region (outside it): the
                level set is finite, the mask is the synthetic disk (or its complement), c1/c2 sit at the disk's levels
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
MAX_CLOCK_GHZ = 2.4    # same guide: one vector instruction issues per SIMD per cycle at best


def resident_roofline(hbm, kind, n, C, seconds_per_unit, tj):
    """Roofline of a RESIDENT kernel (csv_resident_kernel / pm_resident_kernel): the plane lives in the LDS of the CUs between the
    tile load and the tile store of a launch, so HBM is not what bounds it -- its counters show the FP64 vector pipe as the busiest
    unit.  bound = "valu_fp64": achieved = VALU-issue cycles per SIMD per iteration (SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / iterations
    of the committed counter pass, profiles/traffic.json) per second of measured time; peak = the 2.4 GHz a SIMD can issue at.
    The 17 B/px (16 B/px.step) figure over the same time is kept as `hbm_equivalent`, labelled: it may exceed the HBM peak because
    those bytes never move."""
    key = f"{kind}_resident_{n}x{n}x{C}_valu_cycles_per_{'iteration' if kind == 'csv' else 'step'}"
    cyc = (tj or {}).get(key)
    ach = None if cyc is None else cyc / seconds_per_unit / 1e9
    out = {"bound": "valu_fp64", "achieved": ach, "peak": MAX_CLOCK_GHZ, "unit": "Gcycle/s of vector-instruction issue per SIMD",
           "frac": None if ach is None else ach / MAX_CLOCK_GHZ,
           "valu_issue_cycles_per_simd": cyc, "valu_source": (tj or {}).get("_valu_source"),
           "traffic": hbm.get("traffic"), "traffic_source": hbm.get("traffic_source")}
    for k in ("kernel", "launch_info", "avg_launch_us", "avg_launch_us_wall", "steps_per_launch"):
        if k in hbm:
            out[k] = hbm[k]
    out["hbm_equivalent"] = {"what": "algorithmic HBM bytes (SURVEY.md 8d) over the measured time -- NOT this kernel's bound: the state stays in LDS",
                             "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac_of_hbm_peak": hbm["frac"],
                             "frac_of_hbm_peak_wall": hbm.get("frac_wall"),
                             "algorithmic_bytes_per_launch": hbm.get("algorithmic_bytes_per_launch")}
    return out

CONFIGS = {
    #            n     C  images  steps  description
    "C2":       (4096, 1, 1, 500, "clean disk (BASELINE configs[1])"),
    "C3":       (4096, 3, 1, 300, "3-channel disks, lambda1 1 1 0.5, lambda2 1 0.5 1 (BASELINE configs[2])"),
    "C4":       (2048, 1, 1, 200, "noisy disk (noise 32, seed 1) after Perona-Malik 1000 steps K=30 L=0.25 (BASELINE configs[3])"),
    "C5":       (4096, 1, 8, 500, "per-GPU share of the 64-image batch: noisy disks (noise 16, seed 1000+b) (BASELINE configs[4])"),
    "C4-image": (2048, 1, 1, 200, "noisy disk (noise 32, seed 1), no Perona-Malik"),
    "C5-image": (4096, 1, 1, 500, "noisy disk (noise 16, seed 1000), image 0 of the batch"),
    # the regime of the reference README's second example (README.md:58-63, --dt 0.001): the level set stays below the far-field
    # threshold of H_eps (32 eps) everywhere for the whole run, so every pixel takes the table form of H_eps in every iteration
    "near":     (4096, 1, 1, 500, "clean disk, dt = 0.001: every pixel stays in the near field of H_eps (|u| < 32 eps) for the whole run"),
    # DECLARED FP32-state mode (option "state" = 32; SURVEY.md 8d: 9 / 11 algorithmic bytes per pixel-iteration): never the default
    "C2-f32":   (4096, 1, 1, 500, "clean disk (BASELINE configs[1]) with the level set stored as FLOAT in HBM (declared mode: FP64 arithmetic and sums, every new value rounded to float)"),
    "C3-f32":   (4096, 3, 1, 300, "3-channel disks, lambda1 1 1 0.5, lambda2 1 0.5 1 (BASELINE configs[2]) with the level set stored as FLOAT in HBM (declared mode)"),
}
CONFIG_DT = {"near": 0.001}
CONFIG_STATE = {"C2-f32": 32, "C3-f32": 32}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed iterations (default: the config's count)")
    ap.add_argument("--warmup", type=int, default=100)   # untimed iterations on the same level set
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS))
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--channels", type=int, default=None, choices=[1, 3])
    ap.add_argument("--images-per-gpu", type=int, default=None)
    ap.add_argument("--math", default="fast", choices=["fast", "strict"])
    ap.add_argument("--finalize", type=int, default=0, choices=[0, 1])
    ap.add_argument("--opt", action="append", default=[], help="context option key=value (repeatable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-phases", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--pm-steps", type=int, default=1000, help="C4: Perona-Malik steps (T = steps * 0.25)")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="device clock warm-up before the W warm-up steps: this many ms of the same kernel on a SCRATCH "
                         "context (own buffers; the measured level sets are not touched). 0 = off")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """`bench.py --gpus N` started directly: N child ranks, one per GPU, created BEFORE this process makes
    any HIP / torch.cuda call (the parent never touches the GPU and never re-execs itself).  Rank 0's
    stdout (the JSON line) is relayed; the exit code is the worst child's."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def resolve_workload(args, world):
    name = args.config or ("C2" if world == 1 else "C5")
    n, C, images, steps, desc = CONFIGS[name]
    n = args.size or n
    C = args.channels or C
    if args.channels == 3 and args.config is None:
        name, desc = "C3", CONFIGS["C3"][4]
    images = args.images_per_gpu or images
    steps = steps if args.steps is None else args.steps
    return name, n, C, images, steps, desc


def expected_disk(name, n, gb):
    """(boolean disk of image gb, [(fg, bg) per channel], noise amplitude) of the synthetic inputs (chan_vese_amd/synth.py)."""
    import numpy as np
    r = n // 4 + 8 * (gb % 8) - 28 if name in ("C5", "C5-image") else n // 4
    ii = np.arange(n, dtype=np.int64)[:, None] - n // 2
    jj = np.arange(n, dtype=np.int64)[None, :] - n // 2
    disk = ii * ii + jj * jj <= r * r
    levels = [(180, 40), (200, 60), (60, 200)] if name in ("C3", "C3-f32") else [(200, 50)]
    noise = {"C4": 32, "C4-image": 32, "C5": 16, "C5-image": 16}.get(name, 0)
    return disk, levels, noise


def verify_result(name, n, gb, ctx, iterations):
    """Property check of one image's result (no oracle: it cannot follow 500 iterations at 4096^2 in minutes).  The
    Chan-Vese fixed point of a two-level disk image is the disk: mask == disk or its complement (which side is 'inside'
    is decided by the sign of c1 - c2 after the first iteration of the symmetric checkerboard start), region means at the
    disk's two levels, every level-set value finite.  Needs enough iterations for the checkerboard to dissolve."""
    import numpy as np
    disk, levels, noise = expected_disk(name, n, gb)
    u = ctx.get_levelset()
    mask = ctx.get_mask().astype(bool)
    c1, c2 = ctx.get_means()
    finite = bool(np.isfinite(u).all())
    if name == "near":
        # dt = 0.001: the contour has not settled on the disk yet and is not meant to -- the claim of this config is the REGIME: every
        # pixel below the far-field threshold of H_eps (32 eps, eps = 1) after all W + K iterations, and (|u| only grows) before;
        # plus what any run must satisfy: finite, both regions populated, means between the image's two levels and c1 != c2
        umax = float(np.abs(u).max())
        lo, hi = min(levels[0]) - 1e-9, max(levels[0]) + 1e-9
        ok = finite and umax < 32.0 and 0.0 < mask.mean() < 1.0 and lo <= c1[0] <= hi and lo <= c2[0] <= hi and abs(c1[0] - c2[0]) > 1.0
        return ok, {"image": gb, "finite": finite, "max_abs_u": umax, "share_below_far_threshold": float((np.abs(u) < 32.0).mean()),
                    "inside_share": float(mask.mean()), "c1": float(c1[0]), "c2": float(c2[0])}
    inter, union = (mask & disk).sum(), (mask | disk).sum()
    iou_d = inter / max(union, 1)
    inter_c, union_c = (mask & ~disk).sum(), (mask | ~disk).sum()
    iou_c = inter_c / max(union_c, 1)
    inside_is_disk = iou_d >= iou_c
    iou = float(max(iou_d, iou_c))
    # c1 = mean over the inside (H ~ 1): the disk's foreground if the inside is the disk, else the background; the smoothed
    # Heaviside leaks ~1/(pi |u|) of the other region into each mean, the noise is zero-mean up to clamping
    tol_c = 4.0 + 0.05 * noise      # oracle, 512^2 / noisy 2048^2: 2.7 / 3.4 after 5 iterations, 2.1 after 25, 1.6 after 64
    dev = 0.0
    for k, (fg, bg) in enumerate(levels):
        want1, want2 = (fg, bg) if inside_is_disk else (bg, fg)
        dev = max(dev, abs(c1[k] - want1), abs(c2[k] - want2))
    settled = iterations >= 5       # oracle: the mask IS the disk from iteration 5 on (512^2 clean, 2048^2 noise 32)
    ok = finite and (not settled or (iou >= 0.999 and dev <= tol_c))
    return ok, {"image": gb, "finite": finite, "mask_iou_vs_disk": iou, "inside_is_disk": bool(inside_is_disk),
                "max_abs_c_minus_level": float(dev)}


def image_planes(name, n, gb):
    from chan_vese_amd import synth
    if name in ("C3", "C3-f32"):
        return synth.config_planes("C3", n)
    if name in ("C4", "C4-image"):
        return synth.config_planes("C4", n)
    if name in ("C5", "C5-image"):
        return [synth.batch_image(gb, n)]
    return synth.config_planes("C2", n)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    from chan_vese_amd import batch
    # torch (if any) is imported inside init_distributed, BEFORE the HIP library is loaded, so that both
    # share torch's bundled HIP runtime.  CHANVESE_DIST_BACKEND=gloo rehearses the multi-rank path on
    # one GPU (all ranks on device 0); CHANVESE_BENCH_DRYRUN=1 additionally skips the GPU work (CPU test
    # of the launch / gather / report path; its line says "data": "dry-run" and measures nothing).
    dist, rank, world, local_rank = batch.init_distributed(os.environ.get("CHANVESE_DIST_BACKEND"))
    dry = os.environ.get("CHANVESE_BENCH_DRYRUN") == "1"
    torch = None
    if dist is not None and dist.get_backend() == "nccl":
        import torch

    name, n, C, images, steps, desc = resolve_workload(args, world)
    math_mode = {"strict": 1, "fast": 2}[args.math]
    state_bits = CONFIG_STATE.get(name, 64)
    pm_info = None
    phases = None
    ctxs = []
    kernel_ms = [0.0]

    if not dry:
        from chan_vese_amd import capi
        device = local_rank % max(capi.device_count(), 1)
        for b in range(images):
            gb = rank * images + b
            planes = image_planes(name, n, gb)
            dt = CONFIG_DT.get(name, 1.0)
            p = capi.make_params(tol=0.0, dt=dt, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1]) if C == 3 else capi.make_params(tol=0.0, dt=dt)
            ctx = capi.Context(n, n, C, p, device=device)
            ctx.set_option("math_mode", math_mode)
            ctx.set_option("finalize", args.finalize)
            ctx.set_option("state", state_bits)
            # (several images on one GPU: their level sets -- 8 x 272 MiB -- do not live in the Infinity Cache, where write-through stores
            # cost.  Rounds 2-3 set "wave_pol" = 0 here by hand; since round 4 the library's automatic choice sees every live context.)
            for kv in args.opt:
                k, v = kv.split("=")
                ctx.set_option(k, int(v))
            ctx.set_image(planes)
            if name == "C4":   # the pre-smoother is its own timed phase (src/main.cpp:940-947 runs it once, before the loop)
                L_, T_ = 0.25, args.pm_steps * 0.25
                # device warm-up as for the CSV phase (a cold GPU runs its first ~100 ms of launches 8-10 % slower): the same PM
                # kernel on the same planes for args.prewarm_ms (at least one pass of <= 100 steps), then the planes are restored
                t_pw = time.perf_counter()
                while args.prewarm_ms > 0:       # (--prewarm-ms 0: ONE launch of the configured length and nothing else -- the profiling pass)
                    ctx.perona_malik(30.0, L_, min(T_, 25.0))
                    if (time.perf_counter() - t_pw) * 1e3 >= args.prewarm_ms:
                        break
                ctx.set_image(planes)
                ctx.perona_malik(30.0, L_, T_)
                pm_ms = ctx.last_pm_ms()
                trips = capi.pm_trip_count(L_, T_)
                pm_bytes = 16.0 * n * n * C                      # read + write the FP64 state of every plane per step
                pm_info = {"steps": trips, "K": 30.0, "L": L_, "T": T_, "us_per_step": pm_ms * 1e3 / max(trips, 1),
                           "value": float(n) * n * C * trips / (pm_ms / 1e3) / 1e6, "unit": "Mplane-pixel-steps/s",
                           "roofline": {"bound": "hbm", "achieved": pm_bytes * trips / (pm_ms / 1e3) / 1e9,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": pm_bytes * trips / (pm_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                                        "algorithmic_bytes_per_launch": pm_bytes, "traffic": None,
                                        "kernel": None, "steps_per_launch": None}}   # filled from cvh_launch_info below
            ctx.init_checkerboard()
            ctxs.append(ctx)

    def run_steps(k):
        if len(ctxs) == 1:       # one image: the library issues k % 16 plain launches, then hipGraphs of 16 steps (api.hip)
            ctxs[0].enqueue_steps(k)
            return
        done = 0                 # several images: interleave their streams in chunks of 8 so that their kernels overlap
        while done < k:
            c = min(8, k - done)
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

    # Device warm-up, not part of W or K: a cold process runs its first ~100 launches 8-10 % slower (DVFS ramp from the
    # idle clock; profiles/README.md), whatever they compute.  The same kernel runs on a scratch context with its own
    # buffers, so the level sets being measured still see exactly W warm-up + K timed iterations.
    prewarm_launches = 0
    scratch = None

    def prewarm():
        nonlocal scratch
        if not ctxs or args.prewarm_ms <= 0:
            return 0
        if scratch is None:
            scratch = capi.Context(n, n, C, capi.make_params(tol=0.0, dt=CONFIG_DT.get(name, 1.0)), device=device)
            scratch.set_option("math_mode", math_mode)
            scratch.set_option("co_resident", 0)      # a warm-up context: idle while the measured contexts stream (not part of the device's live footprint)
            scratch.set_option("state", state_bits)
            for kv in args.opt:
                k, v = kv.split("=")
                scratch.set_option(k, int(v))
            scratch.set_image(image_planes(name, n, rank * images))
            scratch.init_checkerboard()
        launches = 0
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
            scratch.enqueue_steps(64)
            scratch.sync()
            launches += 64
        return launches

    run_steps(args.warmup)
    sync_all()
    for ctx in ctxs:
        ctx.warm(steps if len(ctxs) == 1 else 8)   # hipGraph capture/instantiate is one-off host work: not a step
    # the device warm-up comes LAST, directly before the timed region (the graph build above leaves the GPU idle for a
    # millisecond or two).  Short runs stay exposed to the box: over five boxes the 20-step line read 57.9-59.6 us on four
    # and, in some processes of one, 64-73 us with either order of these steps
    prewarm_launches = prewarm()
    barrier()
    sync_all()
    t0 = time.perf_counter()
    run_steps(steps)
    res = sync_all()
    barrier()
    t1 = time.perf_counter()
    elapsed = (t1 - t0) if not dry else 1.0 + 0.25 * rank
    if not dry:
        kernel_ms = [ctx.last_run_ms() for ctx in ctxs]  # HIP events on each image's stream
        assert all(r[0] == args.warmup + steps and not r[2] for r in res), res

    # result check, outside the timed region: every resident image (the 8 interleaved C5 images included)
    checked, check_info, launch, pm_launch = None, [], None, None
    if not dry:
        checked = True
        for b, ctx in enumerate(ctxs):
            ok, info = verify_result(name, n, rank * images + b, ctx, args.warmup + steps)
            check_info.append(info)
            checked = checked and ok
        if not checked:     # reported after the collectives below (a rank that left early would hang the others)
            print(f"bench.py: result check FAILED on rank {rank}: {check_info}", file=sys.stderr, flush=True)
        launch = ctxs[0].launch_info(0)
        pm_launch = ctxs[0].launch_info(1) if pm_info is not None else None

    # end-of-run gather of the per-rank records (the only collective besides barriers/max)
    records = batch.gather_records(dist, [images, float(n) * n * steps * images, elapsed, 0.0 if checked is False else 1.0])
    elapsed = batch.max_over_ranks(dist, elapsed)
    value = batch.aggregate_throughput(records, elapsed)
    all_checked = all(r[3] == 1.0 for r in records)

    out = None
    if rank == 0:
        bytes_per_launch = (2 * (state_bits // 8) + C) * float(n) * n      # read u, write u, read C planes (FP32 state: 9 / 11 B)
        avg_launch_s = max((sum(kernel_ms) / len(kernel_ms)) / 1e3 / max(steps, 1), 1e-12)
        if images > 1:
            avg_launch_s /= images                         # streams overlap: per-launch share of the span
        achieved = bytes_per_launch / avg_launch_s / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        tj = None
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # the resident kernel runs a chunk of iterations per launch: its counters are recorded per ITERATION (launch total / iterations)
                traffic = tj.get(f"csv_resident_{n}x{n}x{C}_per_iteration") if (launch or {}).get("kernel", "").startswith("csv_resident_kernel") else tj.get(f"csv_step_{n}x{n}x{C}")
                traffic_source = tj.get("_source") if traffic is not None else None
                if pm_info is not None and pm_launch is not None:
                    if pm_launch["kernel"].startswith("pm_resident_kernel"):      # one launch for all steps: counters recorded per STEP
                        pm_info["roofline"]["traffic"] = tj.get(f"pm_resident_{n}x{n}x{C}_per_step")
                    elif int(pm_launch["steps_per_launch"]) == 2 and tj.get(f"pm_2steps_{n}x{n}x{C}") is not None:
                        pm_info["roofline"]["traffic"] = tj[f"pm_2steps_{n}x{n}x{C}"] / 2.0   # per time step (a launch makes two)
                    if pm_info["roofline"]["traffic"] is not None:
                        pm_info["roofline"]["traffic_source"] = tj.get("_source")
            except Exception:
                traffic = None
        kernel_name = launch["kernel"] if not dry else "none (dry run)"     # what the library says it launches (cvh_launch_info)
        # the same bytes over the WALL clock `value` is computed from (max over ranks, all images of a rank): frac_wall;
        # over the HIP events on the kernel's stream: frac_events (= frac, the contract's definition)
        wall_launch_s = max(elapsed / max(steps, 1) / images, 1e-12)
        frac_wall = bytes_per_launch / wall_launch_s / 1e9 / HBM_PEAK_GBS
        if pm_info is not None and pm_launch is not None:
            pm_info["roofline"]["kernel"] = pm_launch["kernel"]
            pm_info["roofline"]["steps_per_launch"] = int(pm_launch["steps_per_launch"])
            pm_info["roofline"]["launch_info"] = pm_launch
        workload = f"{name}: {n}x{n} {C}-channel {desc}, checkerboard init, {steps} CSV iterations after {args.warmup} warm-up, tol 0"
        out = {
            "metric": "Mpixel-iterations/s (CSV u-update)",
            "value": value,
            "unit": "Mpixel-iterations/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / max(steps, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if state_bits == 64 else "f32",
            "data": "synthetic" if not dry else "dry-run",
            "config": {"workload": workload, "images_per_gpu": images, "images_total": images * world,
                       "state": "fp64" if state_bits == 64 else "fp32 in HBM (DECLARED mode: every new level-set value rounded to float; arithmetic, tables and sums FP64 / 64-bit fixed point)",
                       "math": args.math,
                       "device_prewarm": f"{prewarm_launches} launches on a scratch context between the warm-up steps and the timed steps", "parallelism": f"batch-shard x{world}",
                       "ranks_in_group": (dist.get_world_size() if dist is not None else 1),
                       "backend": (dist.get_backend() if dist is not None else "none"),
                       "per_rank_mpx_it_s": [r[1] / r[2] / 1e6 for r in records]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_clock": "hip_events",
                         "frac_events": achieved / HBM_PEAK_GBS, "frac_wall": frac_wall,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name, "launch_info": (launch if not dry else None), "avg_launch_us": avg_launch_s * 1e6,
                         "avg_launch_us_wall": wall_launch_s * 1e6,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
            "checked": (all_checked if not dry else None),
            "check": {"what": "after the timed region: level set finite, mask IoU vs the synthetic disk (or its complement) >= 0.999, "
                              "|c1/c2 - disk level| small; every resident image", "images": check_info},
        }
        if pm_info is not None:
            out["pm"] = pm_info
        # resident kernels: HBM does not bound them (ADVICE r3 / VERDICT r3 item 6) -- report the bound that does, keep the HBM figure labelled
        if not dry and kernel_name.startswith("csv_resident_kernel"):
            out["roofline"] = resident_roofline(out["roofline"], "csv", n, C, avg_launch_s, tj)
        if pm_info is not None and (pm_info["roofline"].get("kernel") or "").startswith("pm_resident_kernel"):
            pm_info["roofline"] = resident_roofline(pm_info["roofline"], "pm", n, C, pm_info["us_per_step"] * 1e-6, tj)

    # (N=1) the same run split by phase: iterations 1-16 (every pixel of the checkerboard is in the near
    # field of H_eps), 17-100, 101-500; one sync per segment, nothing else in the timed spans
    if world == 1 and not dry and not args.no_phases and images == 1:
        ctx = ctxs[0]
        ctx.init_checkerboard()          # on the device: no 134 MB upload before the first segment
        phases = {}
        lo = 1
        for seg in (16, 84, 400):
            ctx.warm(seg)
            if lo == 1:
                prewarm()                # same clock state as the timed region, no host work between it and the first segment
            ctx.enqueue_steps(seg)
            ctx.sync()
            phases[f"{lo}-{lo + seg - 1}"] = ctx.last_run_ms() * 1e3 / seg
            lo += seg
        out["phases"] = {"unit": "us per iteration (HIP events), fresh run from the checkerboard after the same device prewarm", **phases}

    if scratch is not None:
        scratch.close()
    for ctx in ctxs:
        ctx.close()
    if out is not None and world == 1 and not dry and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(name, n, C, args.cpu_iters)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not all_checked:
        sys.exit("bench.py: a result check failed (see stderr of the rank): no line is printed for a wrong result")
    if out is not None:
        print(json.dumps(out), flush=True)


def cpu_baseline(name, n, C, iters):
    """The CPU oracle (kind 'port': same pass structure and OpenMP team sizes as the reference,
    src/main.cpp:963-1001 and :478-560) on a bounded sample of the same workload."""
    from oracle import cv_oracle as O
    planes = image_planes(name, n, 0)
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    extra = {}
    if name == "C4":   # the pre-smoother: a few steps of the same plane (serial per channel, as the reference)
        pm_steps = 8
        t0 = time.perf_counter()
        O.perona_malik(planes, 30.0, 0.25, pm_steps * 0.25)
        dt = time.perf_counter() - t0
        extra["pm"] = {"value": float(n) * n * C * pm_steps / dt / 1e6, "unit": "Mplane-pixel-steps/s",
                       "sample": f"{pm_steps} Perona-Malik steps of the {n}x{n} plane ({dt:.1f} s), serial per channel as the reference"}
    u = O.checkerboard(n, n)
    p = O.make_params(tol=0.0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1]) if C == 3 else O.make_params(tol=0.0)
    O.csv_step(planes, u, p)  # warm-up (page faults, thread pool)
    t0 = time.perf_counter()
    for _ in range(iters):
        O.csv_step(planes, u, p)
    dt = time.perf_counter() - t0
    return {"value": float(n) * n * iters / dt / 1e6, "unit": "Mpixel-iterations/s", "cores": threads,
            "kind": "port",
            "sample": f"{iters} CSV iterations of the same {n}x{n}x{C} image after 1 warm-up iteration "
                      f"({dt:.1f} s); OpenMP teams as the reference: 3 threads in curvature(), "
                      f"{threads} for the delta map, c1/c2 sweeps serial", **extra}


if __name__ == "__main__":
    main()
