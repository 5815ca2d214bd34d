#!/usr/bin/env python3
"""Generates the full-size fixtures tests/golden/c2_4096_500.npz and c3_4096_300.npz: BASELINE configs[1] and [2]
(C2 = 4096² disk 200/50, checkerboard, 500 iterations, tol 0; C3 = 4096² × 3 channels, per-channel
lambdas, 300 iterations) run AT THEIR CONFIGURED LENGTH on the CPU oracle, in two variants:

  ref    the restatement as the reference sums (sequential double sums, /root/reference/src/main.cpp:272-280) -- cvo_csv_run
  exact  the same per-pixel terms added without accumulation error (cvo_csv_step_exact, the adjudicator of DESIGN.md §2)

PARITY UNPINNED: the reference holds no fixtures and cannot be built here (OpenCV 2.4.8 / Boost 1.59 absent); these are
outputs of THIS repository's oracle (oracle/cv_oracle.c), generated in the build container (gcc 11.4, glibc 2.35, x86-64,
-ffp-contract=off).  They let the GPU box compare the configured length without spending 15-20 minutes of CPU per variant.

What is kept of a 128 MiB level set (compact on purpose): the c1 / c2 / norm trace of EVERY iteration, the level set on a
stride-16 grid (offset 5, 3: not aligned with lanes, strips or tiles), eight full rows (four through the disk's centre, four
where the contour is tangent to a row), max|u|, the packed mask, the SHA-256 of the full array.

Generation time in this container (8 cores, four variants side by side, 2 OpenMP threads each): see `seconds` in each file
(about 20-35 minutes per variant).  Run from the repo root:

    python tests/golden/make_golden_4096.py            # all four variants in parallel, then merge
    python tests/golden/make_golden_4096.py C2 ref     # one variant -> tests/golden/_part_C2_ref.npz
    python tests/golden/make_golden_4096.py merge
"""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

N = 4096
GRID = (slice(5, None, 16), slice(3, None, 16))
ROWS = [2046, 2047, 2048, 2049, 1022, 1023, 1024, 1025]
CONFIGS = {
    # name: (file stem, iterations)
    "C2": ("c2_4096_500", 500),
    "C3": ("c3_4096_300", 300),
}


def workload(cfg):
    from chan_vese_amd import synth
    from oracle import cv_oracle as O
    if cfg == "C2":
        return [synth.disk(N, 200, 50)], O.make_params(tol=0.0)
    planes = [synth.disk(N, 180, 40), synth.disk(N, 200, 60), synth.disk(N, 60, 200)]
    return planes, O.make_params(tol=0.0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1])


def run_variant(cfg, variant):
    from oracle import cv_oracle as O
    stem, iters = CONFIGS[cfg]
    planes, p = workload(cfg)
    nc = len(planes)
    u = O.checkerboard(N, N)
    t0 = time.time()
    if variant == "ref":
        u, done, last, tr = O.csv_run(planes, u, p, iters)
        assert done == iters
    else:
        tr = np.zeros((iters, 2 * nc + 1))
        for t in range(iters):
            nrm, c1, c2 = O.csv_step_exact(planes, u, p)
            tr[t, :nc], tr[t, nc:2 * nc], tr[t, 2 * nc] = c1, c2, nrm
            if t % 25 == 0:
                print(cfg, variant, t, f"{time.time() - t0:.0f}s", flush=True)
    secs = time.time() - t0
    mask = O.mask(u)
    np.savez_compressed(os.path.join(OUT, f"_part_{cfg}_{variant}.npz"),
                        trace=tr, grid=u[GRID].copy(), rows=u[ROWS].copy(), umax=np.array([np.abs(u).max()]),
                        mask_bits=np.packbits(mask), mask_sum=np.array([int(mask.sum())]),
                        sha256=np.frombuffer(hashlib.sha256(u.tobytes()).digest(), dtype=np.uint8),
                        seconds=np.array([secs]))
    print(cfg, variant, "done", f"{secs:.0f}s", flush=True)


def merge():
    for cfg, (stem, iters) in CONFIGS.items():
        rec = {"iterations": np.array([iters]), "n": np.array([N]), "grid_offset_stride": np.array([5, 3, 16]),
               "rows_index": np.array(ROWS)}
        parts = {v: np.load(os.path.join(OUT, f"_part_{cfg}_{v}.npz")) for v in ("ref", "exact")}
        for v, d in parts.items():
            for k in ("trace", "grid", "rows", "umax", "mask_sum", "sha256", "seconds"):
                rec[f"{k}_{v}"] = d[k]
        # one packed mask: the reference-order one; the exact-sum run's mask is recorded as the pixels that differ
        rec["mask_bits_ref"] = parts["ref"]["mask_bits"]
        diff = np.flatnonzero(np.unpackbits(parts["ref"]["mask_bits"]) != np.unpackbits(parts["exact"]["mask_bits"]))
        rec["mask_exact_differs_at"] = diff.astype(np.int64)
        np.savez_compressed(os.path.join(OUT, stem + ".npz"), **rec)
        print(stem, {k: v.shape for k, v in rec.items()}, "mask differs at", diff.size, "pixels;",
              os.path.getsize(os.path.join(OUT, stem + ".npz")), "bytes")


def main():
    if len(sys.argv) == 3:
        return run_variant(sys.argv[1], sys.argv[2])
    if len(sys.argv) == 2 and sys.argv[1] == "merge":
        return merge()
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), c, v], env=env)
             for c in CONFIGS for v in ("ref", "exact")]
    rc = [p.wait() for p in procs]
    assert not any(rc), rc
    merge()


if __name__ == "__main__":
    main()
