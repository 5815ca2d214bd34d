#!/usr/bin/env python3
"""Generates tests/golden/*.npz — inputs and expected outputs of the hot path on small cases.

PARITY UNPINNED: the reference (ktht/chan_vese) ships no golden vectors and cannot be built
here (OpenCV 2.4.8 / Boost 1.59 absent), so these vectors come from THIS repository's CPU
oracle (oracle/cv_oracle.c, a reading of /root/reference/src/main.cpp), generated in the
build container with gcc 11.4 / glibc 2.35.  They pin the oracle against regressions and give
the GPU tests fixed data; they are not outputs of the reference binary.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chan_vese_amd import synth  # noqa: E402
from oracle import cv_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SNAP = (1, 2, 3, 10)


def csv_case(name, planes, u0, **pk):
    p = O.make_params(**pk)
    rec = {"planes": np.stack(planes), "u0": u0,
           "params": np.array([pk.get("mu", .5), pk.get("nu", 0.), pk.get("dt", 1.), pk.get("eps", 1.), pk.get("tol", 1e-3)]),
           "lambda1": np.array(list(pk.get("lambda1", [1, 1, 1])), dtype=np.float64),
           "lambda2": np.array(list(pk.get("lambda2", [1, 1, 1])), dtype=np.float64)}
    for s in SNAP:
        u, done, nrm, tr = O.csv_run(planes, u0, p, s)
        rec[f"u_{s}"] = u
        rec[f"trace_{s}"] = tr
    rec["stop_cond"] = np.array([O.stop_condition(planes, pk.get("tol", 1e-3))])
    rec["mask_10"] = O.mask(rec["u_10"])
    rec["contour_10"] = O.video_contour(rec["u_10"])   # the video frame's contour map (src/VideoWriterManager.cpp:60-74)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)


def main():
    rng = np.random.default_rng(20261004)
    csv_case("csv_32x48_defaults", [rng.integers(0, 256, (32, 48), dtype=np.uint8)], O.checkerboard(32, 48), tol=0.0)
    csv_case("csv_37x53_odd_params", [synth.disk(40, 180, 70, noise=20, seed=5, h=37, w=53)], O.checkerboard(37, 53),
             mu=0.2, nu=0.05, dt=0.1, eps=0.5, tol=0.0)
    csv_case("csv_48x40_3ch", [synth.disk(48, 180, 40, h=48, w=40), synth.disk(48, 200, 60, h=48, w=40),
                               synth.disk(48, 60, 200, h=48, w=40)], O.checkerboard(48, 40),
             tol=0.0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1])
    csv_case("csv_24x64_rect_init", [synth.disk(24, 210, 30, h=24, w=64)], O.levelset_rect(24, 64, 10, 4, 30, 12), tol=0.0)
    # Perona-Malik: uint8 in -> uint8 out (+ final double state)
    img = synth.disk(48, 200, 50, noise=32, seed=1, h=40, w=56)
    out, st = O.perona_malik([img], 30, 0.25, 5, want_state=True)
    np.savez_compressed(os.path.join(OUT, "pm_40x56_K30_L025_T5.npz"), img=img, out=out[0], state=st[0],
                        klt=np.array([30, 0.25, 5]), trips=np.array([O.pm_trip_count(0.25, 5)]))
    # known answers: checkerboard census and the 512^2 trajectory
    u, done, last, tr = O.csv_run([synth.disk(512)], O.checkerboard(512, 512), O.make_params(tol=0.0), 100)
    np.savez_compressed(os.path.join(OUT, "traj_512_disk.npz"), trace=tr, umax=np.array([np.abs(u).max()]),
                        mask_sum=np.array([int(O.mask(u).sum())]))
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
