"""Randomised cross-check of the resident-plane kernels against the per-launch flows (one process, seeded):
  Perona-Malik: identical uint8 planes (both flavours run the same operations in the same order);
  CSV: level set within 1e-9 of the per-launch flow's after the same iterations, same iteration count and stop flag.
usage: fuzz_resident.py [CASES=60 SEED=1 MAXDIM=700]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi
cases = int(os.environ.get("CASES", "60")); seed = int(os.environ.get("SEED", "1")); maxdim = int(os.environ.get("MAXDIM", "700"))
rng = np.random.default_rng(seed)
bad = 0
for n in range(cases):
    h = int(rng.integers(16, maxdim + 1)); w = 2 * int(rng.integers(8, maxdim // 2 + 1))
    if h % 16 == 1: h += 1      # (a last tile row of ONE row does not qualify for the resident Perona-Malik kernel -- api.hip, pm_resident_geometry)
    C = int(rng.choice([1, 3])); math = int(rng.choice([1, 2])); steps = int(rng.integers(1, 31))
    planes = [rng.integers(0, 256, size=(h, w), dtype=np.uint8) for _ in range(C)]
    K = float(rng.choice([5, 10, 30, 1000])); L = float(rng.choice([0.05, 0.1, 0.25]))
    out = {}
    for pk in (4, 3):
        with capi.Context(h, w, C) as ctx:
            ctx.set_option("math_mode", math); ctx.set_option("pm_kernel", pk)
            ctx.set_image(planes); ctx.perona_malik(K, L, L * steps)
            out[pk] = (ctx.get_image(), ctx.launch_info(1)["kernel"], capi.pm_trip_count(L, L * steps))
    same = all(np.array_equal(a, b) for a, b in zip(out[4][0], out[3][0]))
    ok_pm = same and out[4][1].startswith("pm_resident_kernel")
    # CSV (1 channel, FAST): resident vs per-launch
    img = planes[0]; its = int(rng.integers(1, 40)); tol = float(rng.choice([0.0, 0.0, 1e-3]))
    res = {}
    for r in (1, 0):
        with capi.Context(h, w, 1, capi.make_params(tol=tol)) as ctx:
            ctx.set_option("resident", r); ctx.set_image([img]); ctx.init_checkerboard()
            done, nrm = ctx.run(its)[:2]
            res[r] = (ctx.get_levelset(), done, ctx.launch_info()["kernel"])
    scale = max(np.abs(res[0][0]).max(), 1e-300)
    err = np.abs(res[1][0] - res[0][0]).max() / scale
    ok_csv = res[1][1] == res[0][1] and err <= 1e-9 and res[1][2].startswith("csv_resident_kernel")
    if not (ok_pm and ok_csv): bad += 1
    print("%3d  %4dx%-4d C=%d math=%d  PM %2d steps (trips %d) K=%g L=%g %-34s %s | CSV %2d it tol=%g done %d/%d err %.1e %-24s %s" % (
        n, h, w, C, math, steps, out[4][2], K, L, out[4][1], "same" if same else "DIFFERENT", its, tol, res[1][1], res[0][1], err, res[1][2], "ok" if ok_csv else "BAD"), flush=True)
print("cases", cases, "bad", bad)
sys.exit(1 if bad else 0)
