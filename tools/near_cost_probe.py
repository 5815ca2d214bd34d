"""What the pixels near the contour cost: HIP-event time per iteration with the disk's contour (after 400 iterations from the
checkerboard) against a level set that is far from zero everywhere but along the perimeter of one 64 x 64 square.
usage: near_cost_probe.py  [N=4096 STEPS=112 REPS=3 RESIDENT=0]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "4096")); steps = int(os.environ.get("STEPS", "112")); reps = int(os.environ.get("REPS", "3"))
ctx = capi.Context(n, n, 1, capi.make_params(tol=0.0))
ctx.set_option("resident", int(os.environ.get("RESIDENT", "0")))
ctx.set_image([synth.disk(n)])
def timed():
    ctx.warm(steps); ctx.enqueue_steps(steps); ctx.sync()
    return ctx.last_run_ms() * 1e3 / steps
sq = np.full((n, n), -50.0); sq[n // 2 - 32:n // 2 + 32, n // 2 - 32:n // 2 + 32] = 50.0
half = np.full((n, n), -50.0); half[:, : n // 2] = 50.0
rows = np.full((n, n), -50.0); rows[: n // 2, :] = 50.0
for name, u0, pre in (("disk contour (400 iterations from the checkerboard)", capi.checkerboard_host(n, n), 400), ("one 64x64 square", sq, 0),
                      ("one vertical line", half, 0), ("one horizontal line", rows, 0), ("checkerboard itself (every pixel near)", capi.checkerboard_host(n, n), 0)):
    t = []
    for r in range(reps):
        ctx.set_levelset(u0)
        if pre: ctx.enqueue_steps(pre); ctx.sync()
        t.append(timed())
    print("%-55s %s  median %.2f us" % (name, " ".join("%.2f" % v for v in t), np.median(t)))
ctx.close()
