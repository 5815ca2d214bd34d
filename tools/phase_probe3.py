"""Diagnostic: per-chunk time of the 3-channel step over a long run (is the drift data- or clock-related?)."""
import sys; sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = 4096
planes = synth.config_planes("C3", n)
u0 = capi.checkerboard_host(n, n)
ctx = capi.Context(n, n, 3, capi.make_params(tol=0.0, lambda1=[1, 1, .5], lambda2=[1, .5, 1]))
ctx.set_image(planes)
for rnd in range(2):
    ctx.set_levelset(u0)
    out = []
    for k in range(12):
        ctx.warm(112); ctx.enqueue_steps(112); ctx.sync(); out.append(ctx.last_run_ms() * 1e3 / 112)
    print("round", rnd, "us/iter per 112-iteration chunk from the checkerboard:", " ".join("%.1f" % v for v in out), flush=True)
u = ctx.get_levelset(); print("min|u| %.1f" % np.abs(u).min())
