"""Diagnostic: is the 63 / 67 us mode of the default step a property of the process or of the allocation?"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = 4096
img = synth.disk(n); u0 = capi.checkerboard_host(n, n)
keep = []
for trial in range(6):
    ctx = capi.Context(n, n, 1, capi.make_params(tol=0.0))
    ctx.set_image([img]); ctx.set_levelset(u0)
    ctx.enqueue_steps(40); ctx.sync()
    res = []
    for rep in range(3):
        ctx.enqueue_steps(300); ctx.sync()
        res.append(ctx.last_run_ms() * 1e3 / 300)
    print("context %d: %s us/iter" % (trial, ["%.2f" % r for r in res]))
    if trial % 2 == 0: keep.append(ctx)      # keep some alive so that later contexts get other addresses
    else: ctx.close()
