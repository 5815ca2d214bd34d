"""Per-iteration time of the first iterations from the checkerboard: level set from the host function vs cvh_init_checkerboard."""
import sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = 4096
img = synth.disk(n)
for mode in ("host", "device", "host", "device"):
    ctx = capi.Context(n, n, 1, capi.make_params(tol=0.0))
    ctx.set_image([img])
    if mode == "host": ctx.set_levelset(capi.checkerboard_host(n, n))
    else: ctx.init_checkerboard()
    u_start = ctx.get_levelset()
    t = []
    for it in range(30):
        ctx.enqueue_steps(1); ctx.sync(); t.append(ctx.last_run_ms() * 1e3)
    u_end = ctx.get_levelset()
    print(mode, "sum|u0|", np.abs(u_start).sum(), "checksum end", float(np.abs(u_end).sum()), " ".join("%.0f" % x for x in t), flush=True)
    ctx.close()
