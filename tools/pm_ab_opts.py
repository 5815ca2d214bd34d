"""A/B of OPTION SETS on the Perona-Malik phase inside ONE process: each setting gets its own context on the same image, the contexts take
turns; prints HIP-event us per time step per setting and round, and says whether the planes are the same bytes.
usage: pm_ab_opts.py "pm_kernel=4" "pm_kernel=4,res_prio=0" ...   [N=2048 H= W= REPS=4 STEPS=400 MATH=2]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "2048")); reps = int(os.environ.get("REPS", "4")); steps = int(os.environ.get("STEPS", "400"))
H_ = int(os.environ.get("H", n)); W_ = int(os.environ.get("W", n))
img = synth.disk(n, 200, 50, noise=40, seed=1, h=H_, w=W_)
ctxs = []
for arg in sys.argv[1:]:
    ctx = capi.Context(H_, W_, 1)
    ctx.set_option("math_mode", int(os.environ.get("MATH", "2")))
    for kv in arg.split(","):
        if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.set_image([img]); ctx.perona_malik(30.0, 0.25, 25.0)
    ctxs.append((arg, ctx))
res = np.zeros((len(ctxs), reps))
ref = None
for r in range(reps):
    for i, (arg, ctx) in enumerate(ctxs):
        ctx.set_image([img]); ctx.perona_malik(30.0, 0.25, 0.25 * steps)
        res[i, r] = ctx.last_pm_ms() * 1e3 / steps
        if r == 0:
            out = ctx.get_image()[0]
            if ref is None: ref = out
            elif not np.array_equal(out, ref): print("!! %s: result differs from the first setting's (%d bytes)" % (arg, int((out != ref).sum())))
for i, (arg, ctx) in enumerate(ctxs):
    print("%-44s %s  median %.2f us/step  %s grid %s" % (arg, " ".join("%.2f" % v for v in res[i]), np.median(res[i]), ctx.launch_info(1)["kernel"], ctx.launch_info(1)["grid"]))
