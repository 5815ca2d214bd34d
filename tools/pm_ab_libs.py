"""A/B of BUILDS of libchanvese_hip.so on the Perona-Malik phase inside ONE process: each library gets its own context on the same image, the
contexts take turns; prints HIP-event us per time step per library and round.
usage: pm_ab_libs.py path/to/libA.so path/to/libB.so ...   [N=2048 REPS=4 STEPS=400 OPTS=pm_kernel=4,...]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "2048")); reps = int(os.environ.get("REPS", "4")); steps = int(os.environ.get("STEPS", "400"))
img = synth.disk(n, 200, 50, noise=40, seed=1)
ctxs = []
for path in sys.argv[1:]:
    capi._lib = None; capi.LIB_PATH = os.path.abspath(path)      # bind a fresh handle of this build
    capi.lib()
    ctx = capi.Context(n, n, 1)
    for kv in os.environ.get("OPTS", "").split(","):
        if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.set_image([img]); ctx.perona_malik(30.0, 0.25, 25.0)
    ctxs.append((path, ctx))
res = np.zeros((len(ctxs), reps))
ref = None
for r in range(reps):
    for i, (path, ctx) in enumerate(ctxs):
        ctx.set_image([img]); ctx.perona_malik(30.0, 0.25, 0.25 * steps)
        res[i, r] = ctx.last_pm_ms() * 1e3 / steps
        if r == 0:
            out = ctx.get_image()[0]
            if ref is None: ref = out
            elif not np.array_equal(out, ref): print("!! %s: result differs from the first library's" % path)
for i, (path, ctx) in enumerate(ctxs):
    print("%-70s %s  median %.2f us/step  %s" % (path[-70:], " ".join("%.2f" % v for v in res[i]), np.median(res[i]), ctx.launch_info(1)["kernel"]))
