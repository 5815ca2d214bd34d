"""A/B of two (or more) BUILDS of libchanvese_hip.so inside ONE process: each library gets its own context on the same image,
the contexts take turns (STEPS iterations each, REPS rounds); prints HIP-event us per iteration per library and round.
usage: ab_libs.py path/to/libA.so path/to/libB.so ...   [N=4096 REPS=4 STEPS=112 OPTS=key=value,...]"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "4096")); reps = int(os.environ.get("REPS", "4")); steps = int(os.environ.get("STEPS", "112"))
C_ = int(os.environ.get("C", "1"))      # C=3: the 3-channel disks of config C3, per-channel lambdas
imgs = [synth.disk(n, 180, 40), synth.disk(n, 200, 60), synth.disk(n, 60, 200)] if C_ == 3 else [synth.disk(n)]; u0 = None
ctxs = []
for path in sys.argv[1:]:
    capi._lib = None; capi.LIB_PATH = os.path.abspath(path)      # bind a fresh handle of this build
    L = capi.lib()
    if u0 is None: u0 = capi.checkerboard_host(n, n)
    ctx = capi.Context(n, n, C_, capi.make_params(tol=0.0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1]) if C_ == 3 else capi.make_params(tol=0.0))
    for kv in os.environ.get("OPTS", "").split(","):     # e.g. OPTS=kernel=3,wave_cskew=400
        if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.set_image(imgs); ctx.set_levelset(u0); ctx.enqueue_steps(400); ctx.sync()
    ctxs.append((path, ctx))
res = np.zeros((len(ctxs), reps))
for r in range(reps):
    for i, (path, ctx) in enumerate(ctxs):
        ctx.warm(steps); ctx.enqueue_steps(16); ctx.sync()
        ctx.warm(steps); ctx.enqueue_steps(steps); ctx.sync()
        res[i, r] = ctx.last_run_ms() * 1e3 / steps
for i, (path, ctx) in enumerate(ctxs):
    print("%-70s %s  median %.2f us" % (path[-70:], " ".join("%.2f" % v for v in res[i]), np.median(res[i])))
