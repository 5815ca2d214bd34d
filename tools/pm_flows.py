"""Perona-Malik data flows against each other, one context per size, the flows alternated: HIP-event us per time step (load / store of
the uint8 plane included, as bench.py's C4 phase counts them).
usage: pm_flows.py [SIZES=128,256,512,1024,1536,2048 STEPS=400 REPS=3 FLOWS=3,4 OPTS=key=value,...]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
sizes = [int(x) for x in os.environ.get("SIZES", "128,256,512,1024,1536,2048").split(",")]
steps = int(os.environ.get("STEPS", "400")); reps = int(os.environ.get("REPS", "3"))
flows = [int(x) for x in os.environ.get("FLOWS", "3,4").split(",")]
math = int(os.environ.get("MATH", "2"))
for n in sizes:
    img = synth.disk(n, 200, 50, noise=40, seed=1)
    with capi.Context(n, n, 1) as ctx:
        ctx.set_option("math_mode", math)
        for kv in os.environ.get("OPTS", "").split(","):      # e.g. OPTS=wave_pol=0
            if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        res = {f: [] for f in flows}
        for r in range(reps + 1):
            for f in flows:
                ctx.set_option("pm_kernel", f)
                ctx.set_image([img])
                ctx.perona_malik(30.0, 0.25, 0.25 * steps)
                if r: res[f].append(ctx.last_pm_ms() * 1e3 / steps)
        print("%5d^2  " % n + "   ".join("pm_kernel=%d: %s median %.2f us/step (%.3f of 16 B/px at 8 TB/s)" % (
            f, " ".join("%.2f" % v for v in res[f]), np.median(res[f]), 16.0 * n * n / (np.median(res[f]) * 1e-6) / 8e12) for f in flows), flush=True)
