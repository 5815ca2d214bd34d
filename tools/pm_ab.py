"""PM timing A/B in one process: us per step of 1000 steps at N^2 for several (pm_kernel, pm_strip_rows)."""
import os, sys
sys.path.insert(0, '.')
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "2048"))
img = synth.config_planes("C4", n) if n == 2048 else [synth.disk(n, 200, 50, noise=32, seed=1)]
ctx = capi.Context(n, n, 1, capi.make_params(tol=0.0))
ctx.set_image(img); ctx.perona_malik(30, 0.25, 50)
for rep in range(2):
    for arg in sys.argv[1:]:
        opts = dict((k, int(v)) for k, v in (kv.split("=") for kv in arg.split(",")))
        ctx.set_option("pm_kernel", opts.get("pm_kernel", -1)); ctx.set_option("pm_strip_rows", opts.get("pm_strip_rows", 0))
        ctx.set_option("graph", opts.get("graph", 1)); ctx.set_option("wave_pol", opts.get("wave_pol", -1))
        ctx.set_image(img); ctx.perona_malik(30, 0.25, 250)
        ms = ctx.last_pm_ms()
        print("%-40s %.2f us/step  frac %.3f" % (arg, ms, 16.0 * n * n / (ms * 1e-6) / 8e12), flush=True)
ctx.close()
