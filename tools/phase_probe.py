"""Diagnostic: per-iteration time of the default step as the level set evolves from the checkerboard."""
import sys
sys.path.insert(0, '.')
from chan_vese_amd import capi, synth
n = 4096
ctx = capi.Context(n, n, 1, capi.make_params(tol=0.0))
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
ctx.set_image([synth.disk(n)]); ctx.set_levelset(capi.checkerboard_host(n, n))
done = 0
for batch in [16] * 8 + [64] * 4 + [128] * 2:
    ctx.enqueue_steps(batch); ctx.sync()
    done += batch
    ms = ctx.last_run_ms()
    u = ctx.get_levelset()
    import numpy as np
    a_ = np.abs(u)
    rows_near = (a_.reshape(n, -1, 128).min(axis=2) < 64).mean()   # share of 128-column row segments holding a near-field pixel
    print("iterations %4d..%4d: %.2f us/iter   |u|<64: %.5f of pixels, %.4f of 128-px row segments; |u| median %.0f max %.0f" % (done - batch, done, ms * 1e3 / batch, (a_ < 64).mean(), rows_near, np.median(a_), a_.max()))
