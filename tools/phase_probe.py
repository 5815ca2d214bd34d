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
    print("iterations %4d..%4d: %.2f us/iter" % (done - batch, done, ctx.last_run_ms() * 1e3 / batch))
