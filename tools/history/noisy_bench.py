import sys; sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = 2048
u0 = capi.checkerboard_host(n, n)
img = synth.disk(n, 200, 50, noise=32, seed=1)
with capi.Context(n, n, 1, capi.make_params(tol=0.0)) as ctx:
    ctx.set_image([img]); ctx.perona_malik(30, 0.25, 250); ctx.set_levelset(u0)
    ctx.enqueue_steps(30); ctx.sync()
    ctx.enqueue_steps(200); ctx.sync()
    u = ctx.get_levelset()
    print("after PM, enqueue: %.1f us/iter" % (ctx.last_run_ms() * 1e3 / 200), "min|u| %.2f frac|u|<64 %.4f" % (np.abs(u).min(), (np.abs(u) < 64).mean()))
    ctx.set_levelset(u0)
    done, nrm = ctx.run(230)
    print("after PM, run(230): %.1f us/iter" % (ctx.last_run_ms() * 1e3 / 230), done)
    ctx.set_levelset(u0); ctx.set_option("sync_every", 1000)
    done, nrm = ctx.run(230)
    print("after PM, run(230) sync_every=1000: %.1f us/iter" % (ctx.last_run_ms() * 1e3 / 230), done)
