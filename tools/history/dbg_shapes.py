"""Diagnostic: realistic non-square shapes, default settings vs the 1-pixel kernel (GPU vs GPU, 5 iterations) + time per iteration."""
import sys; sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
for (h, w) in [(1080, 1920), (2160, 3840), (1000, 1504), (3000, 4000), (4320, 7680), (720, 1280)]:
    img = synth.disk(min(h, w), 200, 50, noise=12, seed=7, h=h, w=w)
    u0 = capi.checkerboard_host(h, w)
    res = {}
    for name, opts in (("auto", {}), ("k2", {"kernel": 2})):
        with capi.Context(h, w, 1, capi.make_params(tol=0)) as ctx:
            for k, v in opts.items(): ctx.set_option(k, v)
            ctx.set_image([img]); ctx.set_levelset(u0); ctx.run(5); res[name] = ctx.get_levelset()
            ctx.enqueue_steps(100); ctx.sync(); ctx.warm(96); ctx.enqueue_steps(96); ctx.sync()
            res[name + "_us"] = ctx.last_run_ms() * 1e3 / 96
    d = np.abs(res["auto"] - res["k2"]).max() / np.abs(res["k2"]).max()
    print("%dx%d: auto %.1f us (%.3f of roofline), k2 %.1f us; rel diff %.2e" % (h, w, res["auto_us"], 17.0 * h * w / res["auto_us"] / 8e6, res["k2_us"], d), flush=True)
