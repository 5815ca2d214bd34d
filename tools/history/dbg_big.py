"""Diagnostic: 8192^2 (beyond the Infinity Cache by far): the wave kernels agree with each other and how fast they are."""
import sys; sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
planes = [synth.disk(n)]
u0 = capi.checkerboard_host(n, n)
res = {}
for name, opts in (("auto", {}), ("k2", {"kernel": 2}), ("k3", {"kernel": 3})):
    with capi.Context(n, n, 1, capi.make_params(tol=0)) as ctx:
        for k, v in opts.items(): ctx.set_option(k, v)
        ctx.set_image(planes); ctx.set_levelset(u0); ctx.run(3); res[name] = ctx.get_levelset()
        ctx.enqueue_steps(64); ctx.sync(); ctx.warm(64); ctx.enqueue_steps(64); ctx.sync()
        us = ctx.last_run_ms() * 1e3 / 64
        print("%-5s %.1f us/iter  frac %.3f" % (name, us, 17.0 * n * n / us / 1e6 / 8e6), flush=True)
for name in ("auto", "k3"):
    print(name, "vs k2 max diff %.3e (max|u| %.1f)" % (np.abs(res[name] - res["k2"]).max(), np.abs(res["k2"]).max()))
