"""Is the FIRST launch of an instantiated step graph slower than later ones?  (20-step runs contain exactly one graph launch.)"""
import sys, time
sys.path.insert(0, '.')
from chan_vese_amd import capi, synth
n = 4096
img = synth.disk(n)
scratch = capi.Context(n, n, 1, capi.make_params(tol=0.0)); scratch.set_image([img]); scratch.init_checkerboard()
for rep in range(6):
    ctx = capi.Context(n, n, 1, capi.make_params(tol=0.0))
    ctx.set_image([img]); ctx.init_checkerboard()
    ctx.enqueue_steps(5); ctx.sync()
    ctx.warm(16)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15: scratch.enqueue_steps(64); scratch.sync()
    out = []
    for k in range(4):
        ctx.warm(16); ctx.enqueue_steps(16); ctx.sync(); out.append(ctx.last_run_ms() * 1e3 / 16)
    print("rep %d: us/iteration of graph launches 1..4: %s" % (rep, " ".join("%.1f" % x for x in out)), flush=True)
    ctx.close()
