"""PM and 3-channel timings (diagnostic)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = 2048
img = synth.config_planes("C4", n)
with capi.Context(n, n, 1, capi.make_params(tol=0.0)) as ctx:
    ctx.set_image(img)
    ctx.perona_malik(30, 0.25, 25)      # warm-up 100 steps
    ctx.set_image(img)
    ctx.perona_malik(30, 0.25, 250)
    ms = ctx.last_pm_ms()
    print("PM 2048^2 1000 steps: %.2f ms  %.2f us/step  %.0f Mpx-steps/s  %.2f TB/s (16 B/px)" % (ms, ms, n*n*1000/ms/1e3, 16.0*n*n*1000/ms/1e9))
    ctx.init_checkerboard()
    ctx.run(200)
    ms = ctx.last_run_ms()
    print("CSV 2048^2 200 iters after PM: %.2f ms  %.1f us/iter" % (ms, ms*1e3/200))
n = 4096
pl = synth.config_planes("C3", n)
with capi.Context(n, n, 3, capi.make_params(tol=0.0, lambda1=[1,1,.5], lambda2=[1,.5,1])) as ctx:
    ctx.set_image(pl); ctx.init_checkerboard(); ctx.run(20); ctx.run(100)
    ms = ctx.last_run_ms()
    print("CSV 4096^2 x3ch 100 iters: %.1f us/iter  %.0f Mpx-it/s  %.2f TB/s (19 B/px) frac %.3f" % (ms*1e3/100, n*n*100/ms/1e3, 19.0*n*n*100/ms/1e9, 19.0*n*n*100/ms/1e9/8))
