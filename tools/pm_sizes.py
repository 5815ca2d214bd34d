import sys; sys.path.insert(0, '.')
from chan_vese_amd import capi, synth
for n in (512, 1024, 2048, 4096):
    img = [synth.disk(n, 200, 50, noise=32, seed=1)]
    with capi.Context(n, n, 1) as ctx:
        ctx.set_image(img); ctx.perona_malik(30, 0.25, 25); ctx.set_image(img)
        ctx.perona_malik(30, 0.25, 50)
        ms = ctx.last_pm_ms()
        print("PM %d^2: %.2f us/step  %.2f TB/s (16 B/px)" % (n, ms * 1e3 / 200, 16.0 * n * n * 200 / ms / 1e9))
