"""Diagnostic: Perona-Malik step time by image size; options as key=value arguments (e.g. pm_kernel=3 pm_strip_rows=48)."""
import sys; sys.path.insert(0, '.')
from chan_vese_amd import capi, synth
opts = [kv.split("=") for kv in sys.argv[1:]]
for n in (512, 1024, 2048, 4096):
    img = [synth.disk(n, 200, 50, noise=32, seed=1)]
    with capi.Context(n, n, 1) as ctx:
        for k, v in opts: ctx.set_option(k, int(v))
        ctx.set_image(img); ctx.perona_malik(30, 0.25, 25); ctx.set_image(img)
        ctx.perona_malik(30, 0.25, 50)
        ms = ctx.last_pm_ms()
        print("PM %d^2 %s: %.2f us/step  %.2f TB/s (16 B/px)  frac %.3f" % (n, " ".join(sys.argv[1:]), ms * 1e3 / 200, 16.0 * n * n * 200 / ms / 1e9, 16.0 * n * n * 200 / ms / 1e9 / 8))
