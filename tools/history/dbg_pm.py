import sys; sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi
rng = np.random.default_rng(1)
h, w = 40, 150
img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
outs = {}
for pk in (0, 1):
    for steps in (1, 2):
        with capi.Context(h, w, 1) as ctx:
            ctx.set_option("math_mode", 1); ctx.set_option("pm_kernel", pk)
            ctx.set_image([img]); ctx.perona_malik(30, 0.25, 0.25 * steps)
            outs[(pk, steps)] = ctx.get_image()[0].astype(int)
for steps in (1, 2):
    d = outs[(0, steps)] != outs[(1, steps)]
    print("steps", steps, "mismatches", d.sum(), "rows", np.unique(np.nonzero(d)[0])[:20], "cols", np.unique(np.nonzero(d)[1])[:40])
