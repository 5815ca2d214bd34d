"""Diagnostic: growth of the GPU-vs-oracle difference over the first iterations at 4096^2 (C2)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
from oracle import cv_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
img = synth.config_planes("C2", n)
u0 = O.checkerboard(n, n)
p = O.make_params(tol=0)
marks = [1, 2, 3, 4, 6, 10, 14, 18]
uc = u0.copy(); ref = {}; trc = []
for t in range(1, marks[-1] + 1):
    c1 = O.region_mean(img[0], uc, 0); c2 = O.region_mean(img[0], uc, 1)
    nrm = O.csv_step(img, uc, p)
    trc.append((c1, c2, nrm if np.isscalar(nrm) else 0.0))
    if t in marks: ref[t] = uc.copy()
with capi.Context(n, n, 1, capi.make_params(tol=0.0)) as ctx:
    ctx.set_option("trace", marks[-1])
    ctx.set_image(img); ctx.set_levelset(u0)
    done = 0
    for t in marks:
        ctx.run(t - done); done = t
        ug = ctx.get_levelset()
        d = np.abs(ug - ref[t])
        print("iter %2d: max|du|/max|u| %.3e  (max|du| %.3e, median|du| %.3e, max|u| %.1f)" % (t, d.max() / np.abs(ref[t]).max(), d.max(), np.median(d), np.abs(ref[t]).max()))
    tr = ctx.get_trace(marks[-1])
for t, (c1, c2, _) in enumerate(trc):
    print("iter %2d: c1 rel %.2e c2 rel %.2e   c1 %.9f c2 %.9f" % (t + 1, abs(tr[t, 0] - c1) / abs(c1), abs(tr[t, 1] - c2) / abs(c2), c1, c2))
