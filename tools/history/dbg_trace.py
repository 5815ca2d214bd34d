import sys; sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
from oracle import cv_oracle as O
h, w = 150, 528
img = synth.disk(150, 190, 60, noise=12, seed=9, h=h, w=w)
u0 = O.checkerboard(h, w)
u_c, _, nrm_c, tr_c = O.csv_run([img], u0, O.make_params(tol=0), 6)
for math in (1, 2):
    for rep in range(3):
        with capi.Context(h, w, 1, capi.make_params(tol=0)) as ctx:
            ctx.set_option("math_mode", math); ctx.set_option("kernel", 2); ctx.set_option("trace", 6)
            ctx.set_image([img]); ctx.set_levelset(u0)
            done, nrm = ctx.run(6)
            tr = ctx.get_trace(6); u = ctx.get_levelset()
        print("math", math, "rep", rep, "relerr u", np.abs(u-u_c).max()/np.abs(u_c).max(), "trace ok", np.allclose(tr, tr_c, rtol=1e-9))
        if not np.allclose(tr, tr_c, rtol=1e-9): print(tr, "\n", tr_c)
