"""The all-near-field regime (the reference README's second example: --dt 0.001 keeps |u| ~ 1 for the whole run): HIP-event time per
iteration with dt = 0.001 against the default dt = 1 in ONE process, per size / channel count / flow, and the share of pixels below the
far-field threshold (32 eps) at the end of each run.
usage: near_regime_probe.py  [SIZES=4096,2048 C=1 STEPS=112 REPS=3 RESIDENT=-1]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
C_ = int(os.environ.get("C", "1")); steps = int(os.environ.get("STEPS", "112")); reps = int(os.environ.get("REPS", "3"))
for n in [int(s) for s in os.environ.get("SIZES", "4096,2048").split(",")]:
    planes = synth.config_planes("C3", n) if C_ == 3 else [synth.disk(n)]
    for label, dt, pre in (("dt=1 (far field after 16 iterations)", 1.0, 100), ("dt=0.001 (every pixel near)", 0.001, 100)):
        pk = dict(tol=0.0, dt=dt)
        if C_ == 3: pk.update(lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1])
        with capi.Context(n, n, C_, capi.make_params(**pk)) as ctx:
            ctx.set_option("resident", int(os.environ.get("RESIDENT", "-1")))
            ctx.set_image(planes); ctx.init_checkerboard()
            ctx.enqueue_steps(pre); ctx.sync()
            t = []
            for r in range(reps):
                ctx.warm(steps); ctx.enqueue_steps(steps); ctx.sync()
                t.append(ctx.last_run_ms() * 1e3 / steps)
            u = ctx.get_levelset(); m = ctx.get_mask().astype(bool); d = planes[0] > 100
            iou = max((m & d).sum() / max((m | d).sum(), 1), (m & ~d).sum() / max((m | ~d).sum(), 1))
            print("%5d^2 x%d %-40s %s  median %.2f us   kernel %s   |u|<32: %.4f of the pixels, max|u| %.2f, mask IoU vs disk %.5f" % (
                n, C_, label, " ".join("%.2f" % v for v in t), np.median(t), ctx.launch_info()["kernel"], (np.abs(u) < 32).mean(), np.abs(u).max(), iou), flush=True)
