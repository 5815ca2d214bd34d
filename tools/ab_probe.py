"""A/B inside ONE process and ONE context (same buffers, same clocks): cycles through option settings several times,
112 iterations each (7 graphs), and prints the HIP-event time per iteration of every (setting, repetition).
usage: ab_probe.py "wave_cskew=0" "wave_cskew=100" "wave_cskew=130,chain=0" ...   [N=4096 REPS=3 STEPS=112]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
C_ = int(os.environ.get("C", "1"))
n = int(os.environ.get("N", "4096")); reps = int(os.environ.get("REPS", "3")); steps = int(os.environ.get("STEPS", "112"))
settings = [dict((k, int(v)) for k, v in (kv.split("=") for kv in arg.split(","))) for arg in sys.argv[1:]]
H_ = int(os.environ.get("H", n)); W_ = int(os.environ.get("W", n))
ctx = capi.Context(H_, W_, C_, capi.make_params(tol=0.0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1]) if C_ == 3 else capi.make_params(tol=0.0))
ctx.set_image([synth.disk(n, 180, 40, h=H_, w=W_), synth.disk(n, 200, 60, h=H_, w=W_), synth.disk(n, 60, 200, h=H_, w=W_)] if C_ == 3 else [synth.disk(n, h=H_, w=W_)]); ctx.set_levelset(capi.checkerboard_host(H_, W_))
ctx.enqueue_steps(400); ctx.sync()          # clocks up, far field everywhere
keys = sorted({k for s in settings for k in s})
res = np.zeros((len(settings), reps))
for r in range(reps):
    for i, s in enumerate(settings):
        for k in keys: ctx.set_option(k, s.get(k, {"chain": 1, "wave_cls": 1, "far_terms": 5, "wave_prio": 1, "wave_sync": -1, "kernel": -1, "wave_occupancy": 5, "wave_cskew": 500, "wave_early": 1, "wave_pol": -1, "lut": 1, "strip_rows": 0, "resident": 0, "near_switch": 1, "wave_xcd": 1, "state": 64, "co_resident": 1, "res_prio": 1, "res_go_share": 5}.get(k, 0)))
        ctx.warm(steps); ctx.enqueue_steps(16); ctx.sync()
        ctx.warm(steps); ctx.enqueue_steps(steps); ctx.sync()
        res[i, r] = ctx.last_run_ms() * 1e3 / steps
for i, s in enumerate(settings):
    print("%-40s  %s   median %.2f us  (%dx%d: %.4f ns/px)" % (",".join("%s=%d" % kv for kv in s.items()), " ".join("%.2f" % v for v in res[i]), np.median(res[i]), H_, W_, np.median(res[i]) * 1e3 / (H_ * W_)))
ctx.close()
