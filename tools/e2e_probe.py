"""Host-visible time of one image through the boundary: set_image, init_checkerboard, run, get_mask (ms each).
Usage: python tools/e2e_probe.py [H] [C] [STEPS]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from chan_vese_amd import capi, synth

h = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 500
planes = [synth.disk(h, 200, 50, noise=16, seed=1000 + k) for k in range(C)]
with capi.Context(h, h, C, capi.make_params(tol=0)) as ctx:
    for rep in range(3):
        t = [time.perf_counter()]
        ctx.set_image(planes); t.append(time.perf_counter())
        ctx.init_checkerboard(); t.append(time.perf_counter())
        done, _ = ctx.run(steps); t.append(time.perf_counter())
        m = ctx.get_mask(); t.append(time.perf_counter())
        d = [1e3 * (b - a) for a, b in zip(t, t[1:])]
        print(f"{h}x{h}x{C} rep {rep}: set_image {d[0]:.1f}  init_checkerboard {d[1]:.1f}  run({done}) {d[2]:.1f} "
              f"(device {ctx.last_run_ms():.1f})  get_mask {d[3]:.1f}  total {sum(d):.1f} ms", flush=True)
