"""Persistent step kernel vs one launch per iteration (both kernel 3): same level set, trace and stop iteration; then us/iteration.
usage: persist_probe.py [check|time] sizes..."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth

def run(h, w, persist, steps, tol, chunks):
    img = synth.disk(max(h, w), 200, 50, noise=16, seed=3)[:h, :w]
    ctx = capi.Context(h, w, 1, capi.make_params(tol=tol))
    ctx.set_option("kernel", 3); ctx.set_option("persist", persist); ctx.set_option("trace", steps + 8)
    ctx.set_image([np.ascontiguousarray(img)]); ctx.init_checkerboard()
    import ctypes as C
    cap, nb = C.c_int(0), C.c_int(0)
    L = capi.lib(); L.cvh_debug_persist_active.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    act = L.cvh_debug_persist_active(ctx._h, C.byref(cap), C.byref(nb))
    if persist and not getattr(run, "said", False): print(f"  persistent path active={act} capacity={cap.value} workgroups={nb.value}", flush=True); run.said = True
    for n in chunks: ctx.enqueue_steps(n)
    done, nrm, stopped = ctx.sync()
    u = ctx.get_levelset(); tr = ctx.get_trace(steps + 8); ms = ctx.last_run_ms()
    ctx.close()
    return u, tr, done, nrm, stopped, ms

mode = sys.argv[1]
for arg in sys.argv[2:]:
    h, w = (int(x) for x in arg.split("x"))
    if mode == "check":
        for tol, chunks in ((0.0, [1]), (0.0, [2]), (0.0, [5, 1, 7]), (0.0, [40]), (2e-2, [40]), (1e-3, [150, 150])):
            steps = sum(chunks)
            a = run(h, w, 0, steps, tol, chunks); b = run(h, w, 1, steps, tol, chunks)
            du = np.max(np.abs(a[0] - b[0])) / max(np.max(np.abs(a[0])), 1e-300)
            n = min(len(a[1]), len(b[1]))
            dt = np.max(np.abs(a[1][:n] - b[1][:n]) / np.maximum(np.abs(a[1][:n]), 1e-300)) if n else 0.0
            print(f"{h}x{w} tol {tol} chunks {chunks}: done {a[2]}/{b[2]} stopped {a[4]}/{b[4]} rows {len(a[1])}/{len(b[1])} max|du|/max|u| {du:.2e} trace rel {dt:.2e} norm {a[3]:.6g}/{b[3]:.6g}", flush=True)
    else:
        for persist in (0, 1, 0, 1):
            r = run(h, w, persist, 400, 0.0, [100, 300])
            print(f"{h}x{w} persist {persist}: {r[5] * 1e3 / 400:.2f} us/iteration (400 iterations incl. first chunk)", flush=True)
