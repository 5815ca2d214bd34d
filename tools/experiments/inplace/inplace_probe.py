"""Single-buffer flavour of the 2-pixel kernel vs the ping-pong pair: same level set, trace and stop iteration; then us/iteration.
usage: inplace_probe.py check|time HxW ..."""
import sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth

def run(h, w, inplace, steps, tol, chunks, opts=()):
    img = synth.disk(max(h, w), 200, 50, noise=16, seed=3)[:h, :w]
    ctx = capi.Context(h, w, 1, capi.make_params(tol=tol))
    ctx.set_option("kernel", 3); ctx.set_option("inplace", inplace); ctx.set_option("trace", steps + 8)
    for k, v in opts: ctx.set_option(k, v)
    ctx.set_image([np.ascontiguousarray(img)]); ctx.init_checkerboard()
    for n in chunks: ctx.enqueue_steps(n)
    done, nrm, stopped = ctx.sync()
    u = ctx.get_levelset(); tr = ctx.get_trace(steps + 8); ms = ctx.last_run_ms()
    m = ctx.get_mask()
    ctx.close()
    return u, tr, done, nrm, stopped, ms, m

mode = sys.argv[1]
for arg in sys.argv[2:]:
    h, w = (int(x) for x in arg.split("x"))
    if mode == "check":
        for tol, chunks in ((0.0, [1]), (0.0, [2]), (0.0, [5, 1, 7]), (0.0, [40]), (2e-2, [40]), (1e-3, [150, 150]), (0.0, [16, 33])):
            steps = sum(chunks)
            a = run(h, w, 0, steps, tol, chunks); b = run(h, w, 1, steps, tol, chunks)
            du = np.max(np.abs(a[0] - b[0])) / max(np.max(np.abs(a[0])), 1e-300)
            n = min(len(a[1]), len(b[1]))
            dt = np.max(np.abs(a[1][:n] - b[1][:n]) / np.maximum(np.abs(a[1][:n]), 1e-300)) if n else 0.0
            if du > 0:
                bad = np.argwhere(a[0] != b[0])
                print(f"   {len(bad)} pixels differ; rows {bad[:, 0].min()}..{bad[:, 0].max()} cols {bad[:, 1].min()}..{bad[:, 1].max()}; first {bad[:6].tolist()}; distinct cols {np.unique(bad[:, 1])[:12].tolist()} distinct rows {np.unique(bad[:, 0])[:12].tolist()}")
            print(f"{h}x{w} tol {tol} chunks {chunks}: done {a[2]}/{b[2]} stopped {a[4]}/{b[4]} rows {len(a[1])}/{len(b[1])} max|du|/max|u| {du:.2e} trace rel {dt:.2e} norm {a[3]:.6g}/{b[3]:.6g} mask equal {np.array_equal(a[6], b[6])}", flush=True)
    else:
        for inplace in (0, 1, 0, 1):
            r = run(h, w, inplace, 416, 0.0, [112, 304])
            print(f"{h}x{w} inplace {inplace}: {r[5] * 1e3 / 416:.2f} us/iteration", flush=True)
