"""A batch of IMAGES independent contexts on one GPU, their launches interleaved in chunks of CHUNK iterations: wall time per image-iteration
with the automatic flow (planes that fit the chip run the resident kernel: cooperative launches of different contexts serialise) against
the per-launch flow ("resident" = 0).   usage: batch_probe.py  [N=2048 IMAGES=8 STEPS=400 CHUNK=8 REPS=3]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "2048")); images = int(os.environ.get("IMAGES", "8")); steps = int(os.environ.get("STEPS", "400"))
chunk = int(os.environ.get("CHUNK", "8")); reps = int(os.environ.get("REPS", "3"))
for label, opts in (("auto", {}), ("resident=0", {"resident": 0}), ("resident=1, chunks of 50", {"resident": 1, "_chunk": 50}), ("auto, chunks of 50", {"_chunk": 50}),
                    ("resident=1, chunks of 100", {"resident": 1, "_chunk": 100}), ("resident=1, chunks of 400", {"resident": 1, "_chunk": 400}), ("resident=0, chunks of 400", {"resident": 0, "_chunk": 400})):
    ck = opts.pop("_chunk", chunk)
    ctxs = []
    for b in range(images):
        ctx = capi.Context(n, n, 1, capi.make_params(tol=0.0))
        for k, v in opts.items(): ctx.set_option(k, v)
        ctx.set_image([synth.disk(n, 200, 50, noise=16, seed=1000 + b, radius=n // 4 + 8 * (b % 8) - 28)]); ctx.init_checkerboard()
        ctxs.append(ctx)
    for ctx in ctxs: ctx.enqueue_steps(64)
    for ctx in ctxs: ctx.sync()
    t = []
    for r in range(reps):
        for ctx in ctxs: ctx.warm(ck)
        t0 = time.perf_counter()
        done = 0
        while done < steps:
            c = min(ck, steps - done)
            for ctx in ctxs: ctx.enqueue_steps(c)
            done += c
        for ctx in ctxs: ctx.sync()
        t.append((time.perf_counter() - t0) * 1e6 / (steps * images))
    print("%5d^2 x %d images  %-26s %s  median %.2f us per image-iteration   kernel %s" % (n, images, label, " ".join("%.2f" % v for v in t), np.median(t), ctxs[0].launch_info()["kernel"]), flush=True)
    for ctx in ctxs: ctx.close()
