"""Where does the resident Perona-Malik flow differ from the per-launch flow?  Prints the differing pixels' rows / columns modulo the tile size.
usage: pm_diff_probe.py [N=2048 STEPS=3 MATH=1]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "2048")); steps = int(os.environ.get("STEPS", "3")); math = int(os.environ.get("MATH", "1"))
img = synth.disk(n, 200, 50, noise=40, seed=2)
out = {}
for pk in (4, 3):
    with capi.Context(n, n, 1) as ctx:
        ctx.set_option("math_mode", math); ctx.set_option("pm_kernel", pk)
        ctx.set_image([img]); ctx.perona_malik(30.0, 0.25, 0.25 * steps)
        out[pk] = ctx.get_image()[0].astype(np.int32); print(pk, ctx.launch_info(1)["kernel"])
d = out[4] != out[3]
print("steps", steps, "differing pixels", int(d.sum()))
if d.any():
    r, c = np.nonzero(d)
    print("rows mod 128:", sorted(set((r % 128).tolist()))[:40])
    print("cols mod 128:", sorted(set((c % 128).tolist()))[:40])
    print("tile rows:", sorted(set((r // 128).tolist())), "tile cols:", sorted(set((c // 128).tolist())))
    print("first few:", list(zip(r[:10].tolist(), c[:10].tolist())))
