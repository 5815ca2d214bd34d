"""where do the resident and the per-launch Perona-Malik flows differ?  usage: dbg_pm_resident.py H W STEPS [MATH]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi
h, w, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); math = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(5 + h + 3 * w)
img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
def run(k):
    with capi.Context(h, w, 1) as ctx:
        ctx.set_option("math_mode", math); ctx.set_option("pm_kernel", k)
        ctx.set_image([img]); ctx.perona_malik(30, 0.25, 0.25 * steps)
        return ctx.get_image()[0], ctx.launch_info(1)
a, ia = run(4); b, ib = run(3)
print(ia)
d = np.argwhere(a != b)
print(len(d), "differ"); print(d[:60].tolist())
