"""Where a time step of the resident Perona-Malik kernel (pm_resident_kernel.hip) spends its time: per-workgroup stamps around step 5 of
one cooperative launch.  usage: N=2048 python tools/pm_resident_timeline.py"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "2048"))
ctx = capi.Context(n, n, 1)
ctx.set_option("pm_kernel", 4); ctx.set_option("math_mode", int(os.environ.get("MATH", "2")))
for kv in sys.argv[1:]:          # further options as key=value, e.g. res_prio=0
    k, v = kv.split("="); ctx.set_option(k, int(v))
img = synth.disk(n, 200, 50, noise=40, seed=1)
ctx.set_image([img]); ctx.perona_malik(30.0, 0.25, 50.0)          # clocks up
info = ctx.launch_info(1); nt = int(info["grid"])
ctx.set_option("debug_times", 1)
ctx.set_image([img]); ctx.perona_malik(30.0, 0.25, 5.0)
L = capi.lib()
L.cvh_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_int)]
buf = np.zeros(nt * 12 + 64, dtype=np.uint64); words = C.c_long(0); nb = C.c_int(0)
L.cvh_debug_read(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size, C.byref(words), C.byref(nb))
w = buf[:nt * 12].reshape(nt, 12).astype(np.int64)
t0 = w[:, 0].min()
us = lambda x: (x - t0) / 100.0
names = ["step 5 begins", "", "halo ring in LDS (the neighbours' tagged borders polled and gathered)", "band computed (wave 0)", "all waves", "tile rewritten",
         "border stores issued", "", "step 6 begins"]
print("tiles", nt, info)
for k, nm in enumerate(names):
    col = w[:, k]; ok = col > 0
    if not nm or ok.sum() == 0: continue
    v = us(col[ok])
    print("%-40s n %4d  min %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (nm, ok.sum(), v.min(), np.median(v), np.percentile(v, 90), v.max()))
d = lambda a_, b_: np.median((w[:, b_] - w[:, a_]) / 100.0)
print("per workgroup (median): wait for + gather the neighbours' borders %.2f | band (wave 0) %.2f | other waves %.2f | rewrite %.2f | step %.2f us" % (
    d(0, 2), d(2, 3), d(3, 4), d(4, 5), d(0, 8)))
ctx.close()
