"""Where an iteration of the resident kernel (csv_resident_kernel.hip) spends its time: per-workgroup stamps around iteration 3 of one
cooperative launch.  usage: N=2048 python tools/resident_timeline.py"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "2048"))
ctx = capi.Context(n, n, 1, capi.make_params(tol=0.0))
ctx.set_option("resident", 1)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
info = ctx.launch_info(); assert info["kernel"].startswith("csv_resident_kernel<"), info
nt = int(info["grid"])
ctx.set_image([synth.disk(n)]); ctx.init_checkerboard()
ctx.run(100)
ctx.set_option("debug_times", 1)
ctx.run(8)
L = capi.lib()
L.cvh_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_int)]
buf = np.zeros(256 * 12 + 64, dtype=np.uint64); words = C.c_long(0); nb = C.c_int(0)
L.cvh_debug_read(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size, C.byref(words), C.byref(nb))
w = buf[:nt * 12].reshape(nt, 12).astype(np.int64)
t0 = w[:, 0].min()
us = lambda x: (x - t0) / 100.0
names = ["released into it 3", "table in LDS, band borders read", "march done (wave 0)", "all waves done, sums reduced (arrival lines stored here)", "borders in memory (border signal stored here)",
         "master: everybody has arrived", "master: norm and means known", "master: release issued", "released into it 4"]
print("tiles", nt, info)
for k, nm in enumerate(names):
    col = w[:, k]; ok = col > 0
    if ok.sum() == 0: continue
    v = us(col[ok])
    print("%-34s n %4d  min %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (nm, ok.sum(), v.min(), np.median(v), np.percentile(v, 90), v.max()))
d = lambda a_, b_: np.median((w[:, b_] - w[:, a_]) / 100.0)
print("per workgroup (median): set-up %.2f | march (wave 0) %.2f | other waves + reduce %.2f | border stores complete %.2f us" % (d(0, 1), d(1, 2), d(2, 3), d(3, 4)))
print("last tile's arrival stored %.2f -> master sees all arrivals %.2f -> norm / means %.2f -> release issued %.2f -> seen by the others: p50 %.2f max %.2f ; iteration period (p50) %.2f us" % (
    us(w[:, 3]).max(), us(w[0, 5]), us(w[0, 6]), us(w[0, 7]), np.median(us(w[:, 8])), us(w[:, 8]).max(), np.median((w[:, 8] - w[:, 0]) / 100.0)))
print("workgroup 0 (the master's own tile): released %.2f | table %.2f | march (wave 0) %.2f | all waves %.2f | borders %.2f | released into the next %.2f us" % tuple(us(w[0, k]) for k in (0, 1, 2, 3, 4, 8)))
mw = buf[256 * 12:256 * 12 + 16].astype(np.int64)
if mw[:8].max() > 0:
    print("master's waves, share of the arrivals complete at: " + "  ".join("w%d %.2f (%d rounds; its tiles arrived by %.2f)" % (
        k, us(mw[k]), mw[8 + k], us(w[32 * k:32 * k + 32, 3]).max() if 32 * k < nt else 0.0) for k in range(8)))
mw = buf[256 * 12 + 16:256 * 12 + 40].astype(np.int64)
for k, t in enumerate((0, 100, 255)):
    if t < nt and mw[8 * k:8 * k + 8].max() > 0:
        print("tile %3d: march done by wave  " % t + "  ".join("w%d %.2f" % (v, us(mw[8 * k + v])) for v in range(8)) + "   (released %.2f)" % us(w[t, 0]))
late = np.argsort(-w[:, 3])[:5]
print("last five arrivals: " + ", ".join("tile %d at %.2f (released %.2f)" % (t, us(w[t, 3]), us(w[t, 0])) for t in late))
ctx.close()
