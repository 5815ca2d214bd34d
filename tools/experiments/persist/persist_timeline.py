"""Diagnostic: where an iteration boundary of the persistent step kernel spends its time (stamps of iterations 3 -> 4, 100 MHz clock).
usage: N=4096 python tools/persist_timeline.py"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "4096"))
ctx = capi.Context(n, n, 1, capi.make_params(tol=0.0))
ctx.set_option("kernel", 3); ctx.set_option("persist", 1)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
ctx.set_image([synth.disk(n)]); ctx.set_levelset(capi.checkerboard_host(n, n))
ctx.enqueue_steps(200); ctx.sync()
ctx.set_option("debug_times", 1)
ctx.enqueue_steps(8); ctx.sync()
L = capi.lib()
buf = np.zeros(2_000_000, dtype=np.uint64); words = C.c_long(0); nb = C.c_int(0)
L.cvh_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_int)]
L.cvh_debug_read(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size, C.byref(words), C.byref(nb))
nb = nb.value
w = buf[:nb * 10].reshape(nb, 10).astype(np.int64)
t0 = w[:, 0].min()
us = (w - t0) / 100.0
names = ["compute start (it 3)", "rows done", "published (stores + atomics back)", "arrived (atomicAdd back)", "wait A done (+L2 invalidate)",
         "prologue issued", "go seen", "compute start (it 4)"]
for k, nm in enumerate(names):
    col = us[:, k]
    print("%-36s min %7.2f p10 %7.2f p50 %7.2f p90 %7.2f max %7.2f" % (nm, col.min(), np.percentile(col, 10), np.median(col), np.percentile(col, 90), col.max()))
last = np.nonzero(w[:, 8] > 0)[0]
for b in last:
    print("last arriver: block %d: rows done %.2f published %.2f arrived+go written %.2f (go write stamp %.2f)" % (b, us[b, 1], us[b, 2], us[b, 3], us[b, 8]))
d = us[:, 7] - us[:, 0]
print("iteration 3 -> 4 per workgroup: p50 %.2f max %.2f ; span first start .. last start of it 4: %.2f .. %.2f" % (np.median(d), d.max(), us[:, 7].min(), us[:, 7].max()))
print("steps: rows->published p50 %.2f | published->arrived p50 %.2f | go seen - max(arrived) p50 %.2f | go seen->compute p50 %.2f" % (
    np.median(us[:, 2] - us[:, 1]), np.median(us[:, 3] - us[:, 2]), np.median(us[:, 6]) - us[:, 3].max(), np.median(us[:, 7] - us[:, 6])))

# who is slow?  class = dispatch round of the workgroup inside its XCD (class-major numbering assumes 32 CUs per XCD)
info = w[:, 9]
xcc = info & 0xf; hwid = (info >> 8) & 0xffffffff; rows = (info >> 40) & 0xff; lbid = (info >> 48) & 0xffff
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7   # HW_ID: wave 3:0 simd 5:4 pipe 7:6 cu 11:8 sh 12 se 15:13
cukey = xcc * 1000 + se * 100 + sh * 10 + cu
blk = np.arange(nb)
cls = (blk >> 3) // 32
body = us[:, 1] - us[:, 0]
for k in range(int(cls.max()) + 1):
    m = cls == k
    print("class %d: %3d workgroups, rows of wave 0: %s, body us p50 %.2f max %.2f" % (k, m.sum(), np.unique(rows[m])[:6], np.median(body[m]), body[m].max()))
uniq, cnt = np.unique(cukey, return_counts=True)
print("distinct CUs %d, workgroups per CU min/median/max %d %d %d" % (len(uniq), cnt.min(), np.median(cnt), cnt.max()))
per_cu = {}
for b in range(nb): per_cu.setdefault(cukey[b], []).append(b)
mixed = sum(1 for v in per_cu.values() if len(set(cls[v])) != len(v))
print("CUs holding two workgroups of the same class:", mixed)
order = np.argsort(-us[:, 1])[:12]
for b in order:
    print("  slow: block %4d logical %4d class %d xcc %d cu-key %5d rows %3d: body %.2f (start %.2f end %.2f) published %.2f" % (b, lbid[b], cls[b], xcc[b], cukey[b], rows[b], body[b], us[b, 0], us[b, 1], us[b, 2]))
