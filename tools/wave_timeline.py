"""Diagnostic: per-wave start/end stamps of one launch of the wave kernel (100 MHz realtime)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(__import__("os").environ.get("N", "4096"))
C_ = int(__import__("os").environ.get("C", "1"))     # C=3: the 3-channel disks of config C3
ctx = capi.Context(n, n, C_, capi.make_params(tol=0.0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1]) if C_ == 3 else capi.make_params(tol=0.0))
ctx.set_option("kernel", int(__import__("os").environ.get("KERNEL", "2")))
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
ctx.set_option("debug_times", 1)
ctx.set_image(synth.config_planes("C3", n) if C_ == 3 else [synth.disk(n)]); ctx.set_levelset(capi.checkerboard_host(n, n))
ctx.run(int(__import__('os').environ.get('ITERS', '5')))
L = capi.lib()
buf = np.zeros(2_000_000, dtype=np.uint64); words = C.c_long(0); nb = C.c_int(0)
L.cvh_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_int)]
L.cvh_debug_read(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size, C.byref(words), C.byref(nb))
nb = nb.value
w = buf[:nb * 16].reshape(nb * 4, 4)
blk_end = buf[nb * 16: nb * 17]
ok = w[:, 1] > 0
t0 = w[ok, 0].min()
st = (w[ok, 0] - t0) / 100.0; en = (w[ok, 1] - t0) / 100.0   # microseconds
print("blocks", nb, "waves stamped", ok.sum())
print("start us: min %.2f p50 %.2f p90 %.2f max %.2f" % (st.min(), np.median(st), np.percentile(st, 90), st.max()))
print("end   us: min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f" % (en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
print("dur   us: min %.2f p50 %.2f max %.2f" % ((en - st).min(), np.median(en - st), (en - st).max()))
print("block end (after publish/finalize) max us %.2f" % ((blk_end[blk_end > 0].max() - t0) / 100.0))
tick = buf[nb * 17: nb * 18]; lastw = buf[nb * 18: nb * 18 + 4]; stored = buf[nb * 19: nb * 20]
lb = int(np.argmax(tick))
wend_lb = w[lb * 4: lb * 4 + 4, 1].max()
print("last block %d: waves end %.2f | partial stored %.2f | ticket back %.2f | finalize start %.2f end %.2f | block end %.2f" % (
    lb, (wend_lb - t0) / 100.0, (stored[lb] - t0) / 100.0, (tick[lb] - t0) / 100.0, (lastw[0] - t0) / 100.0, (lastw[1] - t0) / 100.0, (blk_end[lb] - t0) / 100.0))
d1 = (stored.astype(np.int64) - w[:, 1].reshape(nb, 4).max(axis=1).astype(np.int64)) / 100.0
d2 = (tick.astype(np.int64) - stored.astype(np.int64)) / 100.0
print("finalize: partial rows loaded %.2f, reduced %.2f" % ((lastw[2] - t0) / 100.0, (lastw[3] - t0) / 100.0))
print("per block: waves end -> stored: p50 %.2f max %.2f ; stored -> ticket: p50 %.2f max %.2f" % (np.median(d1), d1.max(), np.median(d2), d2.max()))
xcc = w[ok, 3] & 0xf
hw = w[ok, 3] >> 8
first = (w[ok, 2].astype(np.int64) - np.int64(t0)) / 100.0
print('first group starts us: min %.2f p50 %.2f p90 %.2f max %.2f (kernel 3 only)' % (first.min(), np.median(first), np.percentile(first, 90), first.max()))
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc,se,sh,cu):", len(u), "waves per CU min/median/max:", cnt.min(), int(np.median(cnt)), cnt.max())
hist, edges = np.histogram(en, bins=12)
print("end-time histogram:", list(zip(np.round(edges[:-1], 1), hist)))
late = st > 5
print("waves starting later than 5 us:", late.sum())
# where do the late waves live?
okidx = np.nonzero(ok)[0]
blk = okidx // 4; wv = okidx % 4
nwc = (n + 62) // 63; nbc = (nwc + 3) // 4
strip = blk // nbc; wcol = (blk % nbc) * 4 + wv
def grp(name, keyv):
    u_, inv = np.unique(keyv, return_inverse=True)
    m = np.bincount(inv, weights=en) / np.bincount(inv)
    print(name, "groups", len(u_), "mean end: min %.1f p50 %.1f max %.1f" % (m.min(), np.median(m), m.max()), "| first 8:", np.round(m[:8], 1), "last 4:", np.round(m[-4:], 1))
grp("by xcc", xcc); grp("by CU", key); grp("by strip", strip); grp("by wave column", wcol)
simd = (hw >> 4) & 3
grp("by simd", simd)
percu = np.bincount(np.unique(key, return_inverse=True)[1])
u_, inv = np.unique(key, return_inverse=True)
mcu = np.bincount(inv, weights=en) / np.bincount(inv)
for c_ in sorted(set(percu)): print("CUs with", c_, "waves:", (percu == c_).sum(), "mean end %.1f" % mcu[percu == c_].mean())
# spread inside a CU
mx = np.zeros(len(u_)); mn = np.full(len(u_), 1e9)
np.maximum.at(mx, inv, en); np.minimum.at(mn, inv, en)
print("within-CU end spread: median %.1f max %.1f; CU last-end: min %.1f median %.1f max %.1f" % (np.median(mx - mn), (mx - mn).max(), mx.min(), np.median(mx), mx.max()))
out = __import__("os").environ.get("SAVE")
if out:   # raw stamps for offline analysis: per wave {start, end, HW_ID, XCC_ID}, block ends, ticket / stored stamps
    np.savez_compressed(out, w=w, blk_end=blk_end, tick=tick, lastw=lastw, stored=stored, nb=nb, n=n)
ctx.close()
