"""Is the wave-end imbalance of a 4096^2 launch PERSISTENT (the same workgroups / CUs / XCDs late in every launch) or noise?
K launches of the 2-pixel kernel with per-wave stamps (debug_times), one iteration each, in one process.  Prints, per launch,
the end-time percentiles, and over the launches: how stable the workgroup -> (XCD, CU) placement is, which share of the variance of
the per-workgroup end time is explained by the workgroup's mean over launches (persistent part), the same per CU and per XCD, and
what the kernel would take if every strip were re-sized by its persistent speed.
usage: N=4096 K=8 python tools/imbalance_probe.py [k=v ...]      SAVE=file.npz keeps the raw end times"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "4096")); K = int(os.environ.get("K", "8"))
ctx = capi.Context(n, n, 1, capi.make_params(tol=0.0))
ctx.set_option("kernel", 3)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
ctx.set_option("debug_times", 1)
ctx.set_image([synth.disk(n)]); ctx.init_checkerboard()
ctx.run(200)                       # far field everywhere, clocks up
L = capi.lib()
L.cvh_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_int)]
buf = np.zeros(2_000_000, dtype=np.uint64); words = C.c_long(0); nbc = C.c_int(0)
ends, durs, place = [], [], []
for it in range(K):
    ctx.run(1)
    L.cvh_debug_read(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size, C.byref(words), C.byref(nbc))
    nb = nbc.value
    w = buf[:nb * 16].reshape(nb * 4, 4).copy()
    ok = w[:, 1] > 0
    t0 = w[ok, 0].min()
    en = np.where(ok, (w[:, 1].astype(np.int64) - np.int64(t0)) / 100.0, np.nan)
    st = np.where(ok, (w[:, 0].astype(np.int64) - np.int64(t0)) / 100.0, np.nan)
    ends.append(en); durs.append(en - st)
    hw = w[:, 3] >> 8; xcc = w[:, 3] & 0xf
    place.append(np.where(ok, xcc * 100000 + ((hw >> 13) & 7) * 10000 + ((hw >> 12) & 1) * 1000 + ((hw >> 8) & 0xf) * 10 + ((hw >> 4) & 3), -1))
    e = en[ok]
    print("launch %d: waves %d  end p10 %.1f p50 %.1f p90 %.1f max %.1f   mean %.2f" % (it, ok.sum(), *np.percentile(e, [10, 50, 90, 100]), e.mean()))
E = np.array(ends); P = np.array(place)
okw = ~np.isnan(E).any(axis=0)
E = E[:, okw]; P = P[:, okw]
print("waves with stamps in every launch:", okw.sum())
print("placement: share of waves on the same (XCD, SE, SH, CU, SIMD) in every launch: %.3f ; same CU: %.3f ; same XCD: %.3f" % (
    (P == P[0]).all(axis=0).mean(), ((P // 10) == (P[0] // 10)).all(axis=0).mean(), ((P // 100000) == (P[0] // 100000)).all(axis=0).mean()))
def explained(keys, name):
    # share of the variance of the end times (pooled over launches, each launch centred) explained by the group's mean over launches
    Ec = E - E.mean(axis=1, keepdims=True)
    u, inv = np.unique(keys, return_inverse=True)
    gm = np.array([np.bincount(inv, weights=Ec[k]) / np.bincount(inv) for k in range(len(Ec))])    # [launch][group] mean
    pers = gm.mean(axis=0)                                          # persistent part per group
    tot = Ec.var()
    exp = (pers[inv] ** 2).mean()
    print("%-22s groups %4d: persistent component std %.2f us (%.0f %% of the end-time variance); launch-to-launch std of a group mean %.2f us" % (
        name, len(u), pers.std(), 100 * exp / tot, gm.std(axis=0).mean()))
    return pers, inv
wg = np.nonzero(okw)[0] // 4
explained(np.nonzero(okw)[0], "per wave")
pw, iw = explained(wg, "per workgroup")
explained(P[0] // 10, "per CU (launch 0 map)")
explained(P[0] // 100000, "per XCD")
# what a static re-sizing by the persistent per-workgroup speed would give: a workgroup that ends d us late loses d / mean of its rows
mean_end = E.mean()
resid = E - (E.mean(axis=1, keepdims=True) + pw[iw][None, :])
print("kernel end now (mean over launches of the last wave): %.2f us ; mean wave end %.2f us" % (E.max(axis=1).mean(), mean_end))
print("after removing the persistent per-workgroup part: last wave %.2f us  (p99 %.2f)" % ((mean_end + resid).max(axis=1).mean(), np.percentile(mean_end + resid, 99)))
if os.environ.get("SAVE"):
    np.savez_compressed(os.environ["SAVE"], E=E, P=P, okw=okw)
ctx.close()
