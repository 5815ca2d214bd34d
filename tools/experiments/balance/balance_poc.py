"""Proof of concept: strip lengths per workgroup from MEASURED wave durations (tools/imbalance_probe.py: 76-88 % of the wave-end
spread of a 4096^2 launch is persistent per workgroup).  K stamped launches -> per (workgroup column, strip) mean end time ->
rows re-dealt inside every column in proportion to the measured speed -> table installed with cvh_debug_set_strip_table ->
HIP-event time per iteration before / after, alternating, in ONE context.
usage: N=4096 ROUNDS=3 K=4 python tools/balance_poc.py [k=v ...]"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
from chan_vese_amd import capi, synth
n = int(os.environ.get("N", "4096")); K = int(os.environ.get("K", "4")); ROUNDS = int(os.environ.get("ROUNDS", "3"))
OVH = float(os.environ.get("OVH", "6"))        # fixed cost of a strip in row-equivalents (prologue: 7 rows of loads before the first row)
ctx = capi.Context(n, n, 1, capi.make_params(tol=0.0))
ctx.set_option("kernel", 3)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
ctx.set_image([synth.disk(n)]); ctx.init_checkerboard()
ctx.run(300)
L = capi.lib()
L.cvh_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_int)]
L.cvh_debug_set_strip_table.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int]
L.cvh_debug_strip_bounds.argtypes = [C.c_int] * 9 + [C.POINTER(C.c_int)]
info = ctx.launch_info()
nwc, S, sr = int(info["wave_columns"]), int(info["strips"]), int(info["strip_rows"])
nbc = (nwc + 1) // 2
nblocks = nbc * ((S + 1) // 2)
base = np.zeros(S + 1, dtype=np.int32)
cskew = 500 if sr <= 46 else (0 if sr >= 128 else int(500 * (128.0 - sr) / (128.0 - 46.0)))
L.cvh_debug_strip_bounds(3, n, nwc, S, sr, nblocks, 32, cskew, 0, base.ctypes.data_as(C.POINTER(C.c_int)))
table = np.tile(base, (nbc, 1)).astype(np.int32)            # [column][strip]: the library's own table to start from

def timeit(steps=112):
    ctx.warm(steps); ctx.enqueue_steps(16); ctx.sync()
    ctx.warm(steps); ctx.enqueue_steps(steps); ctx.sync()
    return ctx.last_run_ms() * 1e3 / steps

def stamps():
    """K stamped launches: per (column, strip) mean wave end, and per (column, strip) the END OF ITS CU (last wave of the CU the
    strip's workgroup ran on, mean over launches) -- a wave's own end time says little about speed (the SIMD serves its oldest wave
    first), the CU's end does."""
    buf = np.zeros(2_000_000, dtype=np.uint64); words = C.c_long(0); nb_ = C.c_int(0)
    acc = np.zeros((nbc, S)); cnt = np.zeros((nbc, S)); cuend = np.zeros((nbc, S)); last = []
    for it in range(K):
        ctx.run(1)
        L.cvh_debug_read(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size, C.byref(words), C.byref(nb_))
        nb = nb_.value
        w = buf[:nb * 16].reshape(nb * 4, 4)
        ok = w[:, 1] > 0
        t0 = w[ok, 0].min()
        en = (w[:, 1].astype(np.int64) - np.int64(t0)) / 100.0
        bid = ((w[:, 3] >> np.uint64(40)) & np.uint64(0xffffff)).astype(np.int64)
        hw = (w[:, 3] >> np.uint64(8)) & np.uint64(0xffffffff); xcc = w[:, 3] & np.uint64(0xf)
        cu = (xcc * np.uint64(1000) + ((hw >> np.uint64(13)) & np.uint64(7)) * np.uint64(100) + ((hw >> np.uint64(12)) & np.uint64(1)) * np.uint64(16) + ((hw >> np.uint64(8)) & np.uint64(0xf))).astype(np.int64)
        wave = np.arange(nb * 4) % 4
        col = bid % nbc; strip = (bid // nbc) * 2 + (wave >> 1)
        m = ok & (strip < S)
        np.add.at(acc, (col[m], strip[m]), en[m]); np.add.at(cnt, (col[m], strip[m]), 1)
        u, inv = np.unique(cu[m], return_inverse=True)
        e_cu = np.zeros(len(u)); np.maximum.at(e_cu, inv, en[m])
        np.add.at(cuend, (col[m], strip[m]), e_cu[inv])
        last.append(en[ok].max())
    return acc / np.maximum(cnt, 1), cnt, float(np.mean(last)), cuend / np.maximum(cnt, 1)

def install(t):
    t = np.ascontiguousarray(t, dtype=np.int32)
    rc = L.cvh_debug_set_strip_table(ctx._h, t.ctypes.data_as(C.POINTER(C.c_int)), nbc, S)
    assert rc == 0, L.cvh_last_error(ctx._h)

print("geometry: %d wave columns (%d workgroup columns), %d strips of ~%d rows" % (nwc, nbc, S, sr))
t_base = [timeit() for _ in range(3)]
print("library table: %s us" % " ".join("%.2f" % v for v in t_base))
ctx.set_option("debug_times", 1)
MODE = os.environ.get("MODE", "cu")
for rnd in range(ROUNDS):
    install(table)
    T, cnt, last, E = stamps()
    rows = np.diff(table, axis=1).astype(float)
    print("round %d: measured last wave %.2f us; strip end times: mean %.2f std %.2f min %.2f max %.2f; CU ends: mean %.2f std %.2f max %.2f" % (
        rnd, last, T[cnt > 0].mean(), T[cnt > 0].std(), T[cnt > 0].min(), T[cnt > 0].max(), E[cnt > 0].mean(), E[cnt > 0].std(), E[cnt > 0].max()))
    new = np.zeros_like(table)
    target = E[cnt > 0].mean()
    for k in range(nbc):
        if MODE == "cu":      # every strip scaled by how early / late ITS CU finished (all strips of a CU by the same factor)
            f = np.clip(target / np.maximum(E[k], 1.0), 0.85, 1.15) ** float(os.environ.get("DAMP", "0.8"))
            want = (rows[k] + OVH) * f - OVH
        else:                 # (first attempt, kept for the record: by the strip's own end time -- over-corrects, made things worse)
            speed = (rows[k] + OVH) / np.maximum(T[k], 1.0)
            want = 0.7 * (speed / speed.sum() * (n + OVH * S) - OVH) + 0.3 * rows[k]
        want = np.maximum(want, 8.0)
        cum = np.concatenate([[0.0], np.cumsum(want)]) * (n / want.sum())
        b = np.round(cum).astype(np.int64); b[0] = 0; b[-1] = n
        new[k] = np.maximum.accumulate(b)
    table = new.astype(np.int32)
ctx.set_option("debug_times", 0)
res_b, res_a = [], []
for rep in range(4):
    install(table); res_a.append(timeit())
    rc = L.cvh_debug_set_strip_table(ctx._h, None, 0, 0); res_b.append(timeit())
print("balanced table : %s  median %.2f us" % (" ".join("%.2f" % v for v in res_a), np.median(res_a)))
print("library table  : %s  median %.2f us" % (" ".join("%.2f" % v for v in res_b), np.median(res_b)))
rows = np.diff(table, axis=1)
print("balanced rows per strip: min %d p10 %d p50 %d p90 %d max %d" % (rows.min(), *np.percentile(rows, [10, 50, 90]).astype(int), rows.max()))
if os.environ.get("SAVE"):
    np.save(os.environ["SAVE"], table)
ctx.close()
