"""The one-reciprocal near form of H_eps (chan_vese_amd/csrc/wave_math.h, heaviside_centred_near1; round 4) restated in numpy over the
library's own table (cvh_fill_atan3_table, no device needed): atan(a) = atan(c) + atan((a - c) / (1 + a c)) with c = a truncated to six
mantissa bits read from a table indexed by exponent and top mantissa bits, the index clamped below to c = 0 and above to c = 64 -- valid
for ANY argument.  Checked against atan(a) / pi in extended precision: the form itself (table, clamps, series length), not the GPU's
last-bit rounding (the fused kernels are held to the oracle by the parity tests)."""
import ctypes as C

import numpy as np

from chan_vese_amd import capi

EMIN, EXPS = -6, 12
N = 1 + 64 * EXPS + 1


def table():
    L = capi.lib()
    out = np.zeros(2 * N)
    L.cvh_fill_atan3_table.restype = None
    L.cvh_fill_atan3_table.argtypes = [C.POINTER(C.c_double)]
    L.cvh_fill_atan3_table(out.ctypes.data_as(C.POINTER(C.c_double)))
    return out.reshape(N, 2)


def near1(a, tab):
    """numpy restatement of heaviside_centred_near1 for a = |u| / eps >= 0 (returns atan(a) / pi)."""
    a = np.minimum(np.asarray(a, dtype=np.float64), 1e300)
    hi = (a.view(np.uint64) >> np.uint64(32)).astype(np.int64)
    key1 = 64 * (1023 + EMIN)
    idx = np.clip((hi >> 14) - (key1 - 1), 0, N - 1)
    c, t = tab[idx, 0], tab[idx, 1]
    z = (a - c) / (1.0 + a * c)
    z2 = z * z
    p = z2 * (-1.0 / 7.0) + 0.2
    p = z2 * p - 1.0 / 3.0
    az = (z * z2) * p + z
    return az / np.pi + t, z, idx


def test_table_is_what_the_form_assumes():
    tab = table()
    assert tab[0, 0] == 0 and tab[0, 1] == 0
    assert tab[1, 0] == 2.0 ** EMIN and tab[N - 1, 0] == 2.0 ** (EMIN + EXPS) == 64
    c = tab[1:, 0]
    assert np.all(np.diff(c) > 0)
    m, e = np.frexp(c)                                     # c = 2^E (1 + m / 64): six mantissa bits
    assert np.all((m * 128) == np.round(m * 128))
    want = np.arctan(c.astype(np.longdouble)) / np.longdouble(np.pi)   # (x86-64 long double: 64-bit mantissa)
    assert np.abs(tab[1:, 1] - want.astype(np.float64)).max() <= 1.2e-16


def test_one_reciprocal_near_form_is_accurate_for_any_argument():
    tab = table()
    rng = np.random.default_rng(4)
    a = np.concatenate([
        10.0 ** rng.uniform(-12, 6, 200000), rng.uniform(0, 70, 200000), rng.uniform(0, 2, 100000),
        tab[:, 0], np.nextafter(tab[1:, 0], 0), np.nextafter(tab[:, 0], np.inf),               # the bin edges and their neighbours
        [0.0, 5e-324, 2.0 ** -1022, 2.0 ** -7, 2.0 ** -6, 63.999999, 64.0, 64.000001, 1e3, 1e17, 1e300, 1e308, np.inf]])
    got, z, idx = near1(a, tab)
    assert np.abs(z).max() <= 1.0 / 64 + 1e-18 and np.abs(z[(a >= 2.0 ** EMIN) & (a < 64)]).max() <= 1.0 / 128 + 1e-18
    want = (np.arctan(np.minimum(a, 1e300).astype(np.longdouble)) / np.longdouble(np.pi)).astype(np.float64)
    err = np.abs(got - want)
    assert err.max() <= 2.3e-16, (err.max(), a[err.argmax()])     # ~2 ulp of 0.5: table entry + the final sum, as the two-reciprocal form
    # first bin (c = 0): the plain series; last entry (c = 64): everything beyond the table
    assert np.all(idx[a < 2.0 ** EMIN] == 0) and np.all(idx[a >= 64] == N - 1)
