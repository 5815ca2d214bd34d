"""Host logic of the wave kernels' launch geometry, on CPU: the strip table the library uploads (api.hip,
compute_strip_bounds, reached through the diagnostic export cvh_debug_strip_bounds -- no device needed) and a Python
restatement of the class-major workgroup numbering the kernels apply (csv_wave2_kernel.hip)."""
import ctypes as C

import numpy as np
import pytest

from chan_vese_amd import capi


def bounds(kind, h, tiles_x, S, strip_rows, nblocks, cls, cskew, skew=0):
    L = capi.lib()
    fn = L.cvh_debug_strip_bounds
    fn.restype = C.c_int
    fn.argtypes = [C.c_int] * 9 + [C.POINTER(C.c_int)]
    out = (C.c_int * (S + 1))()
    assert fn(kind, h, tiles_x, S, strip_rows, nblocks, cls, cskew, skew, out) == 0
    return np.array(out[:], dtype=np.int64)


def class_major_rank(bid, nb, S):
    """csv_wave2_kernel.hip: rank of workgroup `bid` when all workgroups of dispatch round 0 come first, XCD by XCD."""
    x, j, q, r = bid & 7, bid >> 3, nb >> 3, nb & 7
    cl, rank = j // S, 0
    for xx in range(8):
        nx = q + (1 if xx < r else 0)
        rank += min(nx, cl * S) + (max(0, min(S, nx - cl * S)) if xx < x else 0)
    return rank + (j - cl * S), cl


@pytest.mark.parametrize("nb", [1, 7, 8, 9, 96, 255, 256, 257, 468, 510, 765, 1275, 3000])
def test_class_major_numbering_is_a_permutation(nb):
    ranks = [class_major_rank(b, nb, 32) for b in range(nb)]
    assert sorted(r for r, _ in ranks) == list(range(nb))
    # ranks are ordered by dispatch round: every workgroup of round c precedes every workgroup of round c + 1
    by_rank = sorted(ranks)
    assert [c for _, c in by_rank] == sorted(c for _, c in by_rank)
    # inside a round an XCD's workgroups are contiguous (its L2 sees neighbouring wave-columns / strips)
    for cl in set(c for _, c in ranks):
        seq = [b & 7 for _, b in sorted((class_major_rank(b, nb, 32)[0], b) for b in range(nb) if class_major_rank(b, nb, 32)[1] == cl)]
        assert seq == sorted(seq)


def test_strip_table_of_the_4096_bench_geometry():
    """BASELINE configs[1]: 33 wave-columns of 126, 90 strips, 765 workgroups, 32 workgroups per XCD per round, skew 0.5."""
    h, tiles_x, S, nb = 4096, 33, 90, 765
    b = bounds(3, h, tiles_x, S, 46, nb, 32, 500)
    assert b[0] == 0 and b[-1] == h and np.all(np.diff(b) > 0)
    length = np.diff(b)
    # three dispatch rounds of 15 strip pairs: 1.5 : 1 : 0.5 of the mean length
    assert abs(length[:30].mean() / length[30:60].mean() - 1.5) < 0.03 and abs(length[60:].mean() / length[30:60].mean() - 0.5) < 0.03
    # the two strips of a workgroup march in lock step (one barrier per group of 4 rows): same length up to rounding
    assert np.all(np.abs(length[0::2] - length[1::2]) <= 1)
    # no skew: equal strips, no short last strip
    b0 = bounds(3, h, tiles_x, S, 46, nb, 32, 0)
    assert set(np.diff(b0)) <= {45, 46}


@pytest.mark.parametrize("kind,h,tiles_x,S,strip_rows,nb", [(3, 2048, 17, 103, 20, 468), (3, 150, 5, 19, 8, 30), (3, 1, 2, 1, 8, 1),
                                                           (2, 4096, 66, 75, 55, 1275), (2, 37, 1, 5, 8, 5)])
@pytest.mark.parametrize("cls,cskew", [(0, 0), (32, 0), (32, 500), (32, 900)])
def test_strip_table_covers_the_image(kind, h, tiles_x, S, strip_rows, nb, cls, cskew):
    b = bounds(kind, h, tiles_x, S, strip_rows, nb, cls, cskew)
    assert b[0] == 0 and b[-1] == h and np.all(np.diff(b) >= 0)      # monotone cover of rows [0, h): the kernels' exit test relies on it
    assert len(b) == S + 1


def data_flow(h, w, channels=1, math_mode=2, kernel=-1, state=64, cus=256):
    L = capi.lib()
    fn = L.cvh_debug_data_flow
    fn.restype = C.c_int
    fn.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int)] * 4
    out = [C.c_int(-1) for _ in range(4)]
    assert fn(h, w, channels, math_mode, kernel, state, cus, *[C.byref(o) for o in out]) == 0
    return tuple(o.value for o in out)          # (flow, tiles_x, tiles_y, strip_rows)


def test_which_per_launch_flow_runs():
    """resolve_geometry() (api.hip) on the host: 3 = wave kernel with 2 pixels per lane, 2 = wave kernel, 0 = tile kernel.  The wave
    kernels address the level set with 32-bit byte offsets and mark dropped lanes with offset 2^31, so planes of 2^28 pixels (2 GiB of
    level set) or more take the tile kernel -- a dispatch no GPU test launches (a 2 GiB level-set pair + planes per context)."""
    assert data_flow(4096, 4096)[0] == 3                              # BASELINE configs[1]
    assert data_flow(4096, 4096, channels=3)[0] == 3                  # configs[2], FAST
    assert data_flow(4096, 4096, channels=3, math_mode=1)[0] == 2     # STRICT three channels: 1-pixel kernel
    assert data_flow(4096, 4100)[0] == 2                              # width not a multiple of 16
    assert data_flow(512, 512)[0] == 2                                # below 0.6 Mpixel (where the resident flow does not take over)
    assert data_flow(512, 512, kernel=3)[0] == 3 and data_flow(512, 512, state=32)[0] == 3      # on request / FP32 state
    assert data_flow(16384, 16368)[0] == 3                            # 2^28 - 2^18 pixels: still the wave kernel
    assert data_flow(16384, 16384)[0] == 0                            # 2^28 pixels: the tile kernel
    assert data_flow(16384, 16384, kernel=3)[0] == 0 and data_flow(16384, 16384, kernel=2)[0] == 0 and data_flow(20000, 20000, channels=3)[0] == 0
    flow, tx, ty, rows = data_flow(16384, 16384)
    assert tx * ty > 0 and rows in (14, 16) and ty * rows >= 16384    # the tile grid covers the plane
    # the bench geometry: 33 wave-columns of 126, 90 strips (tests above pin the strip table itself)
    assert data_flow(4096, 4096)[1:3] == (33, 90)
