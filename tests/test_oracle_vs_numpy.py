"""Cross-check: C oracle vs the independent numpy restatement (tests/np_restatement.py)."""
import numpy as np
import pytest

import np_restatement as R
from chan_vese_amd import synth


def test_checkerboard_identical(oracle):
    assert np.array_equal(oracle.checkerboard(97, 131), R.checkerboard(97, 131))


def test_curvature_matches(oracle):
    rng = np.random.default_rng(0)
    u = rng.normal(scale=3.0, size=(37, 53))
    assert np.allclose(oracle.curvature(u), R.curvature(u), rtol=0, atol=5e-15)


@pytest.mark.parametrize("shape", [(37, 53), (64, 64), (1, 40), (40, 1), (2, 2)])
def test_csv_steps_match(oracle, shape):
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    u_o = oracle.checkerboard(h, w) if min(h, w) > 2 else rng.normal(size=shape)
    u_n = u_o.copy()
    p = oracle.make_params(mu=0.3, nu=0.01, dt=0.5, eps=0.7, tol=0)
    for _ in range(5):
        nrm_o, c1, c2 = oracle.csv_step([img], u_o, p)
        u_n, nrm_n, c1n, c2n = R.csv_step([img], u_n, mu=0.3, nu=0.01, dt=0.5, eps=0.7)
        assert c1[0] == pytest.approx(c1n[0], rel=1e-12) and c2[0] == pytest.approx(c2n[0], rel=1e-12)
        assert nrm_o == pytest.approx(nrm_n, rel=1e-10)
        assert np.allclose(u_o, u_n, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(u_n).max()))


def test_csv_three_channel_match(oracle):
    h, w = 48, 40
    planes = [synth.disk(48, 180, 40, h=h, w=w), synth.disk(48, 200, 60, h=h, w=w),
              synth.disk(48, 60, 200, h=h, w=w)]
    l1, l2 = [1, 1, 0.5], [1, 0.5, 1]
    p = oracle.make_params(tol=0, lambda1=l1, lambda2=l2)
    u_o = oracle.checkerboard(h, w)
    u_n = u_o.copy()
    for _ in range(6):
        nrm_o, c1, c2 = oracle.csv_step(planes, u_o, p)
        u_n, nrm_n, c1n, c2n = R.csv_step(planes, u_n, lambda1=l1, lambda2=l2)
        assert np.allclose(c1, c1n, rtol=1e-12) and np.allclose(c2, c2n, rtol=1e-12)
        assert np.allclose(u_o, u_n, rtol=1e-9, atol=1e-9 * np.abs(u_n).max())


def test_stop_condition_match(oracle):
    planes = [synth.disk(40, 180, 40), synth.disk(40, 200, 60), synth.disk(40, 60, 200)]
    assert oracle.stop_condition(planes, 1e-3) == pytest.approx(R.stop_condition(planes, 1e-3), rel=1e-13)


def test_perona_malik_match(oracle):
    img = synth.disk(48, 200, 50, noise=32, seed=1, h=40, w=56)
    trips = oracle.pm_trip_count(0.25, 5)
    out_o, st_o = oracle.perona_malik([img], 30, 0.25, 5, want_state=True)
    out_n, st_n = R.perona_malik(img, 30, 0.25, trips)
    assert np.allclose(st_o[0], st_n, rtol=0, atol=1e-10)
    assert (out_o[0] != out_n).sum() == 0


def test_video_contour_matches(oracle):
    """The frame contour of -V (single-sourced in the oracle until round 2): numpy restatement vs the C oracle on a
    level set with holes, components touching the border, half-integer values (round half to even) and NaN."""
    rng = np.random.default_rng(21)
    u = rng.normal(scale=2.0, size=(41, 57))
    u[10:20, 10:30] = 3.0
    u[13:16, 14:18] = -1.0          # a hole
    u[0:5, 40:57] = 2.0             # touches the top/right border
    u[30, 5:9] = [0.5, 1.5, 2.5, -0.5]   # half-even: 0, 2, 2, -0
    u[35, 35] = np.nan
    assert np.array_equal(oracle.video_contour(u), R.video_contour(u))
    for shape in [(3, 3), (1, 9), (9, 1), (2, 7)]:
        v = rng.normal(scale=2.0, size=shape)
        assert np.array_equal(oracle.video_contour(v), R.video_contour(v))


def test_rect_levelset_matches(oracle):
    for (h, w, x, y, rw, rh) in [(24, 64, 5, 3, 20, 10), (24, 64, 50, 15, 30, 30), (10, 10, 0, 0, 10, 10), (10, 10, 3, 3, 1, 1)]:
        assert np.array_equal(oracle.levelset_rect(h, w, x, y, rw, rh), R.levelset_rect(h, w, x, y, rw, rh))


def test_combine_is_the_addweighted_fold_not_the_unfused_expression(oracle):
    """src/main.cpp:985 `dt*(mu*kappa - nu + u_diff/N)`: the oracle (and the HIP kernels) evaluate OpenCV's MatExpr FOLD,
    one addWeighted(K, dt*mu, U, dt/N, -dt*nu).  This is an ASSUMPTION about OpenCV 2.4 that cannot be verified offline
    (DESIGN.md §2); the test makes it explicit: for dt != 1, N = 3 the folded and the operation-by-operation results differ
    in the last bits on most pixels, the oracle reproduces the fold bit for bit, and the two stay within 4 ulp of each other
    (far below every tolerance in this repo), so the assumption cannot change any parity verdict."""
    h, w = 23, 31
    rng = np.random.default_rng(8)
    planes = [rng.integers(0, 256, size=(h, w), dtype=np.uint8) for _ in range(3)]
    u = rng.normal(scale=4.0, size=(h, w))
    mu, nu, dt, eps = 0.37, 0.013, 0.3, 1.0
    l1, l2 = [1.0, 0.7, 0.5], [0.9, 0.5, 1.0]
    p = oracle.make_params(mu=mu, nu=nu, dt=dt, eps=eps, tol=0, lambda1=l1, lambda2=l2)
    # u_diff of :979 from the oracle's own pieces (serial channel order)
    ud = np.zeros((h, w))
    for k in range(3):
        c1 = oracle.region_mean(planes[k], u, 0, eps)
        c2 = oracle.region_mean(planes[k], u, 1, eps)
        ud += oracle.variance_penalty(planes[k], c2, l2[k]) - oracle.variance_penalty(planes[k], c1, l1[k])
    kappa = oracle.curvature(u)
    folded = R.combine_folded(kappa, ud, mu, nu, dt, 3) * R.delta(u, eps)
    unfused = R.combine_unfused(kappa, ud, mu, nu, dt, 3) * R.delta(u, eps)
    u_o = u.copy()
    oracle.csv_step(planes, u_o, p)
    got = u_o - u                      # exact only up to the final u += u_diff rounding: compare u + step instead
    assert np.array_equal(u_o, u + folded)
    assert not np.array_equal(u_o, u + unfused)
    assert (folded != unfused).mean() > 0.2
    assert np.abs(folded - unfused).max() <= 4 * np.spacing(np.abs(folded).max())
