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
