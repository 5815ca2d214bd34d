"""Perona-Malik on a resident plane (pm_resident_kernel.hip, option "pm_kernel" = 4; the automatic flow for planes that qualify): the
FP64 state of a channel stays in the LDS of the CUs for all time steps, only the tiles' borders cross workgroups.  Checked against the
oracle (src/main.cpp:478-560 restated) -- STRICT bit-exact on the uint8 planes, FAST <= 1 LSB on <= 1e-6 of the pixels -- and against the
per-launch flow, whose doubles it must reproduce exactly."""
import numpy as np
import pytest

from chan_vese_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from chan_vese_amd import capi as m
    m.lib()
    return m


def run_pm(capi, planes, K, L, T, math, **opts):
    h, w = planes[0].shape
    with capi.Context(h, w, len(planes)) as ctx:
        ctx.set_option("math_mode", math)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_image(planes)
        ctx.perona_malik(K, L, T)
        return ctx.get_image(), ctx.launch_info(1)


# one tile; two tile rows; ragged last tile column of 2 / 116 / 4 columns; odd tile-row heights; more tile rows than columns
SHAPES = [(16, 16), (40, 56), (64, 64), (37, 130), (70, 372), (128, 128), (130, 256), (333, 260), (600, 132)]


@pytest.mark.parametrize("K,L,T", [(30, 0.25, 0.25), (30, 0.25, 0.5), (10, 0.25, 5), (1000, 0.1, 1.5), (20, 0.2, 1.0)])
@pytest.mark.parametrize("shape", SHAPES)
def test_pm_resident_against_the_oracle(capi, oracle, shape, K, L, T):
    rng = np.random.default_rng(5 + shape[0] + 3 * shape[1])
    planes = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(3)]
    cpu = oracle.perona_malik(planes, K, L, T)
    for math in (1, 2):
        gpu, info = run_pm(capi, planes, K, L, T, math, pm_kernel=4)
        assert info["kernel"].startswith("pm_resident_kernel<false, " if math == 1 else "pm_resident_kernel<true, ")
        assert int(info["trips"]) == oracle.pm_trip_count(L, T)
        for g, c in zip(gpu, cpu):
            if math == 1:
                assert (g != c).sum() == 0
            else:
                d = np.abs(g.astype(int) - c.astype(int))
                assert d.max() <= 1 and (d != 0).sum() <= max(1, int(1e-6 * d.size))


@pytest.mark.parametrize("shape", [(64, 64), (70, 372), (333, 260), (512, 512)])
def test_pm_resident_reproduces_the_per_launch_flow(capi, shape):
    """Same operations in the same order: both flavours give the per-launch kernels' bytes, smooth images (where few values sit near a
    rounding boundary) and noisy ones, 23 steps."""
    h, w = shape
    for img in (synth.disk(max(h, w), 200, 50, noise=40, seed=3, h=h, w=w), np.random.default_rng(1).integers(0, 256, size=shape, dtype=np.uint8)):
        for math in (1, 2):
            res, _ = run_pm(capi, [img], 30, 0.25, 5.75, math, pm_kernel=4)
            ref, info = run_pm(capi, [img], 30, 0.25, 5.75, math, pm_kernel=3)
            assert info["kernel"].startswith("pm_wave_k2_kernel")
            assert np.array_equal(res[0], ref[0])


def test_pm_resident_is_the_default_where_it_fits(capi, oracle):
    img = synth.disk(512, 200, 50, noise=30, seed=9)
    gpu, info = run_pm(capi, [img], 30, 0.25, 10, 2)
    assert info["kernel"] == "pm_resident_kernel<true, 2>" and info["launches"] == "1" and info["trips"] == "40"   # 32 x 4 tiles of 16 x 128: the shortest bands whose tiles fit the CUs
    cpu = oracle.perona_malik([img], 30, 0.25, 10)
    d = np.abs(gpu[0].astype(int) - cpu[0].astype(int))
    assert d.max() <= 1 and (d != 0).sum() <= 1
    # a short run keeps the per-launch flow (a cooperative launch costs ~25 us more than it gains in 4 steps)
    _, info = run_pm(capi, [img], 30, 0.25, 1, 2)
    assert info["kernel"].startswith("pm_wave_k2_kernel") and info["trips"] == "4"
    # shapes that do not qualify (odd width, fewer than 16 rows) keep the per-launch flow; asking for the resident kernel there is an error
    with capi.Context(40, 57, 1) as ctx:
        ctx.set_image([np.zeros((40, 57), np.uint8)])
        ctx.perona_malik(30, 0.25, 1)
        assert not ctx.launch_info(1)["kernel"].startswith("pm_resident")
        ctx.set_option("pm_kernel", 4)
        with pytest.raises(capi.CvhError):
            ctx.perona_malik(30, 0.25, 1)


def test_pm_resident_2048_full_plane(capi, oracle):
    """256 tiles of 128 x 128 -- every CU -- 40 steps, STRICT: bit-exact against the oracle; FAST: the per-launch flow's bytes."""
    img = synth.disk(2048, 200, 50, noise=40, seed=2)
    cpu = oracle.perona_malik([img], 30, 0.25, 10)
    gpu, info = run_pm(capi, [img], 30, 0.25, 10, 1, pm_kernel=4)
    assert info["kernel"] == "pm_resident_kernel<false, 16>" and info["grid"] == "256" and info["tiles_y"] == "16" and info["tiles_x"] == "16"
    assert np.array_equal(gpu[0], cpu[0])
    res, _ = run_pm(capi, [img], 30, 0.25, 10, 2, pm_kernel=4)
    ref, _ = run_pm(capi, [img], 30, 0.25, 10, 2, pm_kernel=3)
    assert np.array_equal(res[0], ref[0])
