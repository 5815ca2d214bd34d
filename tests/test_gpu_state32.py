"""The DECLARED FP32-state mode (cvh_set_option "state" = 32; SURVEY.md section 7 step 8 / section 8(d), BASELINE.md optional row): the
level set lives in HBM as float -- 9 (three channels: 11) instead of 17 (19) bytes per pixel-iteration -- while the arithmetic, the
tables and the 64-bit fixed-point sums are the FP64 mode's.  It deliberately departs from the reference's CV_64FC1 state
(/root/reference/src/main.cpp:225), so its bar is the survey's statistical one against the FP64 oracle (mask IoU >= 0.999, median
|du| / max|u| <= 1e-4, p99 reported: tests/test_gpu_fullsize.py runs it on the BASELINE configurations) -- and, sharper, what this file
checks on small shapes: ONE iteration from a float-representable level set is the FP64 oracle's iteration rounded to float, and a few
iterations follow an oracle whose level set is rounded to float after every step."""
import numpy as np
import pytest

from chan_vese_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from chan_vese_amd import capi as m
    m.lib()
    assert m.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return m


def f32(a):
    return a.astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("channels", [1, 3])
@pytest.mark.parametrize("shape", [(40, 144), (17, 1008), (150, 528), (9, 272), (64, 2016)])
def test_state32_follows_the_float_rounded_oracle(capi, oracle, shape, channels):
    h, w = shape
    rng = np.random.default_rng(1000 * h + w + channels)
    planes = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(channels)]
    pk = dict(tol=0, nu=0.01)
    if channels == 3:
        pk.update(lambda1=[1, 0.8, 0.5], lambda2=[0.7, 0.5, 1])
    p = oracle.make_params(**pk)
    u0 = f32(rng.normal(scale=3.0, size=shape))          # float-representable: what the float buffers take over is u0 itself
    ulp = lambda x: np.spacing(np.abs(x).astype(np.float32)).astype(np.float64)
    with capi.Context(h, w, channels, capi.make_params(**pk)) as ctx:
        ctx.set_option("state", 32)
        ctx.set_option("trace", 8)
        ctx.set_image(planes)
        ctx.set_levelset(u0)
        assert ctx.launch_info()["kernel"].startswith(f"csv_wave2_kernel<{channels}, true, 3, ") and ctx.launch_info()["kernel"].endswith("true>")
        assert np.array_equal(ctx.get_levelset(), u0)
        # one iteration = the FP64 iteration, rounded once
        u_c = u0.copy()
        nrm_c, c1_c, c2_c = oracle.csv_step(planes, u_c, p)
        done, nrm = ctx.run(1)
        u_g = ctx.get_levelset()
        assert done == 1 and np.array_equal(u_g, f32(u_g))                  # the mirror holds floats
        d = np.abs(u_g - f32(u_c))
        # the GPU's and the oracle's doubles differ by ~1e-9 absolute here (region means that agree to 1e-13 times a region term of ~1e4: the FP64
        # mode's 1e-9 max|u| tolerance): a value that close to a float tie rounds the other way -- one float step (two of the lower binade's at a
        # power of two) on ~1e-3 of the pixels --, and where |u| is small a float step is finer than that difference
        tol = 1e-9 * np.abs(u_c).max()
        assert (d > tol).mean() <= 1e-2 and np.all(d <= 2 * ulp(u_c) + tol), ((d > tol).mean(), ((d - tol) / ulp(u_c)).max())
        tr = ctx.get_trace(1)[0]
        assert np.allclose(tr, list(c1_c) + list(c2_c) + [nrm_c], rtol=1e-9, atol=0)
        # seven more: an oracle whose level set is rounded to float after every iteration
        u_f = f32(u_c)
        for _ in range(7):
            oracle.csv_step(planes, u_f, p)
            u_f = f32(u_f)
        done, _ = ctx.run(7)
        u_g = ctx.get_levelset()
        m_g = ctx.get_mask()
    assert done == 7
    scale = np.abs(u_f).max()
    assert np.abs(u_g - u_f).max() <= 2e-5 * scale, np.abs(u_g - u_f).max() / scale     # float roundings that fell the other way, amplified by the recurrence
    assert (m_g != oracle.mask(u_f)).mean() <= 1e-4


def test_state32_option_rules(capi, oracle):
    """Widths the 2-pixel kernel cannot take are refused when the option is set; STRICT arithmetic is refused when the run starts; switching
    the state keeps the level set (rounded once on the way to 32, exact on the way back); the cache policy follows the smaller footprint."""
    with capi.Context(64, 100, 1) as ctx:
        with pytest.raises(capi.CvhError):
            ctx.set_option("state", 32)
        with pytest.raises(capi.CvhError):
            ctx.set_option("state", 16)
    h, w = 48, 160
    img = synth.disk(48, 200, 50, noise=8, seed=2, h=h, w=w)
    rng = np.random.default_rng(5)
    u0 = rng.normal(size=(h, w))
    with capi.Context(h, w, 1, capi.make_params(tol=0)) as ctx:
        ctx.set_image([img])
        ctx.set_levelset(u0)
        ctx.set_option("state", 32)
        assert np.array_equal(ctx.get_levelset(), f32(u0))
        assert ctx.run(3)[0] == 3
        u3 = ctx.get_levelset()
        ctx.set_option("state", 64)                      # back: the same values, now iterated in double
        assert np.array_equal(ctx.get_levelset(), u3)
        assert ctx.launch_info()["kernel"].endswith("false>") or not ctx.launch_info()["kernel"].startswith("csv_wave2")
        assert ctx.run(2)[0] == 2
        u5 = ctx.get_levelset()
        u_c = u3.copy()
        for _ in range(2):
            oracle.csv_step([img], u_c, oracle.make_params(tol=0))
        assert np.abs(u5 - u_c).max() <= 1e-9 * np.abs(u_c).max()
        ctx.set_option("state", 32)
        ctx.set_option("math_mode", 1)
        with pytest.raises(capi.CvhError):
            ctx.run(1)
