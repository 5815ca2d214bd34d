"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(chan_vese_amd.capi -> libchanvese_hip.so), against the CPU oracle on the same seeded inputs.

Stated tolerances (FP64 state): per-pixel max|u_gpu - u_cpu| / max|u_cpu|
  <= 1e-9 over the first ten iterations, <= 1e-6 after the configured iteration count;
  c1/c2 relative <= 1e-9 per iteration; mask IoU >= 0.999; identical stop iteration.
The only sources of difference are the device atan, the summation order of the region sums
and (FAST mode) <= 2 ulp rsqrt/rcp refinements."""
import numpy as np
import pytest

from chan_vese_amd import synth

pytestmark = pytest.mark.gpu

MODES = [("strict", 1), ("fast", 2)]


@pytest.fixture(scope="module")
def capi():
    from chan_vese_amd import capi as m
    m.lib()
    assert m.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return m


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def iou(a, b):
    a, b = a.astype(bool), b.astype(bool)
    u = (a | b).sum()
    return 1.0 if u == 0 else (a & b).sum() / u


def gpu_run(capi, planes, u0, steps, math=1, finalize=0, trace=True, opts=None, **pk):
    h, w = planes[0].shape
    with capi.Context(h, w, len(planes), capi.make_params(**pk)) as ctx:
        ctx.set_option("math_mode", math)
        ctx.set_option("finalize", finalize)
        for k, v in (opts or {}).items():
            ctx.set_option(k, v)
        if trace:
            ctx.set_option("trace", max(steps, 1))
        ctx.set_image(planes)
        ctx.set_levelset(u0)
        done, nrm = ctx.run(steps)
        return ctx.get_levelset(), done, nrm, ctx.get_trace(steps) if trace else None, ctx.get_mask()


@pytest.mark.parametrize("mode,math", MODES)
@pytest.mark.parametrize("finalize,kernel", [(0, -1), (1, -1), (0, 2)])
@pytest.mark.parametrize("shape", [(32, 48), (64, 64), (37, 53), (1, 40), (40, 1), (2, 2), (3, 700),
                                   (100, 517)])
def test_csv_small_shapes(capi, oracle, shape, mode, math, finalize, kernel):
    h, w = shape
    rng = np.random.default_rng(h * 7919 + w)
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    u0 = oracle.checkerboard(h, w) if min(h, w) > 2 else rng.normal(size=shape)
    for steps in (1, 2, 3, 10):
        u_c, done_c, nrm_c, tr_c = oracle.csv_run([img], u0, oracle.make_params(tol=0), steps)
        u_g, done_g, nrm_g, tr_g, _ = gpu_run(capi, [img], u0, steps, math, finalize, opts=dict(kernel=kernel), tol=0)
        assert done_g == done_c == steps
        assert rel_err(u_g, u_c) <= 1e-9, (steps, rel_err(u_g, u_c))
        assert np.allclose(tr_g, tr_c, rtol=1e-9, atol=0), (tr_g, tr_c)
        assert nrm_g == pytest.approx(nrm_c, rel=1e-9)


@pytest.mark.parametrize("mode,math", MODES)
@pytest.mark.parametrize("shape", [(1, 144), (2, 160), (3, 256), (5, 2016), (9, 272), (40, 144), (17, 1008), (8, 4096)])
def test_csv_two_pixel_kernel_edge_shapes(capi, oracle, shape, mode, math):
    """csv_wave2_kernel (2 pixels per lane, w % 16 == 0, w >= 144): one-row images, a single wave-column, widths that are
    exact multiples of its 126-column stride (full last wave), a partial last wave, strips shorter than a group."""
    h, w = shape
    rng = np.random.default_rng(h * 104729 + w)
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    u0 = oracle.checkerboard(h, w) if h > 2 else rng.normal(size=shape)
    for steps, opts in ((1, dict(kernel=3)), (4, dict(kernel=3, strip_rows=8)), (9, dict(kernel=3, wave_occupancy=3, wave_xcd=0))):
        u_c, done_c, nrm_c, tr_c = oracle.csv_run([img], u0, oracle.make_params(tol=0), steps)
        u_g, done_g, nrm_g, tr_g, m_g = gpu_run(capi, [img], u0, steps, math, opts=opts, tol=0)
        assert done_g == done_c == steps
        assert rel_err(u_g, u_c) <= 1e-9, (steps, rel_err(u_g, u_c))
        assert np.allclose(tr_g, tr_c, rtol=1e-9, atol=0)
        assert np.array_equal(m_g, oracle.mask(u_c))


@pytest.mark.parametrize("shape", [(1, 144), (2, 160), (3, 256), (5, 2016), (9, 272), (40, 144), (17, 1008), (8, 4096), (150, 528)])
def test_csv_three_channel_two_pixel_kernel(capi, oracle, shape):
    """Round 3: the 2-pixel wave kernel with three channels (csv_wave2_kernel<3, ...>, FAST only, "kernel" = 3): one image
    tile per channel, samples read inside the row, region term from three tables (the table-free quadratic form of round 3 was
    pruned in round 4: tools/experiments/pruned_flavours/).  Same shapes as the 1-channel edge-shape test, per-channel lambdas,
    nu != 0, ragged strips."""
    h, w = shape
    rng = np.random.default_rng(7 * h + w)
    planes = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(3)]
    u0 = oracle.checkerboard(h, w) if min(h, w) > 2 else rng.normal(size=shape)
    pk = dict(tol=0, lambda1=[1, 0.8, 0.5], lambda2=[0.7, 0.5, 1], nu=0.01)
    for steps, opts in ((1, {}), (7, dict(strip_rows=8)), (4, dict(chain=0)), (19, {})):
        u_c, _, nrm_c, tr_c = oracle.csv_run(planes, u0, oracle.make_params(**pk), steps)
        with capi.Context(h, w, 3, capi.make_params(**pk)) as ctx:
            ctx.set_option("kernel", 3)
            for k, v in opts.items():
                ctx.set_option(k, v)
            assert ctx.launch_info()["kernel"].startswith("csv_wave2_kernel<3, true, 3, "), ctx.launch_info()
            ctx.set_option("trace", steps)
            ctx.set_image(planes)
            ctx.set_levelset(u0)
            done, nrm = ctx.run(steps)
            u_g, tr_g, m_g = ctx.get_levelset(), ctx.get_trace(steps), ctx.get_mask()
        assert done == steps
        assert rel_err(u_g, u_c) <= 1e-9, (steps, opts, rel_err(u_g, u_c))
        assert np.allclose(tr_g, tr_c, rtol=1e-9, atol=0)
        assert np.array_equal(m_g, oracle.mask(u_c))


def test_launch_info_names_what_runs(capi):
    """cvh_launch_info is written by the launch sites themselves: the kernel instantiation as rocprofv3 prints it."""
    with capi.Context(1024, 1024, 1) as ctx:      # >= 0.6 Mpixel: the 2-pixel kernel is the default
        ctx.set_image([synth.disk(1024)])
        ctx.init_checkerboard()
        assert ctx.launch_info()["kernel"] == "csv_resident_kernel<4>"       # the default for a plane that fits the chip's LDS: 32 x 8 tiles of 32 rows, 4 per wave
        ctx.set_option("resident", 0)
        i = ctx.launch_info()
        assert i["kernel"] == "csv_wave2_kernel<1, true, 3, 1, false>" and int(i["grid"]) > 1 and i["chain"] == "1" and i["math"] == "fast"
        ctx.set_option("kernel", 2)
        assert ctx.launch_info()["kernel"].startswith("csv_wave_kernel<1, true, true, 5, true, 1, 1>")
        ctx.set_option("math_mode", 1)
        assert ctx.launch_info()["kernel"].startswith("csv_wave_kernel<1, false, false, 3,") and ctx.launch_info()["math"] == "strict"
        ctx.set_option("kernel", 0)
        assert ctx.launch_info()["kernel"].startswith("csv_step_kernel<1, 14, false, false,")
        with pytest.raises(capi.CvhError):
            ctx.launch_info(1)            # no Perona-Malik run yet
        ctx.perona_malik(30.0, 0.25, 5.0)
        p = ctx.launch_info(1)      # the default for a plane that fits the chip's LDS and a run of >= 16 steps: one cooperative launch
        assert p["kernel"].startswith("pm_resident_kernel<false, ") and p["launches"] == "1" and p["trips"] == "20"
        ctx.perona_malik(30.0, 0.25, 2.25)      # a short run keeps the per-launch flow
        p = ctx.launch_info(1)
        assert p["kernel"].startswith("pm_wave_k2_kernel<false,") and p["steps_per_launch"] == "2" and p["trips"] == "9"
        assert p["last_step_kernel"].startswith("pm_wave_kernel<false>")


@pytest.mark.parametrize("mode,math", MODES)
@pytest.mark.parametrize("opts", [dict(kernel=0, tile_rows=14), dict(kernel=0, tile_rows=16, lut=0),
                                  dict(kernel=0, tile_rows=14, dma=1),
                                  dict(kernel=2), dict(kernel=2, strip_rows=8, lut=0),
                                  dict(kernel=2, strip_rows=52, wave_occupancy=4),
                                  dict(kernel=2, strip_rows=1000, wave_skew=150, wave_prio=2),
                                  dict(kernel=3), dict(kernel=3, strip_rows=8), dict(kernel=3, strip_rows=37),
                                  dict(kernel=3, strip_rows=1000, wave_xcd=0)])
def test_csv_kernel_variants(capi, oracle, mode, math, opts):
    """Every data-flow variant of the step kernel (LDS tile / streaming strip, ring chunking,
    ragged last chunk, LDS-DMA loader, LUT on/off) against the oracle on a multi-tile image."""
    h, w = 150, 528
    img = synth.disk(150, 190, 60, noise=12, seed=9, h=h, w=w)
    u0 = oracle.checkerboard(h, w)
    u_c, _, nrm_c, tr_c = oracle.csv_run([img], u0, oracle.make_params(tol=0), 6)
    u_g, _, nrm_g, tr_g, _ = gpu_run(capi, [img], u0, 6, math, opts=opts, tol=0)
    assert rel_err(u_g, u_c) <= 1e-9
    assert np.allclose(tr_g, tr_c, rtol=1e-9, atol=0)


@pytest.mark.parametrize("mode,math", MODES)
def test_csv_nondefault_params(capi, oracle, mode, math):
    h, w = 48, 300
    img = synth.disk(48, 180, 70, noise=20, seed=5, h=h, w=w)
    u0 = oracle.checkerboard(h, w)
    pk = dict(mu=0.2, nu=0.05, dt=0.1, eps=0.5, tol=0)
    u_c, _, nrm_c, tr_c = oracle.csv_run([img], u0, oracle.make_params(**pk), 10)
    u_g, _, nrm_g, tr_g, _ = gpu_run(capi, [img], u0, 10, math, **pk)
    assert rel_err(u_g, u_c) <= 1e-9
    assert np.allclose(tr_g, tr_c, rtol=1e-9, atol=0)


@pytest.mark.parametrize("mode,math", MODES)
def test_csv_three_channel(capi, oracle, mode, math):
    h, w = 80, 272
    planes = [synth.disk(80, 180, 40, h=h, w=w), synth.disk(80, 200, 60, h=h, w=w),
              synth.disk(80, 60, 200, h=h, w=w)]
    pk = dict(tol=0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1])
    u0 = oracle.checkerboard(h, w)
    u_c, _, _, tr_c = oracle.csv_run(planes, u0, oracle.make_params(**pk), 10)
    u_g, _, _, tr_g, m_g = gpu_run(capi, planes, u0, 10, math, **pk)
    assert rel_err(u_g, u_c) <= 1e-9
    assert np.allclose(tr_g, tr_c, rtol=1e-9, atol=0)
    assert np.array_equal(m_g, oracle.mask(u_c))


@pytest.mark.parametrize("mode,math", MODES)
def test_config1_512_disk_100_iters(capi, oracle, mode, math):
    """BASELINE.json configs[0]: 512x512 disk, checkerboard init, 100 CSV iterations."""
    img = synth.disk(512)
    u0 = oracle.checkerboard(512, 512)
    u_c, done_c, nrm_c, tr_c = oracle.csv_run([img], u0, oracle.make_params(tol=0), 100)
    u_g, done_g, nrm_g, tr_g, m_g = gpu_run(capi, [img], u0, 100, math, tol=0)
    assert done_g == done_c == 100
    assert np.allclose(tr_g[:10], tr_c[:10], rtol=1e-9, atol=0)
    assert rel_err(u_g, u_c) <= 1e-6, rel_err(u_g, u_c)
    assert iou(m_g, oracle.mask(u_c)) >= 0.999
    assert iou(m_g, img == 200) == 1.0


def test_checkerboard_host_matches_oracle(capi, oracle):
    assert np.array_equal(capi.checkerboard_host(301, 517), oracle.checkerboard(301, 517))
    with capi.Context(64, 80, 1) as ctx:
        ctx.init_checkerboard()
        assert np.array_equal(ctx.get_levelset(), oracle.checkerboard(64, 80))


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (5, 1), (3, 5), (11, 10), (301, 517), (2500, 1200)])
def test_checkerboard_on_device_is_bit_identical(capi, oracle, shape):
    """cvh_init_checkerboard uploads h + w sine factors and takes the sign of their product on the device: the level set
    must be the host function's (and the oracle's) bit for bit, whatever ran on the context before."""
    h, w = shape
    with capi.Context(h, w, 1, capi.make_params(tol=0)) as ctx:
        ctx.init_checkerboard()
        assert np.array_equal(ctx.get_levelset(), oracle.checkerboard(h, w))
        ctx.set_image([synth.disk(max(h, w), 200, 50)[:h, :w]])
        ctx.enqueue_steps(3)          # leaves u in the other buffer, work in flight
        ctx.init_checkerboard()
        assert np.array_equal(ctx.get_levelset(), capi.checkerboard_host(h, w))
        assert ctx.sync()[0] == 0


@pytest.mark.parametrize("shape,channels", [((1, 1), 1), ((3, 5), 1), ((7, 9), 3), ((33, 47), 1), ((64, 80), 3), ((1000, 1037), 1), ((500, 777), 3)])
def test_stop_condition_is_the_oracles_exactly(capi, oracle, shape, channels):
    """sum(I^2) of one channel is an exact integer sum on the device; three channels keep the reference's serial order on
    the host: both must equal the oracle's double, not approximate it.  The region means' sum(I_k) ride on the same kernel."""
    h, w = shape
    rng = np.random.default_rng(h * 131 + w)
    planes = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(channels)]
    with capi.Context(h, w, channels, capi.make_params(tol=0.37)) as ctx:
        ctx.set_image(planes)
        ctx.init_checkerboard()
        assert ctx.get_stop_condition() == oracle.stop_condition(planes, 0.37)
        planes2 = [255 - p for p in planes]
        ctx.set_image(planes2)
        assert ctx.get_stop_condition() == oracle.stop_condition(planes2, 0.37)
        # c1/c2 use sum(I_k) from the same pass
        u0 = oracle.checkerboard(h, w)
        c1, c2 = ctx.get_means()
        for k in range(channels):
            assert c1[k] == pytest.approx(oracle.region_mean(planes2[k], u0, 0), rel=1e-12)
            assert c2[k] == pytest.approx(oracle.region_mean(planes2[k], u0, 1), rel=1e-12)


def test_stop_rule_same_iteration(capi, oracle):
    """Default tolerance: the loop must break at the reference's iteration (after the update)."""
    img = synth.disk(128, 200, 50)
    u0 = oracle.checkerboard(128, 128)
    for tol in (1e-3, 0.05, 0.5):
        u_c, done_c, nrm_c, _ = oracle.csv_run([img], u0, oracle.make_params(tol=tol), 400)
        for sync_every in (1, 7, 32):
            with capi.Context(128, 128, 1, capi.make_params(tol=tol)) as ctx:
                ctx.set_option("sync_every", sync_every)
                ctx.set_image([img])
                ctx.set_levelset(u0)
                assert ctx.get_stop_condition() == pytest.approx(oracle.stop_condition([img], tol), rel=1e-14)
                done_g, nrm_g = ctx.run(400)
                u_g = ctx.get_levelset()
            assert done_g == done_c, (tol, sync_every, done_g, done_c)
            assert nrm_g == pytest.approx(nrm_c, rel=1e-7)
            assert rel_err(u_g, u_c) <= 1e-6


@pytest.mark.parametrize("shape,channels,kernel", [((64, 160), 1, 3), ((64, 160), 1, 2), ((48, 100), 3, 2), ((700, 1008), 1, 3)])
def test_chain_mode_stop_edges(capi, oracle, shape, channels, kernel):
    """Chain mode books an iteration one launch late (chain_device.h): the stop iteration, the level set, the trace and the
    region means must still be the reference's when the stop fires at the FIRST iteration, at the last enqueued one, in the
    middle of a hipGraph, and a run / an enqueue chain must continue correctly afterwards."""
    h, w = shape
    rng = np.random.default_rng(h + w + channels)
    planes = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(channels)]
    u0 = oracle.checkerboard(h, w)
    lam = dict(lambda1=[1, 0.8, 0.5], lambda2=[0.7, 0.5, 1]) if channels == 3 else {}
    # norms of the free run decide the tolerances that stop at chosen iterations
    _, _, _, tr = oracle.csv_run(planes, u0, oracle.make_params(tol=0, **lam), 24)
    base = oracle.stop_condition(planes, 1.0)
    for stop_at in (1, 2, 17, 24):
        # the first iteration whose norm is <= tol * base must be `stop_at`: pick tol just above that norm
        norms = tr[:, -1]
        tol = float(norms[stop_at - 1] / base * (1 + 1e-9))
        if np.any(norms[:stop_at - 1] <= tol * base):
            continue          # an earlier iteration already has a smaller norm: this stop point cannot be isolated
        pk = dict(tol=tol, **lam)
        u_c, done_c, nrm_c, tr_c = oracle.csv_run(planes, u0, oracle.make_params(**pk), 40)
        assert done_c == stop_at
        for mode in ("run", "enqueue"):
            with capi.Context(h, w, channels, capi.make_params(**pk)) as ctx:
                ctx.set_option("kernel", kernel)
                ctx.set_option("trace", 64)
                ctx.set_image(planes)
                ctx.set_levelset(u0)
                if mode == "run":
                    done, nrm = ctx.run(40)
                else:
                    ctx.enqueue_steps(19); ctx.enqueue_steps(21)
                    done, nrm, stopped = ctx.sync()
                    assert stopped
                assert done == stop_at, (mode, stop_at, done)
                assert nrm == pytest.approx(nrm_c, rel=1e-9)
                u_g = ctx.get_levelset()
                assert rel_err(u_g, u_c) <= 1e-9
                assert np.allclose(ctx.get_trace(64), tr_c, rtol=1e-9, atol=0)
                c1g, c2g = ctx.get_means()
                c1c = [oracle.region_mean(p, u_c, 0) for p in planes]
                c2c = [oracle.region_mean(p, u_c, 1) for p in planes]
                assert np.allclose(c1g, c1c, rtol=1e-9) and np.allclose(c2g, c2c, rtol=1e-9)
                # continuation: a new run from the stopped state with tol 0 matches the oracle's continuation
                ctx.set_params(capi.make_params(tol=0, **lam))
                done2, _ = ctx.run(3)
                assert done2 == 3
                u_c2, _, _, _ = oracle.csv_run(planes, u_c, oracle.make_params(tol=0, **lam), 3)
                assert rel_err(ctx.get_levelset(), u_c2) <= 1e-9
    # zero and one step, no stop
    with capi.Context(h, w, channels, capi.make_params(tol=0, **lam)) as ctx:
        ctx.set_option("kernel", kernel)
        ctx.set_image(planes)
        ctx.set_levelset(u0)
        assert ctx.run(0)[0] == 0 and np.array_equal(ctx.get_levelset(), u0)
        assert ctx.run(1)[0] == 1
        u1 = u0.copy()
        oracle.csv_step(planes, u1, oracle.make_params(tol=0, **lam))
        assert rel_err(ctx.get_levelset(), u1) <= 1e-12


@pytest.mark.parametrize("shape,channels,kernel", [((64, 160), 1, 3), ((48, 100), 3, 2)])
def test_graph_chunks_that_are_not_multiples_of_four(capi, oracle, shape, channels, kernel):
    """The captured step arguments have period 4 in chain mode (the sum set of the first step): chunk sizes >= 16 that are
    not multiples of 4 -- sync_every 18 and 50, cvh_enqueue_steps(18) repeated without a sync, a run that follows a run of
    odd length -- cycle through the four cached graphs (api.hip, ensure_step_graph).  Results equal the oracle's and the
    plain-launch path's bit for bit."""
    h, w = shape
    rng = np.random.default_rng(11 * h + w + channels)
    planes = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(channels)]
    u0 = oracle.checkerboard(h, w)
    lam = dict(lambda1=[1, 0.8, 0.5], lambda2=[0.7, 0.5, 1]) if channels == 3 else {}
    pk = dict(tol=0, **lam)
    steps = 118
    u_c, _, nrm_c, tr_c = oracle.csv_run(planes, u0, oracle.make_params(**pk), steps)

    def gpu(drive, graph=1):
        with capi.Context(h, w, channels, capi.make_params(**pk)) as ctx:
            ctx.set_option("kernel", kernel)
            ctx.set_option("graph", graph)
            ctx.set_option("trace", steps)
            ctx.set_image(planes)
            ctx.set_levelset(u0)
            done = drive(ctx)
            assert done == steps
            return ctx.get_levelset(), ctx.get_trace(steps)

    def by_sync_every(n):
        def drive(ctx):
            ctx.set_option("sync_every", n)
            return ctx.run(steps)[0]
        return drive

    def by_enqueue_18(ctx):
        for _ in range(6):
            ctx.enqueue_steps(18)          # 2 plain launches + one graph of 16, starting at a new phase every time
        ctx.enqueue_steps(10)
        return ctx.sync()[0]

    def by_odd_runs(ctx):                  # every run ends on a count that is not a multiple of 4: the next starts at another phase
        total = 0
        for n in (17, 19, 33, 18, 31):
            total += ctx.run(n)[0]
        return total

    ref_u, ref_tr = gpu(by_sync_every(32), graph=0)
    assert rel_err(ref_u, u_c) <= 1e-6 and np.allclose(ref_tr[:10], tr_c[:10], rtol=1e-9, atol=0)
    for drive in (by_sync_every(18), by_sync_every(50), by_enqueue_18):
        u_g, tr_g = gpu(drive)
        assert np.array_equal(u_g, ref_u) and np.array_equal(tr_g, ref_tr)
    u_g, _ = gpu(by_odd_runs)              # the trace restarts with every run: compare the level set
    assert np.array_equal(u_g, ref_u)


def test_warm_does_not_touch_the_chain_state(capi, oracle):
    """cvh_warm only captures: it must not mark chain-mode launches as pending.  chain=0 run (means valid in the state block),
    chain=1 + warm(16), back to chain=0: the next launches must still use the right region means (a flush
    kernel used to overwrite them from sum sets that were never seeded)."""
    h, w = 64, 160
    img = synth.disk(64, 200, 50, noise=10, seed=4, h=h, w=w)
    u0 = oracle.checkerboard(h, w)
    u_c, _, _, _ = oracle.csv_run([img], u0, oracle.make_params(tol=0), 9)
    with capi.Context(h, w, 1, capi.make_params(tol=0)) as ctx:
        ctx.set_option("kernel", 3)
        ctx.set_option("chain", 0)
        ctx.set_image([img])
        ctx.set_levelset(u0)
        assert ctx.run(4)[0] == 4
        ctx.set_option("chain", 1)
        ctx.warm(16)
        ctx.set_option("chain", 0)
        assert ctx.run(5)[0] == 5
        assert rel_err(ctx.get_levelset(), u_c) <= 1e-9


def test_unlimited_steps_runs_to_stop(capi, oracle):
    img = synth.disk(96, 220, 30)
    u0 = oracle.checkerboard(96, 96)
    u_c, done_c, _, _ = oracle.csv_run([img], u0, oracle.make_params(tol=0.2), 100000)
    with capi.Context(96, 96, 1, capi.make_params(tol=0.2)) as ctx:
        ctx.set_image([img])
        ctx.set_levelset(u0)
        done_g, _ = ctx.run(-1)      # src/main.cpp:890: negative => unlimited
    assert done_g == done_c and done_c < 100000


def test_run_continues_and_is_deterministic(capi, oracle):
    img = synth.disk(200, 200, 50, noise=10, seed=2, h=120, w=530)
    u0 = oracle.checkerboard(120, 530)
    outs = []
    for _ in range(2):
        with capi.Context(120, 530, 1, capi.make_params(tol=0)) as ctx:
            ctx.set_image([img])
            ctx.set_levelset(u0)
            ctx.run(7)
            ctx.run(6)              # continue from the current level set
            outs.append(ctx.get_levelset())
    assert np.array_equal(outs[0], outs[1])       # fixed summation order => bitwise reproducible
    u_c, _, _, _ = oracle.csv_run([img], u0, oracle.make_params(tol=0), 13)
    assert rel_err(outs[0], u_c) <= 1e-9


def test_enqueue_sync_interleaved_contexts(capi, oracle):
    """Two independent images on one GPU driven through the asynchronous halves of cvh_run."""
    imgs = [synth.disk(96, 200, 50, noise=5, seed=s) for s in (1, 2)]
    u0 = oracle.checkerboard(96, 96)
    ctxs = [capi.Context(96, 96, 1, capi.make_params(tol=0)) for _ in imgs]
    for c, im in zip(ctxs, imgs):
        c.set_image([im])
        c.set_levelset(u0)
    for _ in range(3):
        for c in ctxs:
            c.enqueue_steps(4)
    for c, im in zip(ctxs, imgs):
        done, nrm, stopped = c.sync()
        assert done == 12 and not stopped
        u_c, _, _, _ = oracle.csv_run([im], u0, oracle.make_params(tol=0), 12)
        assert rel_err(c.get_levelset(), u_c) <= 1e-9
        c.close()


@pytest.mark.parametrize("pm_opts", [dict(pm_kernel=1), dict(pm_kernel=1, pm_strip_rows=8), dict(pm_kernel=0), dict(),
                                     dict(pm_kernel=3), dict(pm_kernel=3, pm_strip_rows=8), dict(pm_kernel=3, pm_strip_rows=24),
                                     dict(pm_strip_rows=16)])
@pytest.mark.parametrize("shape,K,L,T", [((40, 56), 30, 0.25, 5), ((64, 64), 10, 0.25, 20),
                                         ((37, 130), 1000, 0.1, 1.5), ((1, 50), 30, 0.25, 2),
                                         ((50, 1), 30, 0.25, 2), ((3, 3), 30, 0.2, 1),
                                         ((1, 128), 10, 0.25, 2), ((9, 248), 20, 0.25, 3), ((70, 372), 30, 0.25, 4)])
def test_perona_malik_parity(capi, oracle, shape, K, L, T, pm_opts):
    """Every Perona-Malik data flow (tile, wave; wave with TWO time steps per launch: even and odd trip counts, strips
    shorter than / equal to / longer than the image; no option: the resident-plane kernel where the shape qualifies --
    tests/test_gpu_pm_resident.py -- and the 2-step kernel elsewhere) against the oracle."""
    h, w = shape
    rng = np.random.default_rng(11 + h + w)
    planes = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(3)]
    cpu = oracle.perona_malik(planes, K, L, T)
    assert capi.pm_trip_count(L, T) == oracle.pm_trip_count(L, T)
    for math in (1, 2):
        with capi.Context(h, w, 3) as ctx:
            ctx.set_option("math_mode", math)
            for k, v in pm_opts.items():
                ctx.set_option(k, v)
            ctx.set_image(planes)
            ctx.perona_malik(K, L, T)
            gpu = ctx.get_image()
        for g, c in zip(gpu, cpu):
            if math == 1:
                assert (g != c).sum() == 0   # STRICT: bit-exact uint8 (contraction off on both sides)
            else:
                # FAST (rcp + refinement, FMAs): differs only where a value sits on a rounding
                # boundary: <= 1 LSB on <= 1e-6 of the pixels
                d = np.abs(g.astype(int) - c.astype(int))
                assert d.max() <= 1 and (d != 0).sum() <= max(1, int(1e-6 * d.size))


def test_pm_then_csv_pipeline(capi, oracle):
    """Config 4 in miniature: PM-smoothed 8-bit image feeds CSV and the stop condition."""
    img = synth.disk(96, 200, 50, noise=32, seed=1)
    u0 = oracle.checkerboard(96, 96)
    sm = oracle.perona_malik([img], 30, 0.25, 10)
    u_c, done_c, _, _ = oracle.csv_run(sm, u0, oracle.make_params(tol=1e-3), 60)
    with capi.Context(96, 96, 1, capi.make_params(tol=1e-3)) as ctx:
        ctx.set_image([img])
        ctx.perona_malik(30, 0.25, 10)
        ctx.set_levelset(u0)
        assert ctx.get_stop_condition() == oracle.stop_condition(sm, 1e-3)   # planes changed on the device: exact all the same
        done_g, _ = ctx.run(60)
        u_g = ctx.get_levelset()
    assert done_g == done_c
    assert rel_err(u_g, u_c) <= 1e-6


def test_mask_and_separate(capi, oracle):
    rng = np.random.default_rng(4)
    u = rng.normal(size=(33, 47))
    u[0, :5] = [1e-50, -1e-50, 0.0, 1e-46, 5e-324]
    img3 = rng.integers(0, 256, size=(33, 47, 3), dtype=np.uint8)
    with capi.Context(33, 47, 1) as ctx:
        ctx.set_levelset(u)
        for inv in (False, True):
            assert np.array_equal(ctx.get_mask(inv), oracle.mask(u, inv))
            assert np.array_equal(ctx.separate(img3, inv), oracle.separate(img3, u, inv))


@pytest.mark.parametrize("shape", [(1, 1), (2, 7), (3, 3), (33, 47), (150, 528)])
def test_video_contour(capi, oracle, shape):
    """cvh_get_contour = the pixels VideoWriterManager::draw_contour paints (src/VideoWriterManager.cpp:60-74):
    mask rule uint8(round(u)) > 0, half-way cases (0.5, 1.5), negatives, an evolved level set."""
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    u = rng.normal(scale=1.5, size=shape)
    u.ravel()[: min(6, u.size)] = [0.5, 1.5, -0.5, 0.5000000001, 1e300, -1e300][: min(6, u.size)]
    with capi.Context(h, w, 1, capi.make_params(tol=0.0)) as ctx:
        ctx.set_levelset(u)
        assert np.array_equal(ctx.get_contour(), oracle.video_contour(u))
        if h >= 33:
            img = synth.disk(h, 190, 60, noise=8, seed=2, h=h, w=w)
            ctx.set_image([img]); ctx.init_checkerboard(); ctx.run(12)
            ue = ctx.get_levelset()
            c = ctx.get_contour()
            assert np.array_equal(c, oracle.video_contour(ue))
            assert c.sum() > 0 and c[0].sum() == 0 and c[:, 0].sum() == 0


@pytest.mark.parametrize("op", [0, 1, 2])
def test_parallel_pixel_function_ops(capi, oracle, op):
    """ParallelPixelFunction(data, w, f)(Range(a, b)) — src/ParallelPixelFunction.cpp:12-17."""
    rng = np.random.default_rng(op)
    data = rng.normal(scale=50, size=(19, 23))
    for (a, b) in [(0, data.size), (5, 100), (7, 7), (436, 437)]:
        g, c = data.copy(), data.copy()
        capi.ppf_apply(g, op, eps=0.8, start=a, end=b)
        oracle.ppf_apply(c, op, eps=0.8, start=a, end=b)
        assert np.allclose(g, c, rtol=1e-15, atol=3e-16)  # H is formed as (1 + x)/2: absolute error ~1 ulp of 1
        assert np.array_equal(g.ravel()[:a], data.ravel()[:a]) and np.array_equal(g.ravel()[b:], data.ravel()[b:])


def test_error_paths(capi):
    with pytest.raises(capi.CvhError):
        capi.Context(0, 5)
    with pytest.raises(capi.CvhError):
        capi.Context(8, 8, channels=2)
    with pytest.raises(capi.CvhError):
        capi.Context(8, 8, params=capi.make_params(dt=0.0))
    with capi.Context(8, 8) as ctx:
        with pytest.raises(capi.CvhError):
            ctx.run(1)                              # no image / level set yet
        ctx.set_image([np.zeros((8, 8), np.uint8)])
        with pytest.raises(capi.CvhError):
            ctx.perona_malik(10, 0.3, 20)           # L > 0.25, src/main.cpp:863
        with pytest.raises(capi.CvhError):
            ctx.set_option("nonsense", 1)


@pytest.mark.parametrize("shape,channels,kernel", [((700, 1008), 1, 3), ((333, 500), 1, 2), ((200, 272), 3, 2)])
def test_cache_policy_of_the_rows_does_not_change_results(capi, shape, channels, kernel):
    """"wave_pol" selects write-through or plain stores (and sc0 or plain loads) for the streamed level-set rows: a template
    parameter of the wave kernels that must not change a single bit of the level set, the trace or the means."""
    h, w = shape
    rng = np.random.default_rng(h + 7 * w + channels)
    planes = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(channels)]
    out = []
    for pol in (0, 1, -1, 2):
        with capi.Context(h, w, channels, capi.make_params(tol=0.0)) as ctx:
            ctx.set_option("kernel", kernel); ctx.set_option("wave_pol", pol); ctx.set_option("trace", 40)
            ctx.set_image(planes); ctx.init_checkerboard()
            ctx.enqueue_steps(21); ctx.enqueue_steps(16)
            done, nrm, _ = ctx.sync()
            out.append((ctx.get_levelset(), ctx.get_trace(40), done, nrm))
    for o in out[1:]:
        assert o[2] == out[0][2] == 37 and o[3] == out[0][3]
        assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1])


@pytest.mark.parametrize("case", ["two_pixel_c1", "two_pixel_c3", "resident", "one_pixel_c3", "mixed_strips"])
def test_all_near_regime(capi, oracle, case):
    """Round 4: the regime of the reference README's second example (--dt 0.001: |u| stays below the far-field threshold of H_eps, 32 eps, for
    the whole run), sustained over 25 iterations in every FAST flow -- the per-strip / per-band near-form march of the 2-pixel and resident
    kernels ("near_switch", default), the same flows with it switched off (far series + per-group correction on every row), the 1-pixel
    kernel -- against the oracle at 1e-9; "mixed_strips": the upper half of the plane far (|u| = 60), the lower half near, so strips of
    both kinds and the strip the border runs through coexist in one launch."""
    steps = 25
    rng = np.random.default_rng(len(case))
    if case == "resident":
        shape, C, opts = (256, 256), 1, [dict(), dict(near_switch=0), dict(resident=0, kernel=3)]
    elif case == "two_pixel_c1":
        shape, C, opts = (150, 528), 1, [dict(kernel=3, resident=0), dict(kernel=3, resident=0, near_switch=0), dict(kernel=3, resident=0, strip_rows=8)]
    elif case == "two_pixel_c3":
        shape, C, opts = (150, 528), 3, [dict(kernel=3), dict(kernel=3, near_switch=0)]
    elif case == "one_pixel_c3":
        shape, C, opts = (100, 517), 3, [dict()]
    else:
        shape, C, opts = (192, 272), 1, [dict(kernel=3, resident=0), dict(kernel=3, resident=0, near_switch=0), dict()]
    h, w = shape
    planes = [synth.disk(min(h, w), 200 - 30 * k, 50 + 20 * k, noise=12, seed=7 + k, h=h, w=w) for k in range(C)]
    u0 = oracle.checkerboard(h, w)
    pk = dict(tol=0, dt=0.001, nu=-3.0)
    if C == 3:
        pk.update(lambda1=[1, 1, 0.1], lambda2=[1, 0.7, 1])
    if case == "mixed_strips":
        u0 = u0.copy()
        u0[: h // 2] *= 60.0
    u_c, done_c, nrm_c, tr_c = oracle.csv_run(planes, u0, oracle.make_params(**pk), steps)
    if case != "mixed_strips":
        assert np.abs(u_c).max() < 32.0          # the regime this test is about
    else:
        assert np.abs(u_c[1: h // 2 - 4, 1:]).min() > 32.0 and np.abs(u_c[h // 2 + 4:]).max() < 32.0     # (row 0 / column 0 of a checkerboard are zeros)
    for o in opts:
        u_g, done, nrm, tr_g, m_g = gpu_run(capi, planes, u0, steps, math=2, opts=o, **pk)
        assert done == steps, o
        assert rel_err(u_g, u_c) <= 1e-9, (o, rel_err(u_g, u_c))
        assert np.allclose(tr_g, tr_c, rtol=1e-9, atol=0), o
        assert np.array_equal(m_g, oracle.mask(u_c)), o
