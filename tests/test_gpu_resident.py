"""Resident mode (csv_resident_kernel.hip, option "resident" = 1): cache-resident planes iterate in LDS, one cooperative launch
per chunk of iterations, one workgroup per tile, a grid barrier per iteration.  Same parity bar as every other data flow:
level set, c1 / c2 / norm of every iteration <= 1e-9 against the oracle over the first iterations, identical stop iteration,
mask equal; plus what is specific to this flow: tiles with ragged edges, planes of one tile, chunk boundaries (a chunk writes
the tiles back and the next one reloads them), continuation in the per-launch flow and back."""
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


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def resident_ctx(capi, h, w, pk):
    ctx = capi.Context(h, w, 1, capi.make_params(**pk))
    ctx.set_option("resident", 1)
    info = ctx.launch_info()
    assert info["kernel"].startswith("csv_resident_kernel<"), info     # the shape qualifies: otherwise the test would silently test another kernel
    return ctx, info


@pytest.mark.parametrize("shape", [(16, 16), (16, 128), (32, 256), (48, 130), (96, 160), (128, 128), (130, 258), (200, 384), (256, 1024), (666, 500)])
def test_resident_small_shapes(capi, oracle, shape):
    h, w = shape
    rng = np.random.default_rng(3 * h + w)
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    u0 = oracle.checkerboard(h, w)
    pk = dict(tol=0, nu=0.01, dt=0.5)
    for steps in (1, 2, 9):
        u_c, _, nrm_c, tr_c = oracle.csv_run([img], u0, oracle.make_params(**pk), steps)
        ctx, info = resident_ctx(capi, h, w, pk)
        with ctx:
            ctx.set_option("trace", steps)
            ctx.set_image([img])
            ctx.set_levelset(u0)
            done, nrm = ctx.run(steps)
            u_g, tr_g, m_g = ctx.get_levelset(), ctx.get_trace(steps), ctx.get_mask()
            c1g, c2g = ctx.get_means()
        assert done == steps, (shape, steps, done, info)
        assert rel_err(u_g, u_c) <= 1e-9, (shape, steps, rel_err(u_g, u_c), info)
        assert np.allclose(tr_g, tr_c, rtol=1e-9, atol=0), (shape, steps)
        assert nrm == pytest.approx(nrm_c, rel=1e-9)
        assert np.array_equal(m_g, oracle.mask(u_c))
        assert c1g[0] == pytest.approx(oracle.region_mean(img, u_c, 0), rel=1e-9)
        assert c2g[0] == pytest.approx(oracle.region_mean(img, u_c, 1), rel=1e-9)


def test_resident_chunks_continuation_and_mixing(capi, oracle):
    """Chunk boundaries (enqueue 5 + 8 + 1), a second run from the result, then the per-launch flow from the resident result and
    back: all equal the oracle's 30 iterations; the resident flow is bitwise repeatable."""
    h, w = 160, 384
    img = synth.disk(160, 200, 50, noise=12, seed=9, h=h, w=w)
    u0 = oracle.checkerboard(h, w)
    pk = dict(tol=0)
    u_c, _, _, tr_c = oracle.csv_run([img], u0, oracle.make_params(**pk), 30)
    outs = []
    for rep in range(2):
        ctx, _ = resident_ctx(capi, h, w, pk)
        with ctx:
            ctx.set_option("trace", 64)
            ctx.set_image([img])
            ctx.set_levelset(u0)
            ctx.enqueue_steps(5); ctx.enqueue_steps(8); ctx.enqueue_steps(1)
            done, _, stopped = ctx.sync()
            assert done == 14 and not stopped
            assert np.allclose(ctx.get_trace(14), tr_c[:14], rtol=1e-9, atol=0)
            assert ctx.run(6)[0] == 6                     # resident, continues from the level set in memory
            ctx.set_option("resident", 0)
            assert ctx.launch_info()["kernel"].startswith("csv_wave")
            assert ctx.run(7)[0] == 7                     # per-launch flow from the resident result
            ctx.set_option("resident", 1)
            assert ctx.run(3)[0] == 3                     # and back
            outs.append(ctx.get_levelset())
    assert rel_err(outs[0], u_c) <= 1e-9
    assert np.array_equal(outs[0], outs[1])


def test_resident_stop_rule_same_iteration(capi, oracle):
    """The stop rule is booked at the grid barrier of the iteration itself: the run ends at the reference's iteration (first
    chunk, middle of a chunk), and the level set is the reference's."""
    img = synth.disk(128, 200, 50)
    u0 = oracle.checkerboard(128, 128)
    for tol in (1e-3, 0.05, 0.5):
        u_c, done_c, nrm_c, _ = oracle.csv_run([img], u0, oracle.make_params(tol=tol), 400)
        for sync_every in (1, 7, 32):
            ctx, _ = resident_ctx(capi, 128, 128, dict(tol=tol))
            with ctx:
                ctx.set_option("sync_every", sync_every)
                ctx.set_image([img])
                ctx.set_levelset(u0)
                done_g, nrm_g = ctx.run(400)
                u_g = ctx.get_levelset()
            assert done_g == done_c, (tol, sync_every, done_g, done_c)
            assert nrm_g == pytest.approx(nrm_c, rel=1e-7)
            assert rel_err(u_g, u_c) <= 1e-6


def test_resident_2048_against_the_exact_sum_oracle(capi, oracle):
    """The configuration this flow exists for: 2048 x 2048, 16 x 16 tiles of 128 x 128 on 256 CUs, 8 iterations from the checkerboard
    against the oracle with exact region sums (tests/test_gpu_fullsize.py, check (d)): <= 1e-9 on every iteration; and against
    the per-launch flow of the library."""
    n, steps = 2048, 8
    img = synth.config_planes("C4", n)
    u0 = oracle.checkerboard(n, n)
    p = oracle.make_params(tol=0)
    u_e, tr_e = u0.copy(), []
    for t in range(steps):
        nrm, c1, c2 = oracle.csv_step_exact(img, u_e, p)
        tr_e.append(list(c1) + list(c2) + [nrm])
    ctx, info = resident_ctx(capi, n, n, dict(tol=0))
    assert info["tiles_y"] == "16" and info["tiles_x"] == "16", info
    with ctx:
        ctx.set_option("trace", steps)
        ctx.set_image(img)
        ctx.set_levelset(u0)
        assert ctx.run(steps)[0] == steps
        u_g, tr_g = ctx.get_levelset(), ctx.get_trace(steps)
        ctx.set_option("resident", 0)
        ctx.set_levelset(u0)
        assert ctx.run(steps)[0] == steps
        u_l = ctx.get_levelset()
    assert rel_err(u_g, u_e) <= 1e-9, rel_err(u_g, u_e)
    assert np.allclose(tr_g, np.array(tr_e), rtol=1e-9, atol=0)
    assert rel_err(u_g, u_l) <= 1e-9


@pytest.mark.parametrize("shape,flavour", [((16, 128), 2), ((64, 256), 2), ((128, 128), 2), ((512, 512), 2), ((1024, 1024), 4), ((2048, 1024), 8),
                                           ((2048, 2048), 16)])
def test_resident_straight_line_flavours_are_the_generic_march_bit_for_bit(capi, shape, flavour):
    """Tiles of exactly 16 / 32 / 64 / 128 rows run csv_resident_kernel<2 | 4 | 8 | 16>: the same march as straight-line code (row offsets
    immediate, the band's last row known at compile time, no register rotation).  Same operations in the same order: level set and trace
    are those of the generic flavour (option "res_straight" = 0) bit for bit."""
    h, w = shape
    img = synth.disk(max(h, w), 200, 50, noise=30, seed=5, h=h, w=w)
    outs = []
    for straight in (1, 0):
        with capi.Context(h, w, 1, capi.make_params(tol=0, nu=0.01)) as ctx:
            ctx.set_option("resident", 1); ctx.set_option("res_straight", straight); ctx.set_option("trace", 25)
            assert ctx.launch_info()["kernel"] == "csv_resident_kernel<%d>" % (flavour if straight else 0)
            ctx.set_image([img]); ctx.init_checkerboard()
            assert ctx.run(25)[0] == 25
            outs.append((ctx.get_levelset(), ctx.get_trace(25)))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("shape", [(128, 128), (384, 640), (1024, 1152), (2048, 2048)])
def test_resident_release_lines_shared_or_not_same_results(capi, shape):
    """Round 4: the master releases a line per XCD (option "res_go_share" = 5; 0 = a line per tile as in round 3, 6 = one line for all).  Which line
    a tile polls is plumbing: level set, trace and the stop iteration are the same bits for every setting -- one tile, 15, 72 and 256 tiles, a chunk
    boundary in the middle, a stop rule that fires inside the second launch."""
    h, w = shape
    img = synth.disk(max(h, w), 200, 50, noise=20, seed=11, h=h, w=w)
    outs = []
    for share in (5, 0, 3, 6):
        with capi.Context(h, w, 1, capi.make_params(tol=0, nu=0.01)) as ctx:
            ctx.set_option("resident", 1); ctx.set_option("res_go_share", share); ctx.set_option("trace", 40)
            assert ctx.launch_info()["kernel"].startswith("csv_resident_kernel<")
            ctx.set_image([img]); ctx.init_checkerboard()
            ctx.enqueue_steps(9); ctx.enqueue_steps(14); ctx.sync()
            outs.append((ctx.get_levelset(), ctx.get_trace(23)))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])
    with pytest.raises(Exception):
        with capi.Context(h, w, 1, capi.make_params(tol=0)) as ctx:
            ctx.set_option("res_go_share", 7)


def test_a_batch_of_large_planes_in_long_chunks_takes_the_resident_flow(capi):
    """End of round 4: in a batch (other co-resident contexts on the device) an enqueue still takes the resident flow when the plane is large and
    the enqueue is long -- one cooperative launch after the other then beats interleaved per-launch flows (tools/batch_probe.py: eight 1536^2
    planes 10.8 vs 11.0 us per image-iteration in chunks of 100, 2048^2 13.2 vs 16.2).  Short enqueues and smaller planes keep the per-launch
    flow; the choice is taken per enqueue (a long warm-up chunk must not leave a run with 8-iteration cooperative launches); the two flows
    continue each other on one context and agree to 1e-9."""
    n = 1536
    imgs = [synth.disk(n, 200, 50, noise=10, seed=s) for s in (21, 22)]
    with capi.Context(n, n, 1, capi.make_params(tol=0)) as alone:
        alone.set_image([imgs[0]]); alone.init_checkerboard()
        alone.enqueue_steps(120); alone.enqueue_steps(8); alone.sync()
        assert alone.launch_info()["kernel"].startswith("csv_resident_kernel<")
        ref = alone.get_levelset()
    a = capi.Context(n, n, 1, capi.make_params(tol=0)); b = capi.Context(n, n, 1, capi.make_params(tol=0))
    try:
        a.set_image([imgs[0]]); a.init_checkerboard()
        b.set_image([imgs[1]]); b.init_checkerboard()
        assert a.launch_info()["kernel"].startswith("csv_wave")            # a batch, nothing announced yet: the per-launch flow is what would run
        a.enqueue_steps(120)                                               # long enqueue, large plane: resident
        assert a.launch_info()["kernel"].startswith("csv_resident_kernel<")
        b.enqueue_steps(8)                                                 # short enqueue: per launch
        assert b.launch_info()["kernel"].startswith("csv_wave")
        a.enqueue_steps(8); b.enqueue_steps(120)                           # per enqueue, not per run
        assert a.launch_info()["kernel"].startswith("csv_wave") and b.launch_info()["kernel"].startswith("csv_resident_kernel<")
        assert a.sync()[0] == 128 and b.sync()[0] == 128
        assert rel_err(a.get_levelset(), ref) <= 1e-9                      # 120 resident + 8 per launch against 128 resident
        assert b.run(150)[0] == 150                                        # cvh_run: its chunks are long
        assert b.launch_info()["kernel"].startswith("csv_resident_kernel<")
    finally:
        a.close(); b.close()
    with capi.Context(1024, 1024, 1, capi.make_params(tol=0)) as c1, capi.Context(1024, 1024, 1, capi.make_params(tol=0)) as c2:
        for c in (c1, c2): c.set_image([synth.disk(1024, 200, 50, noise=10, seed=3)]); c.init_checkerboard()
        c1.enqueue_steps(400); c1.sync()
        assert c1.launch_info()["kernel"].startswith("csv_wave")           # 1024^2 in a batch: never resident by itself


def test_priority_by_quarters_is_scheduling_only(capi):
    """Option "res_prio" (both resident kernels: a wave lowers its s_setprio level with every quarter of its band) changes WHEN a wave runs, never what
    it computes: level set, trace and the smoothed plane are the same bits with and without."""
    n = 1024
    img = synth.disk(n, 200, 50, noise=25, seed=13)
    outs = []
    for prio in (1, 0):
        with capi.Context(n, n, 1, capi.make_params(tol=0, nu=0.01)) as ctx:
            ctx.set_option("resident", 1); ctx.set_option("res_prio", prio); ctx.set_option("pm_kernel", 4); ctx.set_option("trace", 20)
            ctx.set_image([img]); ctx.perona_malik(30.0, 0.25, 5.0)
            smooth = ctx.get_image()[0]
            assert ctx.launch_info(1)["kernel"].startswith("pm_resident_kernel<") and ctx.launch_info()["kernel"].startswith("csv_resident_kernel<")
            ctx.init_checkerboard()
            assert ctx.run(20)[0] == 20
            outs.append((smooth, ctx.get_levelset(), ctx.get_trace(20)))
    assert all(np.array_equal(x, y) for x, y in zip(outs[0], outs[1]))


def test_automatic_flow_steps_aside_for_a_batch(capi, oracle):
    """Round 4: the automatic choice of the resident flow looks at the device's live contexts.  One 256^2 context alone: the resident kernel.
    Two contexts that hold an image and a level set (a batch): cooperative launches of different contexts would serialise (tools/batch_probe.py:
    32.6 vs 16.2 us per image-iteration at 2048^2), so a run that starts then takes the per-launch flow; a scratch context ("co_resident" = 0)
    does not count; results agree with the oracle either way."""
    n = 256
    imgs = [synth.disk(n, 200, 50, noise=6, seed=s) for s in (3, 4)]
    u0 = oracle.checkerboard(n, n)
    a = capi.Context(n, n, 1, capi.make_params(tol=0))
    a.set_image([imgs[0]]); a.set_levelset(u0)
    assert a.launch_info()["kernel"].startswith("csv_resident_kernel<")
    b = capi.Context(n, n, 1, capi.make_params(tol=0))
    b.set_option("co_resident", 0)
    b.set_image([imgs[1]]); b.set_levelset(u0)
    assert a.launch_info()["kernel"].startswith("csv_resident_kernel<")          # b is a scratch context
    b.set_option("co_resident", 1)
    assert a.launch_info()["kernel"].startswith("csv_wave") and b.launch_info()["kernel"].startswith("csv_wave")
    for _ in range(2):
        a.enqueue_steps(5); b.enqueue_steps(5)
    for c, im in ((a, imgs[0]), (b, imgs[1])):
        done, _, stopped = c.sync()
        assert done == 10 and not stopped
        u_c, _, _, _ = oracle.csv_run([im], u0, oracle.make_params(tol=0), 10)
        assert np.abs(c.get_levelset() - u_c).max() <= 1e-9 * np.abs(u_c).max()
    b.close()
    a.set_levelset(u0)                                                            # a new run, alone again
    assert a.launch_info()["kernel"].startswith("csv_resident_kernel<")
    a.close()
