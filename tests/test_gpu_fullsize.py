"""GPU parity at the PRODUCTION launch geometry and over long runs (run with -m gpu on an MI355X).

The small-shape tests (test_gpu_parity.py) cannot reach what bench.py launches: 765 workgroups, 90 strips x 33
wave-columns, XCD renumbering with a grid that is not a multiple of 8, hipGraph chunks of 16 steps.  Here the
default kernel / geometry / graph path runs the BASELINE.json configurations at full size against the CPU oracle
(the oracle costs ~0.8 s per iteration at 4096^2, so the oracle-checked runs are 5-16 iterations), plus long runs
checked at the end-of-run tolerance and through size-independent properties.

Tolerances: max|u_gpu - u_cpu| / max|u_cpu| <= 1e-9 over the first iterations, <= 1e-6 after the
configured count; c1/c2/norm relative <= 1e-9 per iteration; mask equal (IoU >= 0.999 after long runs)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from chan_vese_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_LEGS = {}    # oracle results shared between parametrised cases of one test


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


def gpu_steps(capi, planes, u0, steps, pk, opts=None, via_enqueue=False):
    """Default kernel, default geometry, default (graph) launch path."""
    h, w = planes[0].shape
    with capi.Context(h, w, len(planes), capi.make_params(**pk)) as ctx:
        for k, v in (opts or {}).items():
            ctx.set_option(k, v)
        ctx.set_option("trace", steps)
        ctx.set_image(planes)
        ctx.set_levelset(u0)
        if via_enqueue:          # what bench.py does: chunks of 16 = one hipGraph each
            done = 0
            while done < steps:
                c = min(16, steps - done)
                ctx.enqueue_steps(c)
                done += c
            done, nrm, _ = ctx.sync()
        else:
            done, nrm = ctx.run(steps)
        return ctx.get_levelset(), done, nrm, ctx.get_trace(steps), ctx.get_mask()


def oracle_trajectories(oracle, planes, pk, steps, restart_at=(), cache_key=None):
    """The oracle's two free runs from the checkerboard: reference-order sums (with the snapshots the restart checks need) and exact sums.
    Cached under `cache_key` so that two tests of one image share the CPU leg."""
    if cache_key is not None and cache_key in _ORACLE_LEGS:
        return _ORACLE_LEGS[cache_key]
    h, w = planes[0].shape
    u0 = oracle.checkerboard(h, w)
    p = oracle.make_params(**pk)
    u = u0.copy()
    keep, tr_c = {}, []
    for t in range(1, steps + 1):
        if (t - 1) in restart_at:
            keep[t - 1] = u.copy()
        nrm, c1, c2 = oracle.csv_step(planes, u, p)      # c1/c2 = the means this iteration used
        tr_c.append(list(c1) + list(c2) + [nrm])
        if t == 1 or (t - 1) in restart_at:
            keep[("after", t - 1)] = u.copy()
    u_e, tr_e = u0.copy(), []
    for t in range(1, steps + 1):
        nrm, c1, c2 = oracle.csv_step_exact(planes, u_e, p)
        tr_e.append(list(c1) + list(c2) + [nrm])
    leg = dict(u0=u0, u=u, keep=keep, tr_c=np.array(tr_c), u_e=u_e, tr_e=np.array(tr_e), steps=steps)
    if cache_key is not None:
        _ORACLE_LEGS[cache_key] = leg
    return leg


def check_against_oracle(capi, oracle, planes, pk, steps, restart_at, via_enqueue=False, opts=None, cache_key=None):
    """Four checks of one configuration against the oracle's trajectory u_1 .. u_steps:
    (a) ONE GPU iteration from the initial level set: <= 1e-12 (nothing to amplify yet);
    (b) restarts: the oracle's own u_k uploaded, ONE GPU iteration, compared with the oracle's u_{k+1}: <= 1e-9 and
        c1/c2/norm relative <= 1e-9 -- the sharp test of the launch geometry, free of the recurrence's amplification;
    (c) the free run of `steps` iterations against the REFERENCE-ORDER oracle: <= 1e-6 (SURVEY.md §8d end-of-run
        tolerance), mask IoU >= 0.999;
    (d) the same free run against the oracle with EXACT region sums (cvo_csv_step_exact: the reference's per-pixel terms
        added without accumulation error): level set and c1/c2/norm of EVERY iteration <= 1e-9.
    Why (c) cannot be 1e-9 at 4096^2 and (d) can: after iteration 1 of a checkerboard start c1 and c2 agree to 2.7e-5
    relative (79.4504 vs 79.4526), the region term is proportional to c1 - c2, and the reference's sequential 16.7 M-term
    double sums of the few distinct values of that level set carry a SYSTEMATIC rounding error of 3e-10 relative
    (measured against the compensated sums: test_region_sums_adjudicated_at_4096) -- 3e-8 of max|u| at iteration 2.
    The GPU's fixed-point / tree sums agree with the exact sums to <= 1e-13."""
    h, w = planes[0].shape
    leg = oracle_trajectories(oracle, planes, pk, steps, restart_at, cache_key)
    u0, u, keep, tr_c, u_e, tr_e = leg["u0"], leg["u"], leg["keep"], leg["tr_c"], leg["u_e"], leg["tr_e"]
    with capi.Context(h, w, len(planes), capi.make_params(**pk)) as ctx:
        for k, v in (opts or {}).items():
            ctx.set_option(k, v)
        ctx.set_option("trace", steps)
        ctx.set_image(planes)
        # (a) + (b)
        for k in [0] + sorted(restart_at):
            ctx.set_levelset(u0 if k == 0 else keep[k])
            done, nrm = ctx.run(1)
            assert done == 1
            err = rel_err(ctx.get_levelset(), keep[("after", k)])
            assert err <= (1e-12 if k == 0 else 1e-9), (k, err)
            tr = ctx.get_trace(1)[0]
            assert np.allclose(tr, tr_c[k], rtol=1e-9, atol=0), (k, tr, tr_c[k])
        # (c)
        ctx.set_levelset(u0)
        if via_enqueue:          # what bench.py does: chunks of 16 = one hipGraph each
            done = 0
            while done < steps:
                c = min(16, steps - done)
                ctx.enqueue_steps(c)
                done += c
            done, nrm, _ = ctx.sync()
        else:
            done, nrm = ctx.run(steps)
        assert done == steps
        u_g, tr_g, m_g = ctx.get_levelset(), ctx.get_trace(steps), ctx.get_mask()
    assert rel_err(u_g, u) <= 1e-6, rel_err(u_g, u)
    assert np.allclose(tr_g[0], tr_c[0], rtol=1e-9, atol=0)
    assert np.allclose(tr_g, tr_c, rtol=1e-5, atol=0), np.abs(tr_g / tr_c - 1).max()    # every iteration against the REFERENCE's own summation order
    assert iou(m_g, oracle.mask(u)) >= 0.999
    # (d) -- measured against this repository's own adjudicator (compensated long double sums), not against the reference's order
    assert rel_err(u_g, u_e) <= 1e-9, rel_err(u_g, u_e)
    assert np.allclose(tr_g, tr_e, rtol=1e-9, atol=0), np.abs(tr_g / tr_e - 1).max()
    assert np.array_equal(m_g, oracle.mask(u_e))


def test_region_sums_adjudicated_at_4096(capi, oracle):
    """Who is right about c1/c2 at 4096^2?  Iterations 1-3 of the checkerboard start on the GPU (default kernel, chain
    mode: 64-bit fixed-point sums); after each, the GPU's region means of ITS OWN level set against (i) the exact sums
    of that level set (the reference's per-pixel terms, compensated long double) and (ii) the reference-order sequential
    double sums of the same level set.  The GPU agrees with the exact sums to <= 1e-13; the sequential sums are the
    ones that are off (3e-10 after iteration 1: the level set then holds few distinct values, the rounding error of
    every addition has the same sign)."""
    n = 4096
    planes = synth.config_planes("C2", n)
    errs = []
    with capi.Context(n, n, 1, capi.make_params(tol=0)) as ctx:
        ctx.set_image(planes)
        ctx.init_checkerboard()
        for k in range(1, 4):
            done, _ = ctx.run(1)               # continues from the current level set
            assert done == 1
            u_g = ctx.get_levelset()
            c_g = np.concatenate(ctx.get_means())
            c_e = np.concatenate(oracle.region_means_exact(planes, u_g))
            c_p = np.concatenate(oracle.region_means(planes, u_g))
            errs.append((np.abs(c_g / c_e - 1).max(), np.abs(c_p / c_e - 1).max()))
    print("after iteration k: |c_gpu/c_exact - 1|, |c_sequential/c_exact - 1|:", errs)
    assert all(e_gpu <= 1e-13 for e_gpu, _ in errs), errs
    assert all(e_seq <= 1e-8 for _, e_seq in errs), errs
    assert errs[0][1] >= 1e-11 and errs[0][0] < errs[0][1], errs    # the sequential sums carry the error, not the GPU's


def test_config2_4096_one_channel_bench_geometry(capi, oracle):
    """BASELINE configs[1] at full size: 4096x4096x1 disk, checkerboard init, the launch bench.py times
    (default kernel and geometry: 765 workgroups, hipGraph of 16 steps + 2 plain launches), 18 iterations."""
    check_against_oracle(capi, oracle, synth.config_planes("C2", 4096), dict(tol=0), 18, {3, 17}, via_enqueue=True)


def test_config5_image_4096_noisy(capi, oracle):
    """One image of BASELINE configs[4] (noise 16, seed 1003, radius 1020) at full size, 6 iterations via cvh_run."""
    check_against_oracle(capi, oracle, [synth.batch_image(3, 4096)], dict(tol=0), 6, {2, 5}, cache_key="c5_image3")


def test_config5_batch_of_eight_interleaved(capi, oracle):
    """BASELINE configs[4] as ONE GPU sees it: eight 4096^2 contexts (images 0..7 of the batch) resident at once (the library's automatic
    cache policy picks plain stores from the device's live footprint: no option set), their launches interleaved on eight streams --
    a first round of 6 iterations each, then rounds of 8, 8 and 2: 24 iterations.  Image 3 after the first round against the oracle
    (the leg test_config5_image_4096_noisy computes: reference-order sums <= 1e-6, exact sums <= 1e-9, trace rows, mask); after 24
    iterations EVERY level set, trace and mask bitwise equal to the same image run alone in one cvh_run (other chunking, nothing else
    on the GPU): images on one GPU do not see each other."""
    n, total = 4096, 24
    imgs = [synth.batch_image(b, n) for b in range(8)]
    leg = oracle_trajectories(oracle, [imgs[3]], dict(tol=0), 6, {2, 5}, cache_key="c5_image3")
    ctxs = []
    try:
        for b in range(8):
            ctx = capi.Context(n, n, 1, capi.make_params(tol=0))
            ctx.set_option("trace", total)
            ctx.set_image([imgs[b]])
            ctx.init_checkerboard()
            ctxs.append(ctx)
        # the automatic choices see all eight (8 x 285 MB do not live in the 256 MiB Infinity Cache): plain stores, equal strips; no knob set
        assert all(ctx.launch_info()["kernel"] == "csv_wave2_kernel<1, true, 3, 0, false>" for ctx in ctxs)
        for ctx in ctxs:
            ctx.enqueue_steps(6)
        done, _, stopped = ctxs[3].sync()
        assert done == 6 and not stopped
        u3, tr3, m3 = ctxs[3].get_levelset(), ctxs[3].get_trace(6), ctxs[3].get_mask()
        assert rel_err(u3, leg["u"]) <= 1e-6 and rel_err(u3, leg["u_e"]) <= 1e-9, (rel_err(u3, leg["u"]), rel_err(u3, leg["u_e"]))
        assert np.allclose(tr3, leg["tr_c"], rtol=1e-5, atol=0) and np.allclose(tr3, leg["tr_e"], rtol=1e-9, atol=0)
        assert np.array_equal(m3, oracle.mask(leg["u_e"]))
        for c in (8, 8, 2):
            for ctx in ctxs:
                ctx.enqueue_steps(c)
        batch = []
        for ctx in ctxs:
            done, _, stopped = ctx.sync()
            assert done == total and not stopped
            batch.append((ctx.get_levelset(), ctx.get_trace(total), ctx.get_mask()))
    finally:
        for ctx in ctxs:
            ctx.close()
    for b in range(8):
        with capi.Context(n, n, 1, capi.make_params(tol=0)) as ctx:
            ctx.set_option("trace", total)
            ctx.set_option("wave_cskew", 0)      # the batch's strip table (equal strips): the workgroups' partial sums round where the strips end
            ctx.set_image([imgs[b]])
            ctx.init_checkerboard()
            assert ctx.launch_info()["kernel"] == "csv_wave2_kernel<1, true, 3, 1, false>"     # alone: write-through stores (same values, other cache policy)
            done, _ = ctx.run(total)
            assert done == total
            assert np.array_equal(ctx.get_levelset(), batch[b][0]), b
            assert np.array_equal(ctx.get_trace(total), batch[b][1]), b
            assert np.array_equal(ctx.get_mask(), batch[b][2]), b


def test_config3_4096_three_channel(capi, oracle):
    """BASELINE configs[2] at full size: 4096x4096x3, per-channel lambda, 5 iterations."""
    check_against_oracle(capi, oracle, synth.config_planes("C3", 4096), dict(tol=0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1]), 5, {2, 4})


def test_config4_2048_pm_then_csv(capi, oracle):
    """BASELINE configs[3] at full size, shortened: the 2048x2048 noisy disk, Perona-Malik K=30 L=0.25 T=5 (20 steps,
    uint8 result compared exactly up to the round-half-even boundary), then 10 CSV iterations on the smoothed plane."""
    n = 2048
    img = synth.config_planes("C4", n)
    pm_c = oracle.perona_malik(img, 30.0, 0.25, 5.0)
    u0 = oracle.checkerboard(n, n)
    pk = dict(tol=0)
    with capi.Context(n, n, 1, capi.make_params(**pk)) as ctx:
        ctx.set_image(img)
        ctx.perona_malik(30.0, 0.25, 5.0)
        pm_g = ctx.get_image()
        diff = pm_g[0].astype(int) - pm_c[0].astype(int)
        assert np.abs(diff).max() <= 1 and (diff != 0).mean() <= 1e-6      # FAST: <= 1 LSB on <= 1e-6 of the pixels
        ctx.set_image(pm_c)       # identical input for the CSV part even if a boundary pixel rounded the other way
        ctx.set_option("trace", 10)
        ctx.set_levelset(u0)
        done, nrm = ctx.run(10)
        u_g, tr_g, m_g = ctx.get_levelset(), ctx.get_trace(10), ctx.get_mask()
    u_c, _, nrm_c, tr_c = oracle.csv_run(pm_c, u0, oracle.make_params(**pk), 10)
    assert done == 10
    assert rel_err(u_g, u_c) <= 1e-6          # free run vs the reference-order sums: see check_against_oracle (c)
    assert np.allclose(tr_g[0], tr_c[0], rtol=1e-9, atol=0)
    assert iou(m_g, oracle.mask(u_c)) >= 0.999
    u_e, tr_e = u0.copy(), []                 # free run vs the exact-sum oracle: check_against_oracle (d)
    for t in range(10):
        nrm, c1, c2 = oracle.csv_step_exact(pm_c, u_e, oracle.make_params(**pk))
        tr_e.append(list(c1) + list(c2) + [nrm])
    assert rel_err(u_g, u_e) <= 1e-9, rel_err(u_g, u_e)
    assert np.allclose(tr_g, np.array(tr_e), rtol=1e-9, atol=0)


@pytest.mark.parametrize("mode,math", [("fast", 2), ("strict", 1)])
def test_config4_configured_length(capi, oracle, mode, math):
    """BASELINE configs[3] at its CONFIGURED length: Perona-Malik K=30 L=0.25 T=250 (1000 trips of the floating-point
    loop, src/main.cpp:498) on the 2048x2048 noisy disk against the oracle's uint8 plane (STRICT: equal; FAST: <= 1 LSB
    on <= 1e-6 of the pixels), then 200 CSV iterations on the smoothed plane (src/main.cpp:963) against the
    reference-order oracle at the end-of-run tolerance 1e-6, mask IoU >= 0.999."""
    n = 2048
    img = synth.config_planes("C4", n)
    key, cache = "c4", _ORACLE_LEGS
    if key not in cache:      # the oracle leg (~75 s of CPU) is shared by the two arithmetic flavours
        assert oracle.pm_trip_count(0.25, 250.0) == 1000
        pm_c = oracle.perona_malik(img, 30.0, 0.25, 250.0)
        u_c, done_c, nrm_c, tr_c = oracle.csv_run(pm_c, oracle.checkerboard(n, n), oracle.make_params(tol=0), 200)
        assert done_c == 200
        cache[key] = (pm_c, u_c, tr_c)
    pm_c, u_c, tr_c = cache[key]
    with capi.Context(n, n, 1, capi.make_params(tol=0)) as ctx:
        ctx.set_option("math_mode", math)
        ctx.set_image(img)
        ctx.perona_malik(30.0, 0.25, 250.0)
        pm_g = ctx.get_image()
        diff = pm_g[0].astype(int) - pm_c[0].astype(int)
        if mode == "strict":
            assert not diff.any()
        else:
            assert np.abs(diff).max() <= 1 and (diff != 0).mean() <= 1e-6, (np.abs(diff).max(), (diff != 0).sum())
        ctx.set_image(pm_c)       # identical input for the CSV part even if a boundary pixel rounded the other way
        ctx.set_option("trace", 200)
        ctx.init_checkerboard()
        done, nrm = ctx.run(200)
        u_g, tr_g, m_g = ctx.get_levelset(), ctx.get_trace(200), ctx.get_mask()
    assert done == 200
    assert rel_err(u_g, u_c) <= 1e-6, rel_err(u_g, u_c)
    assert np.allclose(tr_g[0], tr_c[0], rtol=1e-9, atol=0)
    assert np.allclose(tr_g, tr_c, rtol=1e-6, atol=0), np.abs(tr_g / tr_c - 1).max()
    assert iou(m_g, oracle.mask(u_c)) >= 0.999


def test_512_disk_500_iterations_drift(capi, oracle):
    """Long-run drift (SURVEY §8d "re-measure at 500"): 512x512 disk, 500 iterations, end-of-run tolerance 1e-6,
    IoU >= 0.999 (expected 1.0), c1/c2/norm of every iteration within 1e-6 relative (the recurrence amplifies the
    1e-16-level differences of the first iterations)."""
    n, steps = 512, 500
    img = synth.config_planes("C1", n)
    u0 = oracle.checkerboard(n, n)
    pk = dict(tol=0)
    u_c, done_c, nrm_c, tr_c = oracle.csv_run(img, u0, oracle.make_params(**pk), steps)
    u_g, done_g, nrm_g, tr_g, m_g = gpu_steps(capi, img, u0, steps, pk)
    assert done_g == done_c == steps
    assert rel_err(u_g, u_c) <= 1e-6, rel_err(u_g, u_c)
    assert np.allclose(tr_g[:10], tr_c[:10], rtol=1e-9, atol=0)
    assert np.allclose(tr_g, tr_c, rtol=1e-6, atol=0)
    assert iou(m_g, oracle.mask(u_c)) >= 0.999
    assert iou(m_g, synth.disk(n) > 100) == 1.0


def check_against_fullsize_fixture(fx, u_g, tr_g, m_g, steps):
    """The GPU run at the CONFIGURED length against the oracle fixture of tests/golden/make_golden_4096.py (the oracle needs 25-30 minutes
    per variant at 4096^2: generated in the build container, committed compact): c1 / c2 / norm of EVERY iteration against the
    reference-order oracle <= 1e-6 and against the exact-sum oracle <= 1e-9 over the first 20 iterations (<= 1e-7 to the end: the two
    oracles themselves part by 1.3e-9 of max|u| over 500 iterations); the level set on the fixture's stride-16 grid and its eight full
    rows <= 1e-6 max|u| (reference order) and <= 1e-7 (exact sums); the mask against the oracle's packed mask: IoU >= 0.999, flips reported."""
    n = int(fx["n"][0])
    assert int(fx["iterations"][0]) == steps and u_g.shape == (n, n)
    oi, oj, st = (int(v) for v in fx["grid_offset_stride"])
    rows = [int(r) for r in fx["rows_index"]]
    out = {}
    for variant, tol_trace, tol_u in (("ref", 1e-6, 1e-6), ("exact", 1e-7, 1e-7)):
        tr_o, umax = fx[f"trace_{variant}"], float(fx[f"umax_{variant}"][0])
        e_tr = float(np.abs(tr_g / tr_o - 1).max())
        e_grid = float(np.abs(u_g[oi::st, oj::st] - fx[f"grid_{variant}"]).max() / umax)
        e_rows = float(np.abs(u_g[rows] - fx[f"rows_{variant}"]).max() / umax)
        out[variant] = (e_tr, e_grid, e_rows)
        assert e_tr <= tol_trace and e_grid <= tol_u and e_rows <= tol_u, (variant, out)
    assert np.allclose(tr_g[:20], fx["trace_exact"][:20], rtol=1e-9, atol=0), np.abs(tr_g[:20] / fx["trace_exact"][:20] - 1).max()
    m_o = np.unpackbits(fx["mask_bits_ref"]).reshape(n, n).astype(bool)
    flips = int((m_g.astype(bool) != m_o).sum())
    print(f"configured length {steps}: trace / grid / rows vs reference-order oracle {out['ref']}, vs exact-sum oracle {out['exact']}; mask flips {flips}")
    assert iou(m_g, m_o) >= 0.999 and flips <= 16, flips
    assert int(fx["mask_exact_differs_at"].size) == 0        # (the two oracles' masks agree pixel for pixel)


def test_4096_500_iterations_properties(capi, oracle, golden_dir):
    """BASELINE configs[1] for its full 500 iterations: bitwise repeatable across two chunkings (fixed-order reductions, no float
    atomics), the mask IS the disk, c1/c2 settle at the disk's foreground / background as the 512^2 oracle trajectory does -- and,
    from round 4 on, AGAINST THE ORACLE AT THE CONFIGURED LENGTH through the committed fixture c2_4096_500.npz (trace of all 500
    iterations, strided level set, full rows, packed mask; check_against_fullsize_fixture)."""
    n, steps = 4096, 500
    img = synth.config_planes("C2", n)
    u0 = oracle.checkerboard(n, n)
    pk = dict(tol=0)
    u_a, done_a, nrm_a, tr_a, m_a = gpu_steps(capi, img, u0, steps, pk, via_enqueue=True)
    u_b, done_b, nrm_b, tr_b, m_b = gpu_steps(capi, img, u0, steps, pk)      # cvh_run: other chunking, same arithmetic
    assert done_a == done_b == steps
    assert np.array_equal(u_a, u_b) and np.array_equal(tr_a, tr_b) and nrm_a == nrm_b
    check_against_fullsize_fixture(np.load(os.path.join(golden_dir, "c2_4096_500.npz")), u_a, tr_a, m_a, steps)
    # the contour IS the disk's edge; which side ends up positive is decided by the sign of c1 - c2 after the first
    # iteration of the symmetric checkerboard start (at 4096^2 the background becomes the "inside", at 512^2 the disk)
    disk = img[0] > 100
    assert max(iou(m_a, disk), iou(m_a, ~disk)) == 1.0
    assert np.all(np.isfinite(u_a))
    lo, hi = sorted((tr_a[-1, 0], tr_a[-1, 1]))
    assert abs(hi - 200) < 2.5 and abs(lo - 50) < 0.5          # region means of the segmented disk / background
    tr512 = np.load(os.path.join(golden_dir, "traj_512_disk.npz"))["trace"]   # oracle trajectory of the same disk family at 512^2
    # iteration 100: c1/c2 agree up to the discretisation of the disk edge (a thinner share of the pixels at 4096^2)
    lo100, hi100 = sorted((tr_a[99, 0], tr_a[99, 1]))
    assert abs(tr512[-1, 0] - hi100) < 2.0 and abs(tr512[-1, 1] - lo100) < 0.5


def test_config3_4096_300_iterations_against_the_fixture(capi, oracle, golden_dir):
    """BASELINE configs[2] at its configured length: 4096^2 x 3 channels (per-channel lambdas), 300 iterations from the checkerboard with
    the default kernel and launch path, against the oracle fixture c3_4096_300.npz (check_against_fullsize_fixture)."""
    n, steps = 4096, 300
    pk = dict(tol=0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1])
    u_g, done, nrm, tr_g, m_g = gpu_steps(capi, synth.config_planes("C3", n), oracle.checkerboard(n, n), steps, pk, via_enqueue=True)
    assert done == steps
    check_against_fullsize_fixture(np.load(os.path.join(golden_dir, "c3_4096_300.npz")), u_g, tr_g, m_g, steps)


def state32_stats(u_g, u_c, umax):
    """SURVEY.md 8(d) bar of the FP32-state mode: median |du| / max|u| <= 1e-4; p99 and the maximum are reported (a per-pixel maximum is
    meaningless there: the survey's probe saw 49 % on isolated pixels with 0 mask flips)."""
    d = np.abs(u_g - u_c).ravel() / umax
    return float(np.median(d)), float(np.percentile(d, 99)), float(d.max())


def test_state32_configs_1_and_4_against_the_oracle(capi, oracle):
    """FP32-state mode on BASELINE configs[0] (512^2, 100 iterations) and configs[3] (2048^2: Perona-Malik 1000 steps, then 200 iterations on
    the smoothed plane; the oracle leg is test_config4_configured_length's): IoU >= 0.999, median |du| / max|u| <= 1e-4."""
    n = 512
    img = synth.config_planes("C1", n)
    u_c, done, _, _ = oracle.csv_run(img, oracle.checkerboard(n, n), oracle.make_params(tol=0), 100)
    with capi.Context(n, n, 1, capi.make_params(tol=0)) as ctx:
        ctx.set_option("state", 32)
        ctx.set_image(img)
        ctx.init_checkerboard()
        assert ctx.run(100)[0] == 100
        u_g, m_g = ctx.get_levelset(), ctx.get_mask()
    med, p99, mx = state32_stats(u_g, u_c, np.abs(u_c).max())
    print(f"state 32, C1 512^2 x 100: median {med:.2e}, p99 {p99:.2e}, max {mx:.2e} of max|u|; mask flips {int((m_g != oracle.mask(u_c)).sum())}")
    assert med <= 1e-4 and iou(m_g, oracle.mask(u_c)) >= 0.999
    n = 2048
    img = synth.config_planes("C4", n)
    if "c4" not in _ORACLE_LEGS:
        pm_c = oracle.perona_malik(img, 30.0, 0.25, 250.0)
        u_c, done_c, nrm_c, tr_c = oracle.csv_run(pm_c, oracle.checkerboard(n, n), oracle.make_params(tol=0), 200)
        _ORACLE_LEGS["c4"] = (pm_c, u_c, tr_c)
    pm_c, u_c, tr_c = _ORACLE_LEGS["c4"]
    with capi.Context(n, n, 1, capi.make_params(tol=0)) as ctx:
        ctx.set_option("state", 32)
        ctx.set_image(pm_c)
        ctx.init_checkerboard()
        assert ctx.launch_info()["kernel"] == "csv_wave2_kernel<1, true, 3, 1, true>"
        assert ctx.run(200)[0] == 200
        u_g, m_g = ctx.get_levelset(), ctx.get_mask()
    med, p99, mx = state32_stats(u_g, u_c, np.abs(u_c).max())
    print(f"state 32, C4 2048^2 x 200: median {med:.2e}, p99 {p99:.2e}, max {mx:.2e} of max|u|; mask flips {int((m_g != oracle.mask(u_c)).sum())}")
    assert med <= 1e-4 and iou(m_g, oracle.mask(u_c)) >= 0.999


@pytest.mark.parametrize("cfg,stem,steps", [("C2", "c2_4096_500", 500), ("C3", "c3_4096_300", 300)])
def test_state32_configs_2_and_3_against_the_fixture(capi, oracle, golden_dir, cfg, stem, steps):
    """FP32-state mode on BASELINE configs[1] / [2] at their configured length against the FP64 oracle fixture: the statistics over the
    fixture's level-set samples (stride-16 grid + eight full rows: 98 k pixels), the mask against the oracle's packed mask."""
    n = 4096
    fx = np.load(os.path.join(golden_dir, stem + ".npz"))
    pk = dict(tol=0) if cfg == "C2" else dict(tol=0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1])
    planes = synth.config_planes(cfg, n)
    with capi.Context(n, n, len(planes), capi.make_params(**pk)) as ctx:
        ctx.set_option("state", 32)
        ctx.set_image(planes)
        ctx.init_checkerboard()
        assert ctx.launch_info()["kernel"].endswith("true>")
        assert ctx.run(steps)[0] == steps
        u_g, m_g = ctx.get_levelset(), ctx.get_mask()
    oi, oj, st = (int(v) for v in fx["grid_offset_stride"])
    rows = [int(r) for r in fx["rows_index"]]
    got = np.concatenate([u_g[oi::st, oj::st].ravel(), u_g[rows].ravel()])
    want = np.concatenate([fx["grid_ref"].ravel(), fx["rows_ref"].ravel()])
    med, p99, mx = state32_stats(got, want, float(fx["umax_ref"][0]))
    m_o = np.unpackbits(fx["mask_bits_ref"]).reshape(n, n).astype(bool)
    flips = int((m_g.astype(bool) != m_o).sum())
    print(f"state 32, {cfg} 4096^2 x {steps}: median {med:.2e}, p99 {p99:.2e}, max {mx:.2e} of max|u| over {got.size} samples; mask flips {flips}")
    assert med <= 1e-4 and iou(m_g, m_o) >= 0.999


@pytest.mark.parametrize("shape", [(1, 144), (144, 1), (3, 700), (100, 517), (150, 530), (9, 272), (64, 2016)])
@pytest.mark.parametrize("mode,math", [("strict", 1), ("fast", 2)])
def test_csv_three_channel_edge_shapes(capi, oracle, shape, mode, math):
    """3 channels on the shapes the 1-channel kernels are tested on: one row, one column, ragged widths (w % 16 != 0),
    ragged strips, a width that is an exact multiple of the wave stride."""
    h, w = shape
    rng = np.random.default_rng(h * 31337 + w)
    planes = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(3)]
    u0 = oracle.checkerboard(h, w) if min(h, w) > 2 else rng.normal(size=shape)
    pk = dict(tol=0, lambda1=[1, 0.8, 0.5], lambda2=[0.7, 0.5, 1], nu=0.01)
    for steps, opts in ((1, {}), (7, dict(strip_rows=8)), (4, dict(kernel=2))):
        u_c, _, nrm_c, tr_c = oracle.csv_run(planes, u0, oracle.make_params(**pk), steps)
        with capi.Context(h, w, 3, capi.make_params(**pk)) as ctx:
            ctx.set_option("math_mode", math)
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_option("trace", steps)
            ctx.set_image(planes)
            ctx.set_levelset(u0)
            done, nrm = ctx.run(steps)
            u_g, tr_g, m_g = ctx.get_levelset(), ctx.get_trace(steps), ctx.get_mask()
        assert done == steps
        assert rel_err(u_g, u_c) <= 1e-9, (steps, opts, rel_err(u_g, u_c))
        assert np.allclose(tr_g, tr_c, rtol=1e-9, atol=0)
        assert np.array_equal(m_g, oracle.mask(u_c))


def test_bench_gpus_2_gloo_on_one_gpu():
    """`python bench.py --gpus 2` started directly spawns two ranks (both on device 0 under
    CHANVESE_DIST_BACKEND=gloo), runs the batch-shard workload on the GPU and prints one line with n_gpus == 2."""
    env = dict(os.environ, CHANVESE_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "CHANVESE_BENCH_DRYRUN"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "512", "--steps", "40",
                          "--warmup", "8", "--images-per-gpu", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_in_group"] == 2 and d["data"] == "synthetic"
    assert d["config"]["images_total"] == 4 and len(d["config"]["per_rank_mpx_it_s"]) == 2
    rl = d["roofline"]          # two images per rank stream side by side: the automatic flow is a launch per iteration (HBM roofline), not the resident kernel
    assert d["value"] > 0 and rl["bound"] == "hbm" and 0 < rl["frac"] <= 1 and rl["kernel"].startswith("csv_wave")


def test_bench_one_rank_over_rccl():
    """The RCCL branch of batch.init_distributed on the hardware that exists: ONE fresh child process (WORLD_SIZE=1,
    CHANVESE_DIST_FORCE=1) runs bench.py with backend "nccl" -- init_process_group(nccl, device_id), the barriers around
    the timed region, the max-all-reduce of the elapsed time and the all-gather of the per-rank records on DEVICE tensors,
    destroy_process_group -- with libchanvese_hip.so loaded after torch and a 512^2 run (checked) in between.  What is
    left untested of the 8-GPU run is then only N > 1."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), CHANVESE_DIST_FORCE="1", OMP_NUM_THREADS="4")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("CHANVESE_DIST_BACKEND", "CHANVESE_BENCH_DRYRUN"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--size", "512", "--steps", "40",
                          "--warmup", "24", "--no-phases", "--no-cpu-baseline", "--prewarm-ms", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["backend"] == "nccl" and d["config"]["ranks_in_group"] == 1 and d["n_gpus"] == 1
    assert d["checked"] is True and d["value"] > 0
    assert d["roofline"]["kernel"].startswith("csv_resident_kernel<")     # 512 x 512 fits the LDS of the chip: the resident flow


def test_kernel_flavours_agree_at_4096(capi):
    """Every CSV data flow / arithmetic flavour against the 1-pixel wave kernel at 4096^2, 3 iterations, GPU vs GPU.
    Regression test for a gfx950 hazard found in round 2 that only showed under memory back-pressure (>= 1024^2) and only in
    the flavours whose register allocation exposed it (csv_wave2_kernel.hip keep[]): a 16-byte buffer store whose data
    registers were re-used by an LDS read right behind it stored stale values in lanes 12-15 of every row of 16."""
    n = 4096
    planes = [synth.disk(n)]
    u0 = capi.checkerboard_host(n, n)

    def run(opts):
        with capi.Context(n, n, 1, capi.make_params(tol=0)) as ctx:
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_image(planes)
            ctx.set_levelset(u0)
            done, _ = ctx.run(3)
            assert done == 3
            return ctx.get_levelset()

    ref_fast, ref_strict = run(dict(kernel=2)), run(dict(kernel=2, math_mode=1))
    scale = np.abs(ref_fast).max()
    assert np.abs(ref_fast - ref_strict).max() <= 1e-9 * scale
    for opts, ref in ((dict(kernel=3), ref_fast), (dict(kernel=3, wave_pol=0), ref_fast), (dict(kernel=3, wave_pol=2), ref_fast),
                      (dict(kernel=3, near_switch=0), ref_fast), (dict(kernel=3, chain=0), ref_fast),
                      (dict(kernel=3, strip_rows=100, wave_cls=0), ref_fast), (dict(kernel=3, math_mode=1), ref_strict),
                      (dict(kernel=0), ref_fast), (dict(kernel=2, chain=0), ref_fast),
                      (dict(kernel=3, wave_sync=0), ref_fast), (dict(kernel=2, wave_sync=0), ref_fast)):
        d = np.abs(run(opts) - ref).max()
        assert d <= 1e-9 * scale, (opts, d)


def test_two_pixel_kernel_beyond_the_cache_policy_switch(capi):
    """4608^2: the footprint (361 MB) is above the 300 MB switch, so the 2-pixel kernel runs its plain-store flavour
    (csv_wave2_kernel<1, true, 3, 0, false>) by default -- compared with the 1-pixel kernel, GPU vs GPU, 3 iterations
    (the store-data hazard only showed under memory back-pressure, and every flavour has its own register allocation)."""
    n = 4608
    planes = [synth.disk(n)]
    u0 = capi.checkerboard_host(n, n)

    def run(opts):
        with capi.Context(n, n, 1, capi.make_params(tol=0)) as ctx:
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_image(planes)
            ctx.set_levelset(u0)
            info = ctx.launch_info()
            assert ctx.run(3)[0] == 3
            return ctx.get_levelset(), info

    ref, _ = run(dict(kernel=2))
    got, info = run({})
    assert info["kernel"] == "csv_wave2_kernel<1, true, 3, 0, false>", info
    assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max()


def test_three_channel_flavours_agree_at_4096(capi):
    """4096^2 x 3: the 2-pixel kernel with tables and with the quadratic region term against the 1-pixel kernel, GPU vs
    GPU, 3 iterations (production geometry, memory back-pressure: see test_kernel_flavours_agree_at_4096)."""
    n = 4096
    planes = synth.config_planes("C3", n)
    u0 = capi.checkerboard_host(n, n)

    def run(opts):
        with capi.Context(n, n, 3, capi.make_params(tol=0, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1])) as ctx:
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_image(planes)
            ctx.set_levelset(u0)
            assert ctx.run(3)[0] == 3
            return ctx.get_levelset()

    ref = run(dict(kernel=2))
    scale = np.abs(ref).max()
    # ("wave_sync": the workgroup barrier per group of rows -- a scheduling matter, off by default for three channels, on for one)
    for opts in (dict(kernel=3), dict(kernel=3, near_switch=0), dict(kernel=3, wave_pol=1), dict(kernel=3, chain=0), dict(kernel=3, wave_sync=1),
                 dict(kernel=2, wave_sync=1)):
        d = np.abs(run(opts) - ref).max()
        assert d <= 1e-9 * scale, (opts, d)


def test_pm_flavours_agree_at_2048(capi):
    """Perona-Malik data flows at 2048^2, STRICT arithmetic, 9 steps (odd: the 2-step kernel's last step runs the 1-step
    kernel): uint8 planes identical across the tile, wave, 2-step and resident-plane kernels."""
    n = 2048
    img = synth.config_planes("C4", n)
    outs = {}
    for pk in (0, 1, 3, 4):
        with capi.Context(n, n, 1) as ctx:
            ctx.set_option("math_mode", 1)
            ctx.set_option("pm_kernel", pk)
            ctx.set_image(img)
            ctx.perona_malik(30.0, 0.25, 2.25)
            outs[pk] = ctx.get_image()[0]
    for pk in (0, 3, 4):
        assert np.array_equal(outs[pk], outs[1]), pk
