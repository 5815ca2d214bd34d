"""The two example runs of the reference's README (/root/reference/README.md:51-54 and :58-63) as parity cases, through the C ABI stage
by stage and through bin/chan_vese end to end.  The example images are not part of the reference (and there is no network): synthetic
stand-ins of the same geometry and character (chan_vese_amd/synth.py: sea_star 370 x 278, night_lights 640 x 480; three channels, B G R).

  A  bin/chan_vese -i seastar.png -s -N 70 -S -L 0.25 -T 100 -K 30                       (400 Perona-Malik steps, 70 iterations)
  B  bin/chan_vese -i 640px-Europe_night.png -N 132 --dt 0.001 -t 0.000001 --nu -293 --lambda1 1 1 0.1
                   -S -L 0.1 -T 1.5 -K 1000 -s -V -f 12 -l red                              (15 steps, 132 iterations)

Example B is the regime no BASELINE config reaches: with dt = 0.001 the level set stays within |u| < 32 eps for the whole run, so EVERY
pixel takes the near-field (table) form of H_eps in every iteration (bench.py --config near measures it).

Tolerances: Perona-Malik planes STRICT equal / FAST <= 1 LSB on <= 1e-6 of the pixels; level set <= 1e-9 of max|u| over the first 10
iterations and <= 1e-6 at the end; c1/c2/norm of every iteration <= 1e-9 (first 10) / 1e-6; identical stop iteration; mask equal.
PARITY UNPINNED: the expected values come from this repository's oracle (oracle/cv_oracle.c)."""
import os
import subprocess

import numpy as np
import pytest

from chan_vese_amd import synth
from test_cli import BIN, read_pnm, write_pgm, write_ppm

pytestmark = pytest.mark.gpu

EXAMPLES = {
    # name: (planes, gray, PM (K, L, T), N, CSV parameters, the README's flags)
    "A_colour": (lambda: synth.sea_star(), False, (30.0, 0.25, 100.0), 70, dict(),
                 ["-s", "-N", "70", "-S", "-L", "0.25", "-T", "100", "-K", "30"]),
    "A_gray": (lambda: [synth.sea_star()[2]], True, (30.0, 0.25, 100.0), 70, dict(),
               ["-g", "-s", "-N", "70", "-S", "-L", "0.25", "-T", "100", "-K", "30"]),
    "B": (lambda: synth.night_lights(), False, (1000.0, 0.1, 1.5), 132, dict(dt=0.001, tol=1e-6, nu=-293.0, lambda1=[1, 1, 0.1]),
          ["-N", "132", "--dt", "0.001", "-t", "0.000001", "--nu", "-293", "--lambda1", "1", "1", "0.1",
           "-S", "-L", "0.1", "-T", "1.5", "-K", "1000", "-s", "-V", "-f", "12", "-l", "red"]),
}
_LEGS = {}


@pytest.fixture(scope="module")
def capi():
    from chan_vese_amd import capi as m
    m.lib()
    assert m.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return m


def oracle_leg(oracle, name):
    if name not in _LEGS:
        make, gray, (K, L, T), N, pk, _ = EXAMPLES[name]
        planes = make()
        h, w = planes[0].shape
        pm = oracle.perona_malik(planes, K, L, T)
        p = oracle.make_params(**pk)
        u0 = oracle.checkerboard(h, w)
        u10, d10, _, tr10 = oracle.csv_run(pm, u0, p, 10)
        u, done, nrm, tr = oracle.csv_run(pm, u0, p, N)
        _LEGS[name] = dict(planes=planes, pm=pm, u10=u10, u=u, done=done, nrm=nrm, tr=tr, stop=oracle.stop_condition(pm, pk.get("tol", 1e-3)))
    return _LEGS[name]


def rel_err(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("mode,math", [("fast", 2), ("strict", 1)])
@pytest.mark.parametrize("name", sorted(EXAMPLES))
def test_readme_example_through_the_c_abi(capi, oracle, name, mode, math):
    _, gray, (K, L, T), N, pk, _ = EXAMPLES[name]
    leg = oracle_leg(oracle, name)
    planes, pm_c = leg["planes"], leg["pm"]
    h, w = planes[0].shape
    with capi.Context(h, w, len(planes), capi.make_params(**pk)) as ctx:
        ctx.set_option("math_mode", math)
        ctx.set_image(planes)
        ctx.perona_malik(K, L, T)                                   # src/main.cpp:940-947
        for g, c in zip(ctx.get_image(), pm_c):
            diff = g.astype(int) - c.astype(int)
            if mode == "strict":
                assert not diff.any()
            else:
                assert np.abs(diff).max() <= 1 and (diff != 0).mean() <= 1e-6
        ctx.set_image(pm_c)       # identical input for the CSV part even if a boundary pixel rounded the other way
        ctx.set_option("trace", N)
        ctx.init_checkerboard()
        assert abs(ctx.get_stop_condition() - leg["stop"]) <= 1e-12 * leg["stop"]     # src/main.cpp:950-959
        done, _ = ctx.run(10)
        assert done == 10
        assert rel_err(ctx.get_levelset(), leg["u10"]) <= 1e-9
        assert np.allclose(ctx.get_trace(10), leg["tr"][:10], rtol=1e-9, atol=0)
        more, nrm = ctx.run(N - 10)                                  # continues (cvh_run counts -- and traces -- the iterations of THIS call): src/main.cpp:963 up to -N
        done = 10 + more
        u_g, tr_g, m_g = ctx.get_levelset(), ctx.get_trace(more), ctx.get_mask()
        assert done == leg["done"]                                   # the stop test (:1000) fires at the same iteration, or never
    assert np.allclose(tr_g, leg["tr"][10:done], rtol=1e-6, atol=0)
    assert rel_err(u_g, leg["u"]) <= 1e-6, rel_err(u_g, leg["u"])
    assert np.array_equal(m_g, oracle.mask(leg["u"]))
    if name == "B":     # the regime this example is here for: every pixel in the near field of H_eps for the whole run
        assert np.abs(leg["u"]).max() < 32.0 and 0.05 < oracle.mask(leg["u"]).mean() < 0.95


@pytest.mark.parametrize("name", sorted(EXAMPLES))
def test_readme_example_through_the_cli(oracle, tmp_path, name):
    """The README's command line itself (plus the build's --dump-u / --dump-mask): <stem>_pm, <stem>_selection, the level set, and for
    example B the frames of -V (input + red contour; one per iteration and one for t = 0)."""
    import __graft_entry__ as g
    g.build()
    _, gray, (K, L, T), N, pk, flags = EXAMPLES[name]
    leg = oracle_leg(oracle, name)
    planes, pm_c = leg["planes"], leg["pm"]
    h, w = planes[0].shape
    ext = "pgm" if gray else "ppm"
    path = tmp_path / f"in.{ext}"
    if gray:
        write_pgm(path, planes[0])
    else:
        write_ppm(path, np.stack(planes[::-1], axis=2))               # file order R G B; cv::imread gives B G R (:879)
    r = subprocess.run([BIN, "-i", str(path), *flags, "--dump-u", str(tmp_path / "u.bin"), "--dump-mask", str(tmp_path / "m.pgm"), "--verbose"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    pm_file = read_pnm(tmp_path / f"in_pm.{ext}")
    pm_want = pm_c[0] if gray else np.stack(pm_c[::-1], axis=2)
    diff = pm_file.astype(int) - pm_want.astype(int)
    assert np.abs(diff).max() <= 1 and (diff != 0).mean() <= 1e-6      # the CLI runs FAST arithmetic
    assert f"{leg['done']} iterations" in r.stderr
    u_g = np.fromfile(tmp_path / "u.bin", dtype=np.float64).reshape(h, w)
    if not diff.any():
        assert rel_err(u_g, leg["u"]) <= 1e-6
        assert np.array_equal(read_pnm(tmp_path / "m.pgm") // 255, oracle.mask(leg["u"]))
        img3 = np.repeat(planes[0][:, :, None], 3, axis=2) if gray else np.stack(planes, axis=2)      # separate() works on the ORIGINAL image (:1004), B G R
        sel = read_pnm(tmp_path / f"in_selection.{ext}")
        want = oracle.separate(img3, leg["u"])
        assert np.array_equal(sel, want if gray else want[:, :, ::-1])
    if "-V" in flags:
        frames = sorted(os.listdir(tmp_path / "in_frames"))
        assert len(frames) == leg["done"] + 1
        last = read_pnm(tmp_path / "in_frames" / frames[-1])
        if not diff.any():
            c = oracle.video_contour(leg["u"]).astype(bool)
            expect = np.stack(planes[::-1], axis=2).copy()             # the frame shows the input image, R G B in the file
            expect[c] = [255, 0, 0]
            assert np.array_equal(last, expect)
