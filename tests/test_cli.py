"""bin/chan_vese: the reference's command-line surface (src/main.cpp:752-874).  Validation
needs no GPU (it runs before the backend is touched); the end-to-end run is a -m gpu test."""
import os
import subprocess

import numpy as np
import pytest

from chan_vese_amd import synth

import png_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin", "chan_vese")


@pytest.fixture(scope="module")
def cli():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(BIN)
    return BIN


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n# synthetic\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, dtype=np.uint8).tobytes())


def write_ppm(path, rgb):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(np.ascontiguousarray(rgb, dtype=np.uint8).tobytes())


def read_pnm(path):
    data = open(path, "rb").read()
    parts = data.split(b"\n", 3)
    magic, dims, maxv, body = parts[0], parts[1], parts[2], parts[3]
    w, h = map(int, dims.split())
    c = 1 if magic == b"P5" else 3
    return np.frombuffer(body, dtype=np.uint8).reshape(h, w, c).squeeze()


def run(cli, *args):
    return subprocess.run([cli, *args], capture_output=True, text=True, timeout=600)


def test_help_lists_reference_options(cli):
    r = run(cli, "-h")
    assert r.returncode == 0
    for opt in ["--input", "--mu", "--nu", "--dt", "--lambda1", "--lambda2", "--epsilon", "--tolerance",
                "--max-steps", "--fps", "--overlay-pos", "--line-color", "--edge-coef", "--laplacian-coef",
                "--segment-time", "--segment", "--grayscale", "--video", "--overlay-text",
                "--invert-selection", "--select", "--rectangle", "--circle"]:
        assert opt in r.stdout


def test_validation_messages_match_reference(cli, tmp_path):
    img = tmp_path / "a.pgm"
    write_pgm(img, synth.disk(32))
    cases = [
        ([], "Error: you have to specify input file name!"),                                   # :792
        (["-i", str(tmp_path / "nope.pgm")], "does not exists!"),                               # :794
        (["-i", str(img), "--dt", "0"], "Cannot have negative or zero timestep: 0.000000."),   # :796
        (["-i", str(img), "--mu", "-1"], "Length penalty parameter cannot be negative: -1.000000."),  # :798
        (["-i", str(img), "-g", "--lambda1", "1", "2"], "Too many lambda1 values for a grayscale image."),  # :802
        (["-i", str(img), "--lambda1", "1"], "Number of lambda1 values must be 3 for a colored input image."),  # :804
        (["-i", str(img), "-g", "--lambda2", "1", "2"], "Too many lambda2 values for a grayscale image."),
        (["-i", str(img), "-P", "XX"], "Invalid text position requested."),                    # :842
        (["-i", str(img), "-l", "pink"], "Invalid contour color requested."),                  # :860
        (["-i", str(img), "-L", "0.3"], "The Laplacian coefficient in Perona-Malik segmentation must be between 0 and 0.25."),
        (["-i", str(img), "-L", "0.25", "-T", "0.1"], "The segmentation duration must exceed the value of Laplacian coefficient, 0.250000."),
        (["-i", str(img), "-R", "-C"], "Cannot initialize with both rectangular and circular contour"),   # :869
        (["-i", str(img), "--rect", "1,1,4,4", "--circ", "8,8,3"], "Cannot initialize with both rectangular and circular contour"),
        (["-i", str(img), "-R"], "use --rect x,y,w,h or --circ cx,cy,r"),
        (["-i", str(img), "--bogus"], "error: unrecognised option '--bogus'"),
        (["-i", str(img), "--mu"], "error: the required argument for option '--mu' is missing"),
        (["-i", str(img), "--mu", "abc"], "error: the argument ('abc') for option '--mu' is invalid"),
    ]
    for args, msg in cases:
        r = run(cli, *args)
        assert r.returncode == 1, (args, r.stderr)
        assert msg in r.stderr, (args, r.stderr)
        assert r.stderr.startswith("\n") and r.stderr.endswith("\n\n")    # msg_exit framing, :176


@pytest.mark.gpu
def test_cli_end_to_end_grayscale(cli, oracle, tmp_path):
    """README-style run: PM pre-smoothing + CSV + selection, compared with the oracle."""
    h, w = 96, 112
    img = synth.disk(96, 200, 50, noise=24, seed=7, h=h, w=w)
    path = tmp_path / "disk.pgm"
    write_pgm(path, img)
    r = run(cli, "-i", str(path), "-g", "-s", "-S", "-K", "30", "-L", "0.25", "-T", "5", "-N", "40",
            "--dump-u", str(tmp_path / "u.bin"), "--dump-mask", str(tmp_path / "m.pgm"), "--verbose")
    assert r.returncode == 0, r.stderr
    sm = oracle.perona_malik([img], 30, 0.25, 5)
    assert np.array_equal(read_pnm(tmp_path / "disk_pm.pgm"), sm[0])               # add_suffix(..., "pm")
    u_c, done_c, nrm_c, _ = oracle.csv_run(sm, oracle.checkerboard(h, w), oracle.make_params(), 40)
    u_g = np.fromfile(tmp_path / "u.bin", dtype=np.float64).reshape(h, w)
    assert f"{done_c} iterations" in r.stderr
    assert np.abs(u_g - u_c).max() / np.abs(u_c).max() <= 1e-6
    m = oracle.mask(u_c)
    assert np.array_equal(read_pnm(tmp_path / "m.pgm") // 255, m)
    sel = read_pnm(tmp_path / "disk_selection.pgm")                                 # 3-channel, white canvas
    img3 = np.repeat(img[:, :, None], 3, axis=2)
    assert np.array_equal(sel, oracle.separate(img3, u_c))


@pytest.mark.gpu
def test_cli_colour_lambda_order_is_bgr(cli, oracle, tmp_path):
    """--lambda1 a b c applies to B, G, R (cv::imread colour order, src/main.cpp:879,936)."""
    h, w = 64, 80
    r_, g_, b_ = synth.disk(64, 60, 200, h=h, w=w), synth.disk(64, 200, 60, h=h, w=w), synth.disk(64, 180, 40, h=h, w=w)
    path = tmp_path / "c.ppm"
    write_ppm(path, np.stack([r_, g_, b_], axis=2))
    r = run(cli, "-i", str(path), "-N", "12", "-t", "0", "--lambda1", "1", "1", "0.5", "--lambda2", "1", "0.5", "1",
            "--nu", "-0.5", "--dump-u", str(tmp_path / "u.bin"))
    assert r.returncode == 0, r.stderr
    p = oracle.make_params(tol=0, nu=-0.5, lambda1=[1, 1, 0.5], lambda2=[1, 0.5, 1])
    u_c, _, _, _ = oracle.csv_run([b_, g_, r_], oracle.checkerboard(h, w), p, 12)
    u_g = np.fromfile(tmp_path / "u.bin", dtype=np.float64).reshape(h, w)
    assert np.abs(u_g - u_c).max() / np.abs(u_c).max() <= 1e-9


@pytest.mark.gpu
def test_cpp_parallel_pixel_function_operator(cli, oracle):
    """tests/cpp/test_ppf.cpp: the reference's call site (src/main.cpp:988-989) compiled against
    include/ParallelPixelFunction.hpp; the three known std::function callables are recognised and run on the GPU, any other one
    runs on the host as in the reference."""
    exe = os.path.join(ROOT, "bin", "test_ppf")
    rng = np.random.default_rng(5)
    h, w, eps = 9, 13, 0.75
    x = rng.normal(scale=20, size=(h, w))
    for op in (0, 1, 2, 3):
        inp = f"{h} {w} {eps} {op}\n" + "\n".join(repr(float(v)) for v in x.ravel()) + "\n"
        r = subprocess.run([exe], input=inp, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        got = np.array([float(t) for t in r.stdout.split()]).reshape(h, w)
        want = x.copy()
        if op == 3:
            oracle.ppf_apply(want, 0, eps, 3, h * w - 2)
        else:
            oracle.ppf_apply(want, op, eps)
        assert np.allclose(got, want, rtol=1e-15, atol=3e-16)
    # any other callable: the reference's own semantics (src/ParallelPixelFunction.cpp:12-17), the caller's
    # function applied on the host over [start, end)
    for op, fn, lo, hi in ((9, np.sin, 0, h * w), (10, np.square, 2, h * w - 5)):
        inp = f"{h} {w} {eps} {op}\n" + "\n".join(repr(float(v)) for v in x.ravel()) + "\n"
        r = subprocess.run([exe], input=inp, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        got = np.array([float(t) for t in r.stdout.split()])
        want = x.ravel().copy()
        want[lo:hi] = fn(want[lo:hi])
        assert np.allclose(got, want, rtol=4e-16, atol=0)


def _passthrough(cli, path, *extra):
    """Zero Perona-Malik steps and zero CSV iterations: <stem>_pm<ext> is the decoded input (src/main.cpp:943-946)."""
    r = run(cli, "-i", str(path), "-S", "-L", "0", "-T", "0", "-N", "0", *extra)
    assert r.returncode == 0, r.stderr
    stem, ext = os.path.splitext(str(path))
    data = open(stem + "_pm" + ext, "rb").read()
    return png_util.decode8(data) if ext == ".png" else read_pnm(stem + "_pm" + ext)


@pytest.mark.gpu
def test_cli_png_decoding_matches_imread_rules(cli, tmp_path):
    """PNG input (png_io.hpp): every colour type / bit depth / filter type, alpha dropped, 16 bit -> high byte,
    low depths expanded, palette looked up; output PNG written by the CLI decodes with an independent decoder."""
    rng = np.random.default_rng(11)
    h, w = 13, 17
    flt = [0, 1, 2, 3, 4, 4, 3, 2, 1]
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (h, w), dtype=np.uint8)
    cases = []
    cases.append(("rgb8", png_util.encode(png_util.pack_samples(rgb.reshape(h, -1), 8), w, h, 8, 2, flt, idat_split=3), rgb, False))
    rgba = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)
    cases.append(("rgba8", png_util.encode(png_util.pack_samples(rgba.reshape(h, -1), 8), w, h, 8, 6, flt), rgb, False))
    cases.append(("gray8", png_util.encode(png_util.pack_samples(gray, 8), w, h, 8, 0, flt), gray, True))
    ga = np.stack([gray, 255 - gray], axis=2)
    cases.append(("graya8", png_util.encode(png_util.pack_samples(ga.reshape(h, -1), 8), w, h, 8, 4, flt), gray, True))
    g16 = rng.integers(0, 65536, (h, w))
    cases.append(("gray16", png_util.encode(png_util.pack_samples(g16, 16), w, h, 16, 0, flt), (g16 >> 8).astype(np.uint8), True))
    c16 = rng.integers(0, 65536, (h, w, 3))
    cases.append(("rgb16", png_util.encode(png_util.pack_samples(c16.reshape(h, -1), 16), w, h, 16, 2, flt), (c16 >> 8).astype(np.uint8), False))
    for d in (1, 2, 4):
        gl = rng.integers(0, 1 << d, (h, w))
        cases.append((f"gray{d}", png_util.encode(png_util.pack_samples(gl, d), w, h, d, 0, [0, 1, 2]), (gl * 255 // ((1 << d) - 1)).astype(np.uint8), True))
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    for d in (4, 8):
        idx = rng.integers(0, 16, (h, w))
        cases.append((f"pal{d}", png_util.encode(png_util.pack_samples(idx, d), w, h, d, 3, [0, 2], palette=pal), pal[idx], False))
    for name, data, expect, is_gray in cases:
        path = tmp_path / (name + ".png")
        path.write_bytes(data)
        got = _passthrough(cli, path, *(["-g"] if is_gray else []))
        assert np.array_equal(got, expect), name
        if is_gray:  # a gray file read as colour: three equal planes
            got3 = _passthrough(cli, path)
            assert np.array_equal(got3, np.repeat(expect[:, :, None], 3, axis=2)), name
    # interlaced files are refused like any undecodable input
    bad = bytearray(cases[0][1]); bad[28] = 1                       # IHDR interlace byte (CRC now wrong as well)
    (tmp_path / "bad.png").write_bytes(bytes(bad))
    r = run(cli, "-i", str(tmp_path / "bad.png"))
    assert r.returncode == 1 and "probably not an image" in r.stderr


@pytest.mark.gpu
def test_cli_grayscale_conversion_rule_depends_on_decoder(cli, tmp_path):
    """-g on a colour file: PxM goes through the 14-bit BT.601 fixed point (R 4899, G 9617, B 1868), PNG through
    libpng's rgb_to_gray (15 bit: R 9798, G 19235, B 3735; equal channels pass through)."""
    rng = np.random.default_rng(12)
    h, w = 16, 21
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rgb[0, :4] = [[7, 7, 7], [255, 255, 255], [255, 0, 0], [0, 0, 255]]
    r_, g_, b_ = (rgb[:, :, k].astype(np.int64) for k in range(3))
    write_ppm(tmp_path / "c.ppm", rgb)
    assert np.array_equal(_passthrough(cli, tmp_path / "c.ppm", "-g"), ((r_ * 4899 + g_ * 9617 + b_ * 1868 + 8192) >> 14).astype(np.uint8))
    (tmp_path / "c.png").write_bytes(png_util.encode(png_util.pack_samples(rgb.reshape(h, -1), 8), w, h, 8, 2, [1, 4]))
    y = ((r_ * 9798 + g_ * 19235 + b_ * 3735 + 16384) >> 15).astype(np.uint8)
    eq = (r_ == g_) & (g_ == b_)
    y[eq] = r_[eq]
    assert np.array_equal(_passthrough(cli, tmp_path / "c.png", "-g"), y)


@pytest.mark.gpu
def test_cli_png_and_pgm_inputs_give_the_same_run(cli, tmp_path):
    h, w = 48, 64
    img = synth.disk(48, 190, 60, noise=10, seed=3, h=h, w=w)
    write_pgm(tmp_path / "a.pgm", img)
    (tmp_path / "b.png").write_bytes(png_util.encode(png_util.pack_samples(img, 8), w, h, 8, 0, [4, 1, 3]))
    us = []
    for name in ("a.pgm", "b.png"):
        r = run(cli, "-i", str(tmp_path / name), "-g", "-s", "-N", "15", "--dump-u", str(tmp_path / (name + ".u")), "--dump-mask", str(tmp_path / (name + ".m.png")))
        assert r.returncode == 0, r.stderr
        us.append(np.fromfile(tmp_path / (name + ".u"), dtype=np.float64))
    assert np.array_equal(us[0], us[1])
    sel_png = png_util.decode8(open(tmp_path / "b_selection.png", "rb").read())
    assert np.array_equal(sel_png, read_pnm(tmp_path / "a_selection.pgm"))
    assert np.array_equal(png_util.decode8(open(tmp_path / "a.pgm.m.png", "rb").read()), png_util.decode8(open(tmp_path / "b.png.m.png", "rb").read()))


def _midpoint_circle(h, w, cx, cy, r):
    """cv::circle(u, centre, r, 1): OpenCV's integer midpoint circle (restated independently of main.cpp)."""
    u = np.zeros((h, w))
    err, dx, dy, plus, minus = 0, r, 0, 1, 2 * r - 1
    while dx >= dy:
        for x, y in [(cx - dx, cy - dy), (cx + dx, cy - dy), (cx - dx, cy + dy), (cx + dx, cy + dy),
                     (cx - dy, cy - dx), (cx + dy, cy - dx), (cx - dy, cy + dx), (cx + dy, cy + dx)]:
            if 0 <= x < w and 0 <= y < h:
                u[y, x] = 1
        dy += 1
        err += plus
        plus += 2
        if err > 0:
            err -= minus
            dx -= 1
            minus -= 2
    return u


@pytest.mark.gpu
def test_cli_circle_initial_contour(cli, oracle, tmp_path):
    """--circ cx,cy,r: zeros with a 1-pixel circle outline of ones (src/InteractiveDataCirc.cpp:18-25), clipped."""
    h, w = 40, 56
    img = synth.disk(40, 200, 50, h=h, w=w)
    write_pgm(tmp_path / "a.pgm", img)
    for (cx, cy, r) in [(28, 20, 12), (5, 35, 9), (28, 20, 1)]:
        rr = run(cli, "-i", str(tmp_path / "a.pgm"), "-g", "--circ", f"{cx},{cy},{r}", "-N", "0", "--dump-u", str(tmp_path / "u0.bin"))
        assert rr.returncode == 0, rr.stderr
        u0 = np.fromfile(tmp_path / "u0.bin", dtype=np.float64).reshape(h, w)
        assert np.array_equal(u0, _midpoint_circle(h, w, cx, cy, r))
        yy, xx = np.nonzero(u0)
        assert np.all(np.abs(np.hypot(xx - cx, yy - cy) - r) < 1.0)      # an outline, not a disc
    rr = run(cli, "-i", str(tmp_path / "a.pgm"), "-g", "--circ", "28,20,12", "-N", "8", "-t", "0", "--dump-u", str(tmp_path / "u8.bin"))
    assert rr.returncode == 0, rr.stderr
    u_c, _, _, _ = oracle.csv_run([img], _midpoint_circle(h, w, 28, 20, 12), oracle.make_params(tol=0), 8)
    u_g = np.fromfile(tmp_path / "u8.bin", dtype=np.float64).reshape(h, w)
    assert np.abs(u_g - u_c).max() / np.abs(u_c).max() <= 1e-6
    assert run(cli, "-i", str(tmp_path / "a.pgm"), "-g", "--circ", "3,3,0").returncode == 1


@pytest.mark.gpu
def test_cli_video_frames(cli, oracle, tmp_path):
    """-V: one frame for t = 0 and one after every iteration (src/main.cpp:926-931,997), each the input with the
    contour of VideoWriterManager::draw_contour in --line-color; the last frame is written before the stop test."""
    h, w = 40, 48
    img = synth.disk(40, 200, 50, noise=6, seed=5, h=h, w=w)
    write_pgm(tmp_path / "v.pgm", img)
    r = run(cli, "-i", str(tmp_path / "v.pgm"), "-g", "-V", "-l", "yellow", "-N", "3", "-t", "0")
    assert r.returncode == 0, r.stderr
    frames = sorted(os.listdir(tmp_path / "v_frames"))
    assert frames == [f"frame_{k:06d}.ppm" for k in range(4)]
    u = oracle.checkerboard(h, w)
    for k in range(4):
        if k:
            u, _, _, _ = oracle.csv_run([img], oracle.checkerboard(h, w), oracle.make_params(tol=0), k)
        c = oracle.video_contour(u).astype(bool)
        expect = np.repeat(img[:, :, None], 3, axis=2)
        expect[c] = [255, 255, 0]
        assert np.array_equal(read_pnm(tmp_path / "v_frames" / frames[k]), expect), k
    # -O: "t = <iteration>" at the chosen corner (padding 5), black over a bright background, white over a dark one
    # (src/VideoWriterManager.cpp:76-114); own 5x7 glyphs: the text box is 6 len - 1 by 7 pixels; nothing else changes
    for pos, bright in (("TL", False), ("BR", True), ("TR", True), ("BL", False)):
        im2 = np.full((h, w), 230 if bright else 40, dtype=np.uint8)
        im2[15:25, 10:38] = 40 if bright else 230
        write_pgm(tmp_path / f"o{pos}.pgm", im2)
        plain = run(cli, "-i", str(tmp_path / f"o{pos}.pgm"), "-g", "-V", "-N", "1", "-t", "0")
        assert plain.returncode == 0, plain.stderr
        f0 = [read_pnm(tmp_path / f"o{pos}_frames" / f"frame_{k:06d}.ppm") for k in range(2)]
        withtxt = run(cli, "-i", str(tmp_path / f"o{pos}.pgm"), "-g", "-V", "-O", "-P", pos.lower(), "-N", "1", "-t", "0")
        assert withtxt.returncode == 0, withtxt.stderr
        for k in range(2):
            f1 = read_pnm(tmp_path / f"o{pos}_frames" / f"frame_{k:06d}.ppm")
            tw, th = 6 * len(f"t = {k}") - 1, 7
            x0 = 5 if pos[1] == "L" else w - 5 - tw
            y0 = 5 if pos[0] == "T" else h - 5 - th
            diff = np.any(f1 != f0[k], axis=2)
            assert diff.any() and not diff[:y0].any() and not diff[y0 + th:].any() and not diff[:, :x0].any() and not diff[:, x0 + tw:].any()
            assert np.all(f1[diff] == (0 if bright else 255))        # 255 - mean < 105 -> black, else white
            assert 8 <= diff.sum() <= tw * th // 2
    # default tolerance: the loop breaks at the reference's iteration and that iteration's frame exists
    write_pgm(tmp_path / "v2.pgm", img)
    r = run(cli, "-i", str(tmp_path / "v2.pgm"), "-g", "-V", "-t", "0.5", "--verbose")
    assert r.returncode == 0, r.stderr
    _, done, _, _ = oracle.csv_run([img], oracle.checkerboard(h, w), oracle.make_params(tol=0.5), 10 ** 6)
    assert f"{done} iterations" in r.stderr
    assert len(os.listdir(tmp_path / "v2_frames")) == done + 1


@pytest.mark.gpu
def test_cli_state_32_is_a_declared_switch(cli, oracle, tmp_path):
    """--state 32 (build-only addition): the declared FP32-state mode of the C ABI through the CLI -- same outputs as the default run up to the
    survey's statistical bar (mask equal, median |du| / max|u| <= 1e-4); widths the 2-pixel kernel cannot take and other values are refused."""
    h, w = 96, 160
    img = synth.disk(96, 200, 50, noise=10, seed=9, h=h, w=w)
    write_pgm(tmp_path / "a.pgm", img)
    us = {}
    for st in ("64", "32"):
        r = run(cli, "-i", str(tmp_path / "a.pgm"), "-g", "-N", "40", "-t", "0", "--state", st, "--dump-u", str(tmp_path / f"u{st}.bin"),
                "--dump-mask", str(tmp_path / f"m{st}.pgm"))
        assert r.returncode == 0, r.stderr
        us[st] = np.fromfile(tmp_path / f"u{st}.bin", dtype=np.float64).reshape(h, w)
    u_c, _, _, _ = oracle.csv_run([img], oracle.checkerboard(h, w), oracle.make_params(tol=0), 40)
    assert np.abs(us["64"] - u_c).max() <= 1e-6 * np.abs(u_c).max()
    assert np.array_equal(us["32"], us["32"].astype(np.float32).astype(np.float64))            # floats
    assert np.median(np.abs(us["32"] - u_c)) <= 1e-4 * np.abs(u_c).max()
    assert np.array_equal(read_pnm(tmp_path / "m32.pgm"), read_pnm(tmp_path / "m64.pgm"))
    write_pgm(tmp_path / "b.pgm", synth.disk(64, 200, 50, h=64, w=100))
    assert run(cli, "-i", str(tmp_path / "b.pgm"), "-g", "-N", "2", "--state", "32").returncode == 1      # width not a multiple of 16
    r = run(cli, "-i", str(tmp_path / "a.pgm"), "-g", "--state", "16")
    assert r.returncode == 1 and "option '--state' is invalid" in r.stderr
