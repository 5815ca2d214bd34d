"""bin/chan_vese: the reference's command-line surface (src/main.cpp:752-874).  Validation
needs no GPU (it runs before the backend is touched); the end-to-end run is a -m gpu test."""
import os
import subprocess

import numpy as np
import pytest

from chan_vese_amd import synth

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
    include/ParallelPixelFunction.hpp; std::function callables are recognised and run on the GPU."""
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
    inp = f"{h} {w} {eps} 9\n" + "\n".join(repr(float(v)) for v in x.ravel()) + "\n"
    r = subprocess.run([exe], input=inp, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr       # an unknown callable is refused loudly
