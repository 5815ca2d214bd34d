"""CPU-side check of the boundary: the C-ABI library builds, loads and exports every
symbol include/chanvese_hip.h declares; without a GPU it refuses to create a context
(no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from chan_vese_amd import capi
    return capi


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(ROOT, "include", "chanvese_hip.h")).read()
    declared = set(re.findall(r"\b(cvh_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"cvh_last_error(NULL"}
    raw = ctypes.CDLL(built.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in chanvese_hip.h but not exported"
    assert declared == set(built.EXPORTS), declared ^ set(built.EXPORTS)


def test_header_is_plain_c(built, tmp_path):
    """The boundary is a C ABI: the header compiles as C99 and a C program links against the library by symbol."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text(
        '#include "chanvese_hip.h"\n#include <stdio.h>\n'
        "int main(void) { cvh_params p; cvh_default_params(&p); int n = -1; cvh_device_count(&n);\n"
        '  printf("%s %g %g %d\\n", cvh_version(), p.mu, p.dt, cvh_pm_trip_count(0.25, 20)); return n < 0; }\n')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(built.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", libdir, "-lchanvese_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "gfx950" in out.stdout and " 0.5 1 80" in out.stdout


def test_version_and_defaults(built):
    assert b"gfx950" in built.lib().cvh_version()
    p = built.make_params()
    assert (p.mu, p.nu, p.dt, p.eps, p.tol) == (0.5, 0.0, 1.0, 1.0, 0.001)   # src/main.cpp:759-765
    assert list(p.lambda1) == [1, 1, 1] and list(p.lambda2) == [1, 1, 1]


def test_pm_trip_count_host(built):
    assert [built.pm_trip_count(L, T) for L, T in [(.25, 20), (.25, 250), (.25, 100), (.1, 1.5), (.1, 1)]] \
        == [80, 1000, 400, 15, 11]


def test_no_cpu_fallback(built):
    if built.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(built.CvhError) as e:
        built.Context(16, 16)
    assert "no HIP device" in str(e.value)


def test_argument_validation_without_gpu(built):
    with pytest.raises(built.CvhError):
        built.Context(16, 16, channels=2)
    with pytest.raises(built.CvhError):
        built.Context(-1, 16)
