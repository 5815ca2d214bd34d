"""ctypes wrapper over oracle/libcv_oracle.so — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see cv_oracle.h): the reference ships no golden vectors and cannot be
built here, so this oracle is a reading of /root/reference/src/main.cpp pinned by
hand-derived known-answer tests and an independent numpy restatement.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Params(C.Structure):
    _fields_ = [("mu", C.c_double), ("nu", C.c_double), ("dt", C.c_double),
                ("eps", C.c_double), ("tol", C.c_double),
                ("lambda1", C.c_double * 3), ("lambda2", C.c_double * 3)]


def make_params(mu=0.5, nu=0.0, dt=1.0, eps=1.0, tol=1e-3, lambda1=None, lambda2=None):
    p = Params()
    p.mu, p.nu, p.dt, p.eps, p.tol = mu, nu, dt, eps, tol
    l1 = list(lambda1) if lambda1 is not None else [1.0, 1.0, 1.0]
    l2 = list(lambda2) if lambda2 is not None else [1.0, 1.0, 1.0]
    for k in range(3):
        p.lambda1[k] = l1[k] if k < len(l1) else 1.0
        p.lambda2[k] = l2[k] if k < len(l2) else 1.0
    return p


def build(force=False):
    so = os.path.join(_HERE, "libcv_oracle.so")
    src = os.path.join(_HERE, "cv_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        u8p = C.POINTER(C.c_uint8)
        u8pp = C.POINTER(u8p)
        L.cvo_regularized_heaviside.restype = C.c_double
        L.cvo_regularized_heaviside.argtypes = [C.c_double, C.c_double]
        L.cvo_regularized_delta.restype = C.c_double
        L.cvo_regularized_delta.argtypes = [C.c_double, C.c_double]
        L.cvo_levelset_checkerboard.argtypes = [C.c_int, C.c_int, dp]
        L.cvo_levelset_rect.argtypes = [C.c_int] * 6 + [dp]
        L.cvo_region_variance.restype = C.c_double
        L.cvo_region_variance.argtypes = [u8p, dp, C.c_int, C.c_int, C.c_int, C.c_double]
        L.cvo_variance_penalty.argtypes = [u8p, C.c_int, C.c_int, C.c_double, C.c_double, dp]
        L.cvo_curvature.argtypes = [dp, C.c_int, C.c_int, dp]
        L.cvo_ppf_apply.argtypes = [dp, C.c_int, C.c_long, C.c_long, C.c_int, C.c_double]
        L.cvo_stop_condition.restype = C.c_double
        L.cvo_stop_condition.argtypes = [u8pp, C.c_int, C.c_int, C.c_int, C.c_double]
        L.cvo_csv_step.restype = C.c_double
        L.cvo_csv_step.argtypes = [u8pp, C.c_int, C.c_int, C.c_int, C.POINTER(Params), dp, dp, dp]
        L.cvo_csv_step_exact.restype = C.c_double
        L.cvo_csv_step_exact.argtypes = L.cvo_csv_step.argtypes
        L.cvo_region_means_exact.argtypes = [u8p, dp, C.c_int, C.c_int, C.c_double, dp, dp]
        L.cvo_csv_run.restype = C.c_int
        L.cvo_csv_run.argtypes = [u8pp, C.c_int, C.c_int, C.c_int, C.POINTER(Params), C.c_int,
                                  dp, dp, dp, C.c_int]
        L.cvo_pm_trip_count.restype = C.c_int
        L.cvo_pm_trip_count.argtypes = [C.c_double, C.c_double]
        L.cvo_perona_malik_channel.argtypes = [u8p, C.c_int, C.c_int, C.c_double, C.c_double,
                                               C.c_double, u8p, dp]
        L.cvo_mask.argtypes = [dp, C.c_int, C.c_int, C.c_int, u8p]
        L.cvo_video_contour.argtypes = [dp, C.c_int, C.c_int, u8p]
        L.cvo_separate.argtypes = [u8p, dp, C.c_int, C.c_int, C.c_int, u8p]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _planes(planes):
    planes = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
    arr = (C.POINTER(C.c_uint8) * len(planes))(*[_u8p(p) for p in planes])
    return planes, arr


def heaviside(x, eps=1.0):
    return lib().cvo_regularized_heaviside(float(x), float(eps))


def delta(x, eps=1.0):
    return lib().cvo_regularized_delta(float(x), float(eps))


def checkerboard(h, w):
    u = np.empty((h, w), dtype=np.float64)
    lib().cvo_levelset_checkerboard(h, w, _dp(u))
    return u


def levelset_rect(h, w, x, y, rw, rh):
    u = np.empty((h, w), dtype=np.float64)
    lib().cvo_levelset_rect(h, w, x, y, rw, rh, _dp(u))
    return u


def region_mean(img, u, region, eps=1.0):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    u = np.ascontiguousarray(u, dtype=np.float64)
    h, w = u.shape
    return lib().cvo_region_variance(_u8p(img), _dp(u), h, w, int(region), float(eps))


def variance_penalty(channel, c, lam):
    channel = np.ascontiguousarray(channel, dtype=np.uint8)
    h, w = channel.shape
    out = np.empty((h, w), dtype=np.float64)
    lib().cvo_variance_penalty(_u8p(channel), h, w, float(c), float(lam), _dp(out))
    return out


def curvature(u):
    u = np.ascontiguousarray(u, dtype=np.float64)
    h, w = u.shape
    k = np.empty((h, w), dtype=np.float64)
    lib().cvo_curvature(_dp(u), h, w, _dp(k))
    return k


def ppf_apply(data, op, eps=1.0, start=0, end=None):
    """In place on a 2-D float64 array, flat range [start, end)."""
    assert data.dtype == np.float64 and data.flags.c_contiguous and data.ndim == 2
    end = data.size if end is None else end
    lib().cvo_ppf_apply(_dp(data), data.shape[1], start, end, int(op), float(eps))
    return data


def stop_condition(planes, tol):
    planes, arr = _planes(planes)
    h, w = planes[0].shape
    return lib().cvo_stop_condition(arr, len(planes), h, w, float(tol))


def csv_step(planes, u, params):
    planes, arr = _planes(planes)
    h, w = planes[0].shape
    assert u.dtype == np.float64 and u.flags.c_contiguous
    c1 = np.zeros(3)
    c2 = np.zeros(3)
    nrm = lib().cvo_csv_step(arr, len(planes), h, w, C.byref(params), _dp(u), _dp(c1), _dp(c2))
    return nrm, c1[:len(planes)].copy(), c2[:len(planes)].copy()


def csv_step_exact(planes, u, params):
    """csv_step with the region means from compensated long double sums (the exact-sum adjudicator)."""
    planes, arr = _planes(planes)
    h, w = planes[0].shape
    assert u.dtype == np.float64 and u.flags.c_contiguous
    c1 = np.zeros(3)
    c2 = np.zeros(3)
    nrm = lib().cvo_csv_step_exact(arr, len(planes), h, w, C.byref(params), _dp(u), _dp(c1), _dp(c2))
    return nrm, c1[:len(planes)].copy(), c2[:len(planes)].copy()


def region_means_exact(planes, u, eps=1.0):
    """(c1[C], c2[C]) of the level set u: the reference's per-pixel terms (src/main.cpp:272-280) added without
    accumulation error (Neumaier-compensated long double)."""
    u = np.ascontiguousarray(u, dtype=np.float64)
    h, w = u.shape
    c1, c2 = [], []
    for pl in planes:
        pl = np.ascontiguousarray(pl, dtype=np.uint8)
        a, b = C.c_double(0.0), C.c_double(0.0)
        lib().cvo_region_means_exact(_u8p(pl), _dp(u), h, w, float(eps), C.byref(a), C.byref(b))
        c1.append(a.value)
        c2.append(b.value)
    return np.array(c1), np.array(c2)


def region_means(planes, u, eps=1.0):
    """(c1[C], c2[C]) as the reference computes them: sequential double sums, one sweep per mean."""
    return (np.array([region_mean(pl, u, 0, eps) for pl in planes]),
            np.array([region_mean(pl, u, 1, eps) for pl in planes]))


def csv_run(planes, u0, params, max_steps, trace=True):
    """Returns (u, steps_done, last_norm, trace[steps, 2C+1])."""
    planes, arr = _planes(planes)
    h, w = planes[0].shape
    u = np.array(u0, dtype=np.float64, order="C", copy=True)
    nc = len(planes)
    cap = max_steps if trace else 0
    tr = np.zeros((max(cap, 1), 2 * nc + 1), dtype=np.float64)
    last = C.c_double(0.0)
    done = lib().cvo_csv_run(arr, nc, h, w, C.byref(params), int(max_steps), _dp(u),
                             C.byref(last), _dp(tr) if trace else None, cap)
    return u, done, last.value, tr[:done] if trace else None


def pm_trip_count(L, T):
    return lib().cvo_pm_trip_count(float(L), float(T))


def perona_malik(planes, K, L, T, want_state=False):
    outs, states = [], []
    for p in planes:
        p = np.ascontiguousarray(p, dtype=np.uint8)
        h, w = p.shape
        o = np.empty((h, w), dtype=np.uint8)
        s = np.empty((h, w), dtype=np.float64)
        lib().cvo_perona_malik_channel(_u8p(p), h, w, float(K), float(L), float(T), _u8p(o), _dp(s))
        outs.append(o)
        states.append(s)
    return (outs, states) if want_state else outs


def mask(u, invert=False):
    u = np.ascontiguousarray(u, dtype=np.float64)
    h, w = u.shape
    m = np.empty((h, w), dtype=np.uint8)
    lib().cvo_mask(_dp(u), h, w, int(bool(invert)), _u8p(m))
    return m


def video_contour(u):
    u = np.ascontiguousarray(u, dtype=np.float64)
    h, w = u.shape
    m = np.empty((h, w), dtype=np.uint8)
    lib().cvo_video_contour(_dp(u), h, w, _u8p(m))
    return m


def separate(img3, u, invert=False):
    img3 = np.ascontiguousarray(img3, dtype=np.uint8)
    u = np.ascontiguousarray(u, dtype=np.float64)
    h, w = u.shape
    out = np.empty((h, w, 3), dtype=np.uint8)
    lib().cvo_separate(_u8p(img3), _dp(u), h, w, int(bool(invert)), _u8p(out))
    return out
