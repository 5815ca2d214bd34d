"""ctypes binding of include/chanvese_hip.h (the C ABI of libchanvese_hip.so).

No CPU fallback: if the HIP library has not been built, importing/using this module raises
with instructions.  In a process that also imports torch, import torch FIRST: torch ships
its own libamdhip64.so.7 and the loader then shares that one runtime with this library.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHANVESE_HIP_LIB selects another BUILD of the same library (A/B variants under csrc/variants/, tools/build_variant.sh)
LIB_PATH = os.environ.get("CHANVESE_HIP_LIB") or os.path.join(_HERE, "csrc", "libchanvese_hip.so")

CVH_OK = 0
OP_DELTA, OP_HEAVISIDE, OP_ONE_MINUS_HEAVISIDE = 0, 1, 2
MATH_DEFAULT, MATH_STRICT, MATH_FAST = 0, 1, 2

# every symbol include/chanvese_hip.h declares
EXPORTS = [
    "cvh_default_params", "cvh_device_count", "cvh_create", "cvh_destroy", "cvh_last_error",
    "cvh_set_params", "cvh_set_option", "cvh_set_image", "cvh_get_image", "cvh_set_levelset",
    "cvh_get_levelset", "cvh_init_checkerboard", "cvh_levelset_checkerboard_host", "cvh_run",
    "cvh_enqueue_steps", "cvh_warm", "cvh_sync", "cvh_reset_run", "cvh_get_means", "cvh_get_trace",
    "cvh_get_stop_condition", "cvh_get_mask", "cvh_get_contour", "cvh_separate", "cvh_perona_malik",
    "cvh_pm_trip_count", "cvh_last_run_ms", "cvh_last_pm_ms", "cvh_ppf_apply",
    "cvh_ppf_apply_device", "cvh_version", "cvh_launch_info",
]


class Params(C.Structure):
    """struct cvh_params (src/main.cpp:731-734 of the reference)."""
    _fields_ = [("mu", C.c_double), ("nu", C.c_double), ("dt", C.c_double),
                ("eps", C.c_double), ("tol", C.c_double),
                ("lambda1", C.c_double * 3), ("lambda2", C.c_double * 3)]


class CvhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"chanvese_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C chan_vese_amd/csrc). "
            "There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    dp, u8p, ip = C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int)
    u8pp, vp = C.POINTER(u8p), C.c_void_p
    sig = {
        "cvh_default_params": (None, [C.POINTER(Params)]),
        "cvh_device_count": (C.c_int, [ip]),
        "cvh_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.POINTER(Params), C.c_int]),
        "cvh_destroy": (None, [vp]),
        "cvh_last_error": (C.c_char_p, [vp]),
        "cvh_set_params": (C.c_int, [vp, C.POINTER(Params)]),
        "cvh_set_option": (C.c_int, [vp, C.c_char_p, C.c_long]),
        "cvh_set_image": (C.c_int, [vp, u8pp]),
        "cvh_get_image": (C.c_int, [vp, u8pp]),
        "cvh_set_levelset": (C.c_int, [vp, dp]),
        "cvh_get_levelset": (C.c_int, [vp, dp]),
        "cvh_init_checkerboard": (C.c_int, [vp]),
        "cvh_levelset_checkerboard_host": (None, [C.c_int, C.c_int, dp]),
        "cvh_run": (C.c_int, [vp, C.c_int, ip, dp]),
        "cvh_enqueue_steps": (C.c_int, [vp, C.c_int]),
        "cvh_warm": (C.c_int, [vp, C.c_int]),
        "cvh_sync": (C.c_int, [vp, ip, dp, ip]),
        "cvh_reset_run": (C.c_int, [vp]),
        "cvh_get_means": (C.c_int, [vp, dp, dp]),
        "cvh_get_trace": (C.c_int, [vp, dp, C.c_int, ip]),
        "cvh_get_stop_condition": (C.c_int, [vp, dp]),
        "cvh_get_mask": (C.c_int, [vp, u8p, C.c_int]),
        "cvh_get_contour": (C.c_int, [vp, u8p]),
        "cvh_separate": (C.c_int, [vp, u8p, C.c_int, u8p]),
        "cvh_perona_malik": (C.c_int, [vp, C.c_double, C.c_double, C.c_double]),
        "cvh_pm_trip_count": (C.c_int, [C.c_double, C.c_double]),
        "cvh_last_run_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "cvh_last_pm_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "cvh_ppf_apply": (C.c_int, [dp, C.c_int, C.c_long, C.c_long, C.c_int, C.c_double, C.c_int]),
        "cvh_ppf_apply_device": (C.c_int, [dp, C.c_long, C.c_int, C.c_double, vp]),
        "cvh_version": (C.c_char_p, []),
        "cvh_launch_info": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def make_params(mu=0.5, nu=0.0, dt=1.0, eps=1.0, tol=1e-3, lambda1=None, lambda2=None):
    p = Params()
    lib().cvh_default_params(C.byref(p))
    p.mu, p.nu, p.dt, p.eps, p.tol = mu, nu, dt, eps, tol
    for name, vals in (("lambda1", lambda1), ("lambda2", lambda2)):
        if vals is not None:
            arr = getattr(p, name)
            for k, v in enumerate(vals):
                arr[k] = float(v)
    return p


def device_count():
    n = C.c_int(0)
    lib().cvh_device_count(C.byref(n))
    return n.value


def pm_trip_count(L, T):
    return lib().cvh_pm_trip_count(float(L), float(T))


def checkerboard_host(h, w):
    u = np.empty((h, w), dtype=np.float64)
    lib().cvh_levelset_checkerboard_host(h, w, _dp(u))
    return u


def ppf_apply(data, op, eps=1.0, start=0, end=None, device=0):
    """ParallelPixelFunction(data, w, f)(Range(start, end)) on the GPU, in place."""
    assert data.dtype == np.float64 and data.flags.c_contiguous and data.ndim == 2
    end = data.size if end is None else end
    rc = lib().cvh_ppf_apply(_dp(data), data.shape[1], start, end, int(op), float(eps), device)
    if rc != CVH_OK:
        raise CvhError(rc, lib().cvh_last_error(None).decode())
    return data


class Context:
    """One image on one GPU: thin RAII wrapper over cvh_context."""

    def __init__(self, h, w, channels=1, params=None, device=0):
        self._L = lib()
        self._h = C.c_void_p(None)
        self.h, self.w, self.channels = h, w, channels
        p = params if params is not None else make_params()
        rc = self._L.cvh_create(C.byref(self._h), h, w, channels, C.byref(p), device)
        if rc != CVH_OK:
            self._h = C.c_void_p(None)
            raise CvhError(rc, self._L.cvh_last_error(None).decode())

    def _chk(self, rc):
        if rc != CVH_OK:
            raise CvhError(rc, self._L.cvh_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._L.cvh_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_params(self, params):
        self._chk(self._L.cvh_set_params(self._h, C.byref(params)))

    def set_option(self, key, value):
        self._chk(self._L.cvh_set_option(self._h, key.encode(), int(value)))

    def _plane_array(self, planes):
        assert len(planes) == self.channels
        keep = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
        for p in keep:
            assert p.shape == (self.h, self.w)
        arr = (C.POINTER(C.c_uint8) * len(keep))(*[_u8p(p) for p in keep])
        return keep, arr

    def set_image(self, planes):
        keep, arr = self._plane_array(planes)
        self._chk(self._L.cvh_set_image(self._h, arr))

    def get_image(self):
        outs = [np.empty((self.h, self.w), dtype=np.uint8) for _ in range(self.channels)]
        arr = (C.POINTER(C.c_uint8) * len(outs))(*[_u8p(p) for p in outs])
        self._chk(self._L.cvh_get_image(self._h, arr))
        return outs

    def set_levelset(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.shape == (self.h, self.w)
        self._chk(self._L.cvh_set_levelset(self._h, _dp(u)))

    def init_checkerboard(self):
        self._chk(self._L.cvh_init_checkerboard(self._h))

    def get_levelset(self):
        u = np.empty((self.h, self.w), dtype=np.float64)
        self._chk(self._L.cvh_get_levelset(self._h, _dp(u)))
        return u

    def run(self, max_steps=-1):
        """Returns (steps_done, last_norm)."""
        done, nrm = C.c_int(0), C.c_double(0.0)
        self._chk(self._L.cvh_run(self._h, int(max_steps), C.byref(done), C.byref(nrm)))
        return done.value, nrm.value

    def enqueue_steps(self, n):
        self._chk(self._L.cvh_enqueue_steps(self._h, int(n)))

    def warm(self, n):
        """One-off host work of an upcoming enqueue_steps(n) (graph build), outside any timed region."""
        self._chk(self._L.cvh_warm(self._h, int(n)))

    def sync(self):
        """Returns (steps_done_total, last_norm, stopped)."""
        done, nrm, stopped = C.c_int(0), C.c_double(0.0), C.c_int(0)
        self._chk(self._L.cvh_sync(self._h, C.byref(done), C.byref(nrm), C.byref(stopped)))
        return done.value, nrm.value, bool(stopped.value)

    def reset_run(self):
        self._chk(self._L.cvh_reset_run(self._h))

    def get_means(self):
        c1, c2 = np.zeros(3), np.zeros(3)
        self._chk(self._L.cvh_get_means(self._h, _dp(c1), _dp(c2)))
        return c1[:self.channels].copy(), c2[:self.channels].copy()

    def get_trace(self, max_rows):
        out = np.zeros((max(max_rows, 1), 2 * self.channels + 1), dtype=np.float64)
        rows = C.c_int(0)
        self._chk(self._L.cvh_get_trace(self._h, _dp(out), int(max_rows), C.byref(rows)))
        return out[:rows.value].copy()

    def get_stop_condition(self):
        v = C.c_double(0.0)
        self._chk(self._L.cvh_get_stop_condition(self._h, C.byref(v)))
        return v.value

    def get_mask(self, invert=False):
        m = np.empty((self.h, self.w), dtype=np.uint8)
        self._chk(self._L.cvh_get_mask(self._h, _u8p(m), int(bool(invert))))
        return m

    def get_contour(self):
        m = np.empty((self.h, self.w), dtype=np.uint8)
        self._chk(self._L.cvh_get_contour(self._h, _u8p(m)))
        return m

    def separate(self, img3, invert=False):
        img3 = np.ascontiguousarray(img3, dtype=np.uint8)
        assert img3.shape == (self.h, self.w, 3)
        out = np.empty_like(img3)
        self._chk(self._L.cvh_separate(self._h, _u8p(img3), int(bool(invert)), _u8p(out)))
        return out

    def perona_malik(self, K=10.0, L=0.25, T=20.0):
        self._chk(self._L.cvh_perona_malik(self._h, float(K), float(L), float(T)))

    def last_run_ms(self):
        v = C.c_float(0.0)
        self._chk(self._L.cvh_last_run_ms(self._h, C.byref(v)))
        return v.value

    def launch_info(self, phase=0):
        """What the library launches (phase 0: the CSV step with the current options; 1: the last Perona-Malik call) as a
        dict of the key=value pairs cvh_launch_info writes; "kernel" is the instantiation as rocprofv3 prints it."""
        buf = C.create_string_buffer(512)
        self._chk(self._L.cvh_launch_info(self._h, int(phase), buf, len(buf)))
        out = {}
        for tok in buf.value.decode().split(" "):
            if "=" in tok:
                k, v = tok.split("=", 1)
                out[k] = v
            elif out:                       # template arguments are separated by ", "
                last = next(reversed(out))
                out[last] += " " + tok
        return out

    def last_pm_ms(self):
        v = C.c_float(0.0)
        self._chk(self._L.cvh_last_pm_ms(self._h, C.byref(v)))
        return v.value
