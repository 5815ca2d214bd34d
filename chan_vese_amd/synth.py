"""Bit-reproducible synthetic inputs (integer-only generation) — SURVEY.md §8(d).

disk(n, fg, bg, noise, seed): I(i,j) = fg inside the disk of radius n//4 centred at
(n//2, n//2) (inclusive, d^2 <= r^2), bg outside; optional uniform integer noise in
[-A, A] drawn per pixel in row-major order from splitmix64(seed), then clamped to [0,255].
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_stream(seed, count):
    """First `count` outputs of splitmix64 seeded with `seed` (uint64 arithmetic only)."""
    with np.errstate(over="ignore"):
        k = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def disk(n, fg=200, bg=50, noise=0, seed=0, radius=None, h=None, w=None):
    """uint8 (h, w) plane; h = w = n unless given."""
    h = n if h is None else h
    w = n if w is None else w
    r = (n // 4) if radius is None else radius
    ci, cj = h // 2, w // 2
    ii = np.arange(h, dtype=np.int64)[:, None] - ci
    jj = np.arange(w, dtype=np.int64)[None, :] - cj
    img = np.where(ii * ii + jj * jj <= r * r, fg, bg).astype(np.int64)
    if noise > 0:
        z = splitmix64_stream(seed, h * w)
        d = (z % np.uint64(2 * noise + 1)).astype(np.int64) - noise
        img = img + d.reshape(h, w)
    return np.clip(img, 0, 255).astype(np.uint8)


def config_planes(name, n=None):
    """Planes for the BASELINE.json configs C1..C5. Returns list of planes."""
    if name == "C1":
        return [disk(n or 512)]
    if name == "C2":
        return [disk(n or 4096)]
    if name == "C3":  # B, G, R disks
        m = n or 4096
        return [disk(m, 180, 40), disk(m, 200, 60), disk(m, 60, 200)]
    if name == "C4":
        return [disk(n or 2048, 200, 50, noise=32, seed=1)]
    raise ValueError(name)


def batch_image(b, n=4096):
    """Image b of config C5: radius n/4 + 8*(b mod 8) - 28, noise 16, seed 1000+b."""
    return disk(n, 200, 50, noise=16, seed=1000 + b, radius=n // 4 + 8 * (b % 8) - 28)
