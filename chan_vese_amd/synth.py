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


# ---- stand-ins for the two example runs of the reference's README (README.md:51-63).  The images themselves (a Wikimedia sea star,
# 370 px wide; "Europe at night", 640 px wide) are not in the reference and there is no network: these have the same geometry and the
# same character (a textured many-armed blob on a smooth background; many small bright blobs on a dark plane), integer arithmetic only.
_ARMS11 = [(1000, 0), (841, 541), (415, 910), (-142, 990), (-655, 756), (-959, 282), (-959, -282), (-655, -756), (-142, -990),
           (415, -910), (841, -541)]      # round(1000 (cos, sin)(2 pi k / 11))


def sea_star(h=278, w=370, seed=11):
    """[B, G, R] uint8 planes: an eleven-armed star (arms = tapered segments from the centre) in orange on a blue-grey gradient, noise 12."""
    ci, cj = h // 2, w // 2
    ii = np.arange(h, dtype=np.int64)[:, None] - ci
    jj = np.arange(w, dtype=np.int64)[None, :] - cj
    length = (min(h, w) * 9) // 20
    inside = (ii * ii + jj * jj) <= (length // 4) ** 2
    for k, (dx, dy) in enumerate(_ARMS11):
        along = (jj * dx + ii * dy)                       # 1000 x the coordinate along the arm
        across = np.abs(jj * dy - ii * dx)                # 1000 x the distance from its axis
        arm_len = 1000 * (length - 6 * (k % 3))
        half = 1000 * (length // 7) * (arm_len - along) // arm_len + 1500      # tapering half-width
        inside |= (along >= 0) & (along <= arm_len) & (across <= half)
    z = splitmix64_stream(seed, 3 * h * w).reshape(3, h, w)
    planes = []
    for ch, (fg, bg, slope) in enumerate(((40, 150, 30), (120, 130, 20), (230, 90, -25))):      # B, G, R
        base = np.where(inside, fg, bg + slope * (ii + ci) // h)
        d = (z[ch] % np.uint64(25)).astype(np.int64) - 12
        planes.append(np.clip(base + d, 0, 255).astype(np.uint8))
    return planes


def night_lights(h=480, w=640, seed=29, blobs=1500):
    """[B, G, R] uint8 planes: a dark plane (level 8-20) with `blobs` small bright discs (radius 1-6) drawn in clusters, noise 4."""
    z = splitmix64_stream(seed, 4 * blobs + 3 * h * w)
    b = z[:4 * blobs].reshape(blobs, 4)
    acc = np.zeros((h, w), dtype=np.int64)
    ii = np.arange(h, dtype=np.int64)[:, None]
    jj = np.arange(w, dtype=np.int64)[None, :]
    for q in range(blobs):
        cl = int(b[q, 0] % np.uint64(12))                           # cluster: lights crowd around twelve "cities"
        ci = (h * (1 + cl % 3)) // 4 + int(b[q, 1] % np.uint64(h // 5)) - h // 10
        cj = (w * (1 + cl // 3)) // 5 + int(b[q, 2] % np.uint64(w // 6)) - w // 12
        r = 1 + int(b[q, 3] % np.uint64(6))
        acc += np.where((ii - ci) ** 2 + (jj - cj) ** 2 <= r * r, 60 + 30 * (q % 4), 0)
    noise = z[4 * blobs:].reshape(3, h, w)
    planes = []
    for ch, (gain, floor) in enumerate(((5, 20), (9, 12), (10, 8))):      # B, G, R: yellowish lights on a bluish night
        d = (noise[ch] % np.uint64(9)).astype(np.int64) - 4
        planes.append(np.clip(floor + gain * acc // 10 + d, 0, 255).astype(np.uint8))
    return planes
