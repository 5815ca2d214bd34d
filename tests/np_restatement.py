"""Independent numpy restatement of the reference hot path (vectorised, written from
/root/reference/src/main.cpp separately from oracle/cv_oracle.c) — used only to
cross-check the C oracle.  Summation order differs from the reference's sequential loops
(numpy pairwise sums), so comparisons use tolerances, not bit equality."""
import numpy as np

PI = np.pi


def heaviside(x, eps=1.0):
    return (1 + 2 / PI * np.arctan(x / eps)) / 2   # src/main.cpp:193


def delta(x, eps=1.0):
    return eps / (PI * (eps ** 2 + x ** 2))         # src/main.cpp:209


def checkerboard(h, w):
    si = np.sin(PI * np.arange(h) / 5)              # src/main.cpp:230-231
    sj = np.sin(PI * np.arange(w) / 5)
    return np.sign(si[:, None] * sj[None, :])


def region_mean(img, u, inside, eps=1.0):
    g = heaviside(u, eps)
    if not inside:
        g = 1 - g
    return float((img.astype(np.float64) * g).sum() / g.sum())   # src/main.cpp:272-280


def curvature(u):
    eta2 = 1e-8 ** 2
    p = np.pad(u, 1, mode="edge")                   # BORDER_REPLICATE on u
    c = p[1:-1, 1:-1]
    upx = p[1:-1, 2:] - c
    upy = p[2:, 1:-1] - c
    ucx = 0.5 * (p[1:-1, 2:] - p[1:-1, :-2])
    ucy = 0.5 * (p[2:, 1:-1] - p[:-2, 1:-1])
    nx = upx / np.sqrt(upx ** 2 + ucx ** 2 + eta2)  # same-axis pairing, src/main.cpp:365-368
    ny = upy / np.sqrt(upy ** 2 + ucy ** 2 + eta2)
    nxp = np.pad(nx, ((0, 0), (1, 0)), mode="edge") # BORDER_REPLICATE on nx
    nyp = np.pad(ny, ((1, 0), (0, 0)), mode="edge")
    return (nx - nxp[:, :-1]) + (ny - nyp[:-1, :])


def csv_step(planes, u, mu=0.5, nu=0.0, dt=1.0, eps=1.0, lambda1=None, lambda2=None):
    C = len(planes)
    lambda1 = [1.0] * C if lambda1 is None else lambda1
    lambda2 = [1.0] * C if lambda2 is None else lambda2
    ud = np.zeros_like(u)
    c1s, c2s = [], []
    for k, img in enumerate(planes):
        c1 = region_mean(img, u, True, eps)
        c2 = region_mean(img, u, False, eps)
        f = img.astype(np.float64)
        ud += -lambda1[k] * (f - c1) ** 2 + lambda2[k] * (f - c2) ** 2
        c1s.append(c1)
        c2s.append(c2)
    ud = dt * (mu * curvature(u) - nu + ud / C)     # src/main.cpp:985
    ud = ud * delta(u, eps)                          # :988-992
    return u + ud, float(np.sqrt((ud ** 2).sum())), c1s, c2s


def stop_condition(planes, tol):
    avg = sum(p.astype(np.float64) for p in planes) / len(planes)
    return tol * float(np.sqrt((avg ** 2).sum()))


def pm_step(I, K, L):
    h, w = I.shape
    p = np.pad(I, 1, mode="edge")
    gx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    gy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    g = 1.0 / (1.0 + (gx ** 2 + gy ** 2) / K ** 2)
    g[0, :] = 1; g[-1, :] = 1; g[:, 0] = 1; g[:, -1] = 1       # src/main.cpp:518-519
    gp = np.pad(g, 1, mode="edge")
    I0 = I
    s = ((gp[2:, 1:-1] + g) * (p[2:, 1:-1] - I0) + (gp[1:-1, 2:] + g) * (p[1:-1, 2:] - I0) +
         (gp[:-2, 1:-1] + g) * (p[:-2, 1:-1] - I0) + (gp[1:-1, :-2] + g) * (p[1:-1, :-2] - I0))
    return I0 + L * s / 4                                       # src/main.cpp:544-547


def perona_malik(img, K, L, trips):
    I = img.astype(np.float64)
    for _ in range(trips):
        I = pm_step(I, K, L)
    return np.clip(np.rint(I), 0, 255).astype(np.uint8), I


def video_contour(u):
    """VideoWriterManager::draw_contour (src/VideoWriterManager.cpp:60-74) written independently of
    oracle/cv_oracle.c: mask = uint8(round-half-even(u)) > 0, cvFindContours clears the image's outer
    ring first, every border pixel (a set pixel with a cleared 4-neighbour) of every component is drawn 1 px wide."""
    r = np.rint(np.nan_to_num(u, nan=0.0))
    m = (np.clip(r, 0, 255).astype(np.uint8) > 0)
    m[0, :] = False; m[-1, :] = False; m[:, 0] = False; m[:, -1] = False
    p = np.pad(m, 1, mode="constant", constant_values=False)
    all4 = p[:-2, 1:-1] & p[2:, 1:-1] & p[1:-1, :-2] & p[1:-1, 2:]
    return (m & ~all4).astype(np.uint8)


def levelset_rect(h, w, x, y, rw, rh):
    """InteractiveDataRect::get_levelset (src/InteractiveDataRect.cpp:20-27): zeros with ones on the rectangle
    cv::Rect(x, y, rw, rh) clipped to the image."""
    u = np.zeros((h, w))
    u[max(y, 0):max(min(y + rh, h), 0), max(x, 0):max(min(x + rw, w), 0)] = 1.0
    return u


def combine_unfused(kappa, u_diff, mu, nu, dt, C):
    """src/main.cpp:985 evaluated operation by operation (NOT what the oracle does: OpenCV's MatExpr folds
    a*K - s + U/c and the outer dt*(...) into ONE addWeighted(K, dt*mu, U, dt/C, -dt*nu))."""
    return dt * ((mu * kappa - nu) + u_diff * (1.0 / C))


def combine_folded(kappa, u_diff, mu, nu, dt, C):
    return kappa * (mu * dt) + u_diff * ((1.0 / C) * dt) + (-nu * dt)
