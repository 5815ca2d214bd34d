/*
 * cv_oracle.c — CPU restatement of the ktht/chan_vese hot path.
 *
 * TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (see cv_oracle.h).  This file keeps the
 * reference's pass structure (separate sweeps for c1 and c2 through an indirect
 * call, one array pass per OpenCV call, the same OpenMP pragmas and team sizes) so
 * that it doubles as the "reference-faithful CPU path" timed by bench.py's
 * cpu_baseline leg (kind "port").  Build with oracle/Makefile, which uses the
 * reference's own flags (reference Makefile:6,16).
 *
 * OpenCV 2.4 semantics restated here (from its documented behaviour; not verifiable
 * offline — every plausible variant differs by <= 1 ulp per operation):
 *   filter2D      correlation, anchor at kernel centre, BORDER_REPLICATE, zero
 *                 coefficients skipped => plain differences.
 *   Sobel k=3     separable [-1 0 1] x [1 2 1], unnormalised, row pass first.
 *   convertTo 8U  cvRound (round-half-to-even) then clamp to [0,255].
 *   MatExpr       a*A - s + B/c  ==> addWeighted(A, a, B, 1/c, -s); dt*(..) scales
 *                 all three coefficients (src/main.cpp:985).
 *   -A + B        ==> subtract(B, A)   (src/main.cpp:979)
 *   norm L2       sqrt(sum x^2), accumulated four squares at a time.
 *   threshold     src > 0 ? 1 : 0 on the float copy (src/main.cpp:397-398).
 * All citations are file:line under /root/reference.
 */
#include "cv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef NUM_THREADS
#define NUM_THREADS 3 /* reference Makefile:6 */
#endif

static const double CVO_PI = 3.14159265358979323846; /* boost pi<double>() */

/* Stand-in for std::function<double(double)> bound with std::bind (src/main.cpp:
 * 894-895): an indirect call carrying its epsilon. */
typedef struct cvo_closure {
  double (*fn)(double, const struct cvo_closure *);
  double eps;
  const struct cvo_closure *inner;
} cvo_closure;

static double call(const cvo_closure *c, double x) { return c->fn(x, c); }

double cvo_regularized_heaviside(double x, double eps)
{
  /* src/main.cpp:193 */
  return (1 + 2 / CVO_PI * atan(x / eps)) / 2;
}

double cvo_regularized_delta(double x, double eps)
{
  /* src/main.cpp:209; std::pow(z, 2) is z*z */
  return eps / (CVO_PI * (eps * eps + x * x));
}

static double heaviside_thunk(double x, const cvo_closure *c)
{
  return cvo_regularized_heaviside(x, c->eps);
}
static double delta_thunk(double x, const cvo_closure *c)
{
  return cvo_regularized_delta(x, c->eps);
}
/* the lambda at src/main.cpp:267 */
static double one_minus_thunk(double x, const cvo_closure *c)
{
  return 1 - call(c->inner, x);
}

static int sign_of(double z) /* boost::math::sign */
{
  return (z == 0) ? 0 : (z < 0 ? -1 : 1);
}

void cvo_levelset_checkerboard(int h, int w, double *u)
{
  /* src/main.cpp:228-231 */
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j)
      u[(size_t)i * w + j] = sign_of(sin(CVO_PI * i / 5) * sin(CVO_PI * j / 5));
}

void cvo_levelset_rect(int h, int w, int x, int y, int rw, int rh, double *u)
{
  /* src/InteractiveDataRect.cpp:24-25 */
  memset(u, 0, sizeof(double) * (size_t)h * w);
  for (int i = y; i < y + rh && i < h; ++i)
    for (int j = x; j < x + rw && j < w; ++j)
      if (i >= 0 && j >= 0) u[(size_t)i * w + j] = 1;
}

static double region_variance_fn(const uint8_t *img, const double *u, int h, int w,
                                 int region, const cvo_closure *heaviside)
{
  /* src/main.cpp:263-280 */
  double nom = 0.0, denom = 0.0;
  cvo_closure outside = {one_minus_thunk, 0.0, heaviside};
  const cvo_closure *H = (region == CVO_INSIDE) ? heaviside : &outside;
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j) {
      const double hv = call(H, u[(size_t)i * w + j]);
      nom += img[(size_t)i * w + j] * hv;
      denom += hv;
    }
  return nom / denom;
}

double cvo_region_variance(const uint8_t *img, const double *u, int h, int w,
                           int region, double eps)
{
  cvo_closure hs = {heaviside_thunk, eps, NULL};
  return region_variance_fn(img, u, h, w, region, &hs);
}

void cvo_variance_penalty(const uint8_t *channel, int h, int w, double c,
                          double lambda, double *out)
{
  /* src/main.cpp:306-310, one pass per OpenCV call */
  const size_t n = (size_t)h * w;
  memset(out, 0, n * sizeof(double));                   /* Mat::zeros */
  for (size_t k = 0; k < n; ++k) out[k] = channel[k];   /* convertTo */
  for (size_t k = 0; k < n; ++k) out[k] -= c;           /* -= c */
  for (size_t k = 0; k < n; ++k) out[k] = out[k] * out[k]; /* cv::pow(., 2) */
  for (size_t k = 0; k < n; ++k) out[k] = out[k] * lambda; /* *= lambda */
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void cvo_curvature(const double *u, int h, int w, double *kappa)
{
  /* src/main.cpp:347-374 */
  const double eta = 1E-8;
  const double eta2 = eta * eta;
  const size_t n = (size_t)h * w;
  double *upx = (double *)malloc(n * sizeof(double));
  double *upy = (double *)malloc(n * sizeof(double));
  double *ucx = (double *)malloc(n * sizeof(double));
  double *ucy = (double *)malloc(n * sizeof(double));
  double *tmp = (double *)malloc(n * sizeof(double));

  /* four filter2D passes, BORDER_REPLICATE on u (:351-354) */
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j)
      upx[(size_t)i * w + j] = u[(size_t)i * w + clampi(j + 1, 0, w - 1)] - u[(size_t)i * w + j];
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j)
      upy[(size_t)i * w + j] = u[(size_t)clampi(i + 1, 0, h - 1) * w + j] - u[(size_t)i * w + j];
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j)
      ucx[(size_t)i * w + j] = -0.5 * u[(size_t)i * w + clampi(j - 1, 0, w - 1)] +
                               0.5 * u[(size_t)i * w + clampi(j + 1, 0, w - 1)];
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j)
      ucy[(size_t)i * w + j] = -0.5 * u[(size_t)clampi(i - 1, 0, h - 1) * w + j] +
                               0.5 * u[(size_t)clampi(i + 1, 0, h - 1) * w + j];

#pragma omp parallel for num_threads(NUM_THREADS)
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j) {
      const size_t k = (size_t)i * w + j;
      upx[k] = upx[k] / sqrt(upx[k] * upx[k] + ucx[k] * ucx[k] + eta2); /* :365-366 */
      upy[k] = upy[k] / sqrt(upy[k] * upy[k] + ucy[k] * ucy[k] + eta2); /* :367-368 */
    }

  /* backward differences with BORDER_REPLICATE on the normalised fields (:371-372) */
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j)
      tmp[(size_t)i * w + j] = upx[(size_t)i * w + j] - upx[(size_t)i * w + clampi(j - 1, 0, w - 1)];
  memcpy(upx, tmp, n * sizeof(double));
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j)
      tmp[(size_t)i * w + j] = upy[(size_t)i * w + j] - upy[(size_t)clampi(i - 1, 0, h - 1) * w + j];
  memcpy(upy, tmp, n * sizeof(double));
  for (size_t k = 0; k < n; ++k) kappa[k] = upx[k] + upy[k]; /* :373 */

  free(upx); free(upy); free(ucx); free(ucy); free(tmp);
}

void cvo_ppf_apply(double *data, int w, long start, long end, int op, double eps)
{
  /* src/ParallelPixelFunction.cpp:15-16 */
  cvo_closure hs = {heaviside_thunk, eps, NULL};
  cvo_closure dl = {delta_thunk, eps, NULL};
  cvo_closure om = {one_minus_thunk, 0.0, &hs};
  const cvo_closure *f = op == 0 ? &dl : (op == 1 ? &hs : &om);
  for (long i = start; i != end; ++i) {
    double *p = &data[(size_t)(i / w) * w + (i % w)];
    *p = call(f, *p);
  }
}

static double norm_l2(const double *a, size_t n)
{
  /* cv::norm(NORM_L2) on CV_64F: four squares per step, then sqrt */
  double s = 0;
  size_t i = 0;
  for (; i + 4 <= n; i += 4) {
    const double v0 = a[i], v1 = a[i + 1], v2 = a[i + 2], v3 = a[i + 3];
    s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
  }
  for (; i < n; ++i) s += a[i] * a[i];
  return sqrt(s);
}

double cvo_stop_condition(const uint8_t *const *channels, int nof_channels, int h,
                          int w, double tol)
{
  /* src/main.cpp:950-959; race-free meaning: zero-initialised, serial in k */
  const size_t n = (size_t)h * w;
  double *avg = (double *)calloc(n, sizeof(double));
  for (int k = 0; k < nof_channels; ++k)
    for (size_t q = 0; q < n; ++q) avg[q] += (double)channels[k][q];
  const double inv = 1.0 / nof_channels; /* Mat /= s  ==> scale by 1/s */
  for (size_t q = 0; q < n; ++q) avg[q] = avg[q] * inv;
  const double r = tol * norm_l2(avg, n);
  free(avg);
  return r;
}

/* ---- exact-sum adjudicator (test infrastructure beside the restatement, not part of it) ------------------
 * The region means of src/main.cpp:272-280 with the SAME per-pixel terms the reference adds -- hv = H_eps(u)
 * for the inside, fl(1 - H_eps(u)) for the outside, fl(I * hv) -- but added without accumulation error:
 * Neumaier-compensated long double sums per row, the rows combined the same way (so the result does not depend
 * on the thread count).  The reference's own sequential double sums of 16.7 M terms carry ~1e-12 relative
 * rounding error at 4096^2; this variant says which of two trajectories that differ by that much is the
 * accurate one (tests/test_exact_sums.py, tests/test_gpu_fullsize.py). */
typedef struct { long double s, c; } cvo_acc;

static void acc_add(cvo_acc *a, long double x)
{
  const long double t = a->s + x;
  if (fabsl(a->s) >= fabsl(x)) a->c += (a->s - t) + x;
  else a->c += (x - t) + a->s;
  a->s = t;
}

void cvo_region_sums_exact(const uint8_t *img, const double *u, int h, int w, double eps,
                           long double out[4])
{
  cvo_acc *rows = (cvo_acc *)calloc((size_t)h * 4, sizeof(cvo_acc));
#pragma omp parallel for schedule(static)
  for (int i = 0; i < h; ++i) {
    cvo_acc a[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int j = 0; j < w; ++j) {
      const double hv = cvo_regularized_heaviside(u[(size_t)i * w + j], eps); /* :193 */
      const double ho = 1 - hv;                                               /* :267 */
      const double pix = img[(size_t)i * w + j];
      acc_add(&a[0], hv);
      acc_add(&a[1], pix * hv);                                               /* :276, product rounded to double */
      acc_add(&a[2], ho);
      acc_add(&a[3], pix * ho);
    }
    for (int s = 0; s < 4; ++s) rows[(size_t)i * 4 + s] = a[s];
  }
  for (int s = 0; s < 4; ++s) {
    cvo_acc t = {0, 0};
    for (int i = 0; i < h; ++i) { acc_add(&t, rows[(size_t)i * 4 + s].s); acc_add(&t, rows[(size_t)i * 4 + s].c); }
    out[s] = t.s + t.c;
  }
  free(rows);
}

void cvo_region_means_exact(const uint8_t *img, const double *u, int h, int w, double eps,
                            double *c1, double *c2)
{
  long double s[4];
  cvo_region_sums_exact(img, u, h, w, eps, s);
  *c1 = (double)(s[1] / s[0]); /* nom / denom, :280 */
  *c2 = (double)(s[3] / s[2]);
}

/* src/main.cpp:977-994 given the region means of this iteration */
static double csv_step_given_means(const uint8_t *const *channels, int nof_channels, int h, int w,
                                   const cvo_params *p, double *u, const double *c1, const double *c2)
{
  const size_t n = (size_t)h * w;
  double *u_diff = (double *)calloc(n, sizeof(double)); /* :965 */
  double *vin = (double *)malloc(n * sizeof(double));
  double *vout = (double *)malloc(n * sizeof(double));
  double *kappa = (double *)malloc(n * sizeof(double));
  double *u_cp = (double *)malloc(n * sizeof(double));

  for (int k = 0; k < nof_channels; ++k) {
    cvo_variance_penalty(channels[k], h, w, c1[k], p->lambda1[k], vin);  /* :977 */
    cvo_variance_penalty(channels[k], h, w, c2[k], p->lambda2[k], vout); /* :978 */
    for (size_t q = 0; q < n; ++q) u_diff[q] += vout[q] - vin[q];        /* :979 */
  }
  cvo_curvature(u, h, w, kappa); /* :982 */

  /* :985  dt*(mu*kappa - nu + u_diff/N) as one addWeighted */
  const double alpha = p->mu * p->dt;
  const double beta = (1.0 / nof_channels) * p->dt;
  const double gamma = -p->nu * p->dt;
  for (size_t q = 0; q < n; ++q) u_diff[q] = kappa[q] * alpha + u_diff[q] * beta + gamma;

  memcpy(u_cp, u, n * sizeof(double)); /* :988 */
  /* :989 cv::parallel_for_ over [0, h*w): stripes over the host threads */
#pragma omp parallel
  {
#ifdef _OPENMP
    extern int omp_get_num_threads(void);
    extern int omp_get_thread_num(void);
    const long nt = omp_get_num_threads(), tid = omp_get_thread_num();
#else
    const long nt = 1, tid = 0;
#endif
    const long lo = (long)n * tid / nt, hi = (long)n * (tid + 1) / nt;
    cvo_ppf_apply(u_cp, w, lo, hi, 0, p->eps);
  }
  for (size_t q = 0; q < n; ++q) u_diff[q] = u_diff[q] * u_cp[q]; /* :992 */
  const double nrm = norm_l2(u_diff, n);                          /* :993 */
  for (size_t q = 0; q < n; ++q) u[q] += u_diff[q];               /* :994 */

  free(u_diff); free(vin); free(vout); free(kappa); free(u_cp);
  return nrm;
}

double cvo_csv_step(const uint8_t *const *channels, int nof_channels, int h, int w,
                    const cvo_params *p, double *u, double *c1, double *c2)
{
  cvo_closure hs = {heaviside_thunk, p->eps, NULL};
  /* channel loop :968-980, serial in k (the race-free meaning); the two serial sweeps per channel come first here,
   * the penalties (which only need c1[k], c2[k]) follow in csv_step_given_means in the same k order */
  for (int k = 0; k < nof_channels; ++k) {
    c1[k] = region_variance_fn(channels[k], u, h, w, CVO_INSIDE, &hs);  /* :973 */
    c2[k] = region_variance_fn(channels[k], u, h, w, CVO_OUTSIDE, &hs); /* :974 */
  }
  return csv_step_given_means(channels, nof_channels, h, w, p, u, c1, c2);
}

double cvo_csv_step_exact(const uint8_t *const *channels, int nof_channels, int h, int w,
                          const cvo_params *p, double *u, double *c1, double *c2)
{
  for (int k = 0; k < nof_channels; ++k) cvo_region_means_exact(channels[k], u, h, w, p->eps, &c1[k], &c2[k]);
  return csv_step_given_means(channels, nof_channels, h, w, p, u, c1, c2);
}

int cvo_csv_run(const uint8_t *const *channels, int nof_channels, int h, int w,
                const cvo_params *p, int max_steps, double *u, double *last_norm,
                double *trace, int trace_cap)
{
  const double stop_cond = cvo_stop_condition(channels, nof_channels, h, w, p->tol);
  double c1[3], c2[3], nrm = 0;
  int t, done = 0;
  for (t = 1; t <= max_steps; ++t) { /* :963 */
    nrm = cvo_csv_step(channels, nof_channels, h, w, p, u, c1, c2);
    done = t;
    if (trace && t <= trace_cap) {
      double *row = trace + (size_t)(t - 1) * (2 * nof_channels + 1);
      for (int k = 0; k < nof_channels; ++k) { row[k] = c1[k]; row[nof_channels + k] = c2[k]; }
      row[2 * nof_channels] = nrm;
    }
    if (nrm <= stop_cond) break; /* :1000 */
  }
  if (last_norm) *last_norm = nrm;
  return done;
}

int cvo_pm_trip_count(double L, double T)
{
  int n = 0;
  for (double t = 0; t < T; t += L) ++n; /* src/main.cpp:498 */
  return n;
}

static uint8_t saturate_u8(double v)
{
  const long r = lrint(v); /* cvRound: round half to even */
  return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

void cvo_perona_malik_channel(const uint8_t *in, int h, int w, double K, double L,
                              double T, uint8_t *out, double *state_out)
{
  /* src/main.cpp:492-554 for one k */
  const size_t n = (size_t)h * w;
  double *I_prev = (double *)malloc(n * sizeof(double));
  double *I_curr = (double *)malloc(n * sizeof(double));
  double *g = (double *)malloc(n * sizeof(double));
  double *dx = (double *)malloc(n * sizeof(double));
  double *dy = (double *)malloc(n * sizeof(double));
  double *row = (double *)malloc(n * sizeof(double));
  for (size_t q = 0; q < n; ++q) I_prev[q] = in[q];
  for (size_t q = 0; q < n; ++q) out[q] = in[q];

  for (double t = 0; t < T; t += L) { /* :498 */
    /* cv::Sobel dx: row pass [-1 0 1], column pass [1 2 1].  Border rows/cols are
     * computed with replicate here; g is forced to 1 there, so they are unobserved. */
    for (int i = 0; i < h; ++i)
      for (int j = 0; j < w; ++j)
        row[(size_t)i * w + j] = I_prev[(size_t)i * w + clampi(j + 1, 0, w - 1)] -
                                 I_prev[(size_t)i * w + clampi(j - 1, 0, w - 1)];
    for (int i = 0; i < h; ++i)
      for (int j = 0; j < w; ++j)
        dx[(size_t)i * w + j] = row[(size_t)clampi(i - 1, 0, h - 1) * w + j] +
                                row[(size_t)i * w + j] * 2 +
                                row[(size_t)clampi(i + 1, 0, h - 1) * w + j];
    /* cv::Sobel dy: row pass [1 2 1], column pass [-1 0 1] */
    for (int i = 0; i < h; ++i)
      for (int j = 0; j < w; ++j)
        row[(size_t)i * w + j] = I_prev[(size_t)i * w + clampi(j - 1, 0, w - 1)] +
                                 I_prev[(size_t)i * w + j] * 2 +
                                 I_prev[(size_t)i * w + clampi(j + 1, 0, w - 1)];
    for (int i = 0; i < h; ++i)
      for (int j = 0; j < w; ++j)
        dy[(size_t)i * w + j] = row[(size_t)clampi(i + 1, 0, h - 1) * w + j] -
                                row[(size_t)clampi(i - 1, 0, h - 1) * w + j];
    memset(I_curr, 0, n * sizeof(double)); /* :505 */

    for (int i = 0; i < h; ++i) /* :513-522 */
      for (int j = 0; j < w; ++j) {
        const double gx = dx[(size_t)i * w + j];
        const double gy = dy[(size_t)i * w + j];
        const double d = (i == 0 || i == h - 1 || j == 0 || j == w - 1)
                             ? 1
                             : 1.0 / (1.0 + (gx * gx + gy * gy) / (K * K));
        g[(size_t)i * w + j] = d;
      }

    for (int i = 0; i < h; ++i) /* :524-548 */
      for (int j = 0; j < w; ++j) {
        const int in_ = i == h - 1 ? i : i + 1;
        const int ip = i == 0 ? i : i - 1;
        const int jn = j == w - 1 ? j : j + 1;
        const int jp = j == 0 ? j : j - 1;
        const double Is = I_prev[(size_t)in_ * w + j];
        const double Ie = I_prev[(size_t)i * w + jn];
        const double In = I_prev[(size_t)ip * w + j];
        const double Iw = I_prev[(size_t)i * w + jp];
        const double I0 = I_prev[(size_t)i * w + j];
        const double cs = g[(size_t)in_ * w + j];
        const double ce = g[(size_t)i * w + jn];
        const double cn = g[(size_t)ip * w + j];
        const double cw = g[(size_t)i * w + jp];
        const double c0 = g[(size_t)i * w + j];
        I_curr[(size_t)i * w + j] = I0 + L * ((cs + c0) * (Is - I0) +
                                              (ce + c0) * (Ie - I0) +
                                              (cn + c0) * (In - I0) +
                                              (cw + c0) * (Iw - I0)) / 4;
      }
    memcpy(I_prev, I_curr, n * sizeof(double));               /* :550 */
    for (size_t q = 0; q < n; ++q) out[q] = saturate_u8(I_prev[q]); /* :551 */
  }
  if (state_out) memcpy(state_out, I_prev, n * sizeof(double));
  free(I_prev); free(I_curr); free(g); free(dx); free(dy); free(row);
}

void cvo_mask(const double *u, int h, int w, int invert, uint8_t *mask)
{
  /* src/main.cpp:397-400 */
  const size_t n = (size_t)h * w;
  for (size_t q = 0; q < n; ++q) {
    const float f = (float)u[q];
    uint8_t m = f > 0 ? 1 : 0;
    mask[q] = invert ? (uint8_t)(1 - m) : m;
  }
}

void cvo_video_contour(const double *u, int h, int w, uint8_t *contour)
{
  const size_t n = (size_t)h * w;
  uint8_t *m = (uint8_t *)calloc(n ? n : 1, 1);
  /* :65-67 u.convertTo(CV_8UC1) = saturate_cast<uchar>(cvRound(x)) (round half to even), then
   * threshold(> 0 -> 1) */
  for (size_t q = 0; q < n; ++q) {
    double r = nearbyint(u[q]);
    if (!(r > 0)) r = 0;         /* negatives and NaN saturate to 0 */
    if (r > 255) r = 255;
    m[q] = ((uint8_t)r) > 0 ? 1 : 0;
  }
  /* cvFindContours works on a copy whose first/last rows and columns are cleared */
  for (int j = 0; j < w; ++j) { m[j] = 0; m[(size_t)(h - 1) * w + j] = 0; }
  for (int i = 0; i < h; ++i) { m[(size_t)i * w] = 0; m[(size_t)i * w + (w - 1)] = 0; }
  /* :68-72 every border (outer and hole) of every 8-connected component, redrawn 1 px wide */
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j) {
      const size_t q = (size_t)i * w + j;
      uint8_t c = 0;
      if (m[q]) {
        /* inside the cleared ring every 4-neighbour exists */
        const int zero_nb = !m[q - w] || !m[q + w] || !m[q - 1] || !m[q + 1];
        c = zero_nb ? 1 : 0;
      }
      contour[q] = c;
    }
  free(m);
}

void cvo_separate(const uint8_t *img3, const double *u, int h, int w, int invert,
                  uint8_t *selection3)
{
  /* src/main.cpp:402-403: white canvas, image copied where mask != 0 */
  const size_t n = (size_t)h * w;
  uint8_t *mask = (uint8_t *)malloc(n);
  cvo_mask(u, h, w, invert, mask);
  for (size_t q = 0; q < n; ++q)
    for (int c = 0; c < 3; ++c) selection3[q * 3 + c] = mask[q] ? img3[q * 3 + c] : 255;
  free(mask);
}
