/*
 * cv_oracle.h — CPU restatement of the ktht/chan_vese hot path (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or sample data
 * (reference Makefile:44,50 have empty TEST/TEST_TRGT) and cannot be built in this
 * image (needs OpenCV 2.4.8 + Boost 1.59, README.md:12-13).  This oracle therefore
 * restates src/main.cpp by reading it; it is pinned only by hand-derived known-answer
 * tests (tests/test_oracle_kat.py) and by an independent numpy restatement
 * (tests/np_restatement.py).  OpenCV semantics are restated from its documented
 * behaviour (see DESIGN.md "Oracle").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 * The product path (chan_vese_amd/csrc, include/chanvese_hip.h) never links it.
 *
 * All citations are file:line under /root/reference.
 */
#ifndef CV_ORACLE_H
#define CV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CVO_INSIDE = 0, CVO_OUTSIDE = 1 }; /* include/ChanVeseCommon.hpp:19 */

typedef struct cvo_params {
  double mu;         /* src/main.cpp:759 default 0.5 */
  double nu;         /* :760 default 0 */
  double dt;         /* :761 default 1 */
  double eps;        /* :764 default 1 */
  double tol;        /* :765 default 0.001 */
  double lambda1[3]; /* :762, default 1's (:812-813) */
  double lambda2[3]; /* :763, default 1's (:828-829) */
} cvo_params;

/* src/main.cpp:188-194 */
double cvo_regularized_heaviside(double x, double eps);
/* src/main.cpp:204-210 */
double cvo_regularized_delta(double x, double eps);
/* src/main.cpp:221-233 */
void cvo_levelset_checkerboard(int h, int w, double *u);
/* src/InteractiveDataRect.cpp:20-27: zeros, ones on rows [y,y+rh) x cols [x,x+rw) */
void cvo_levelset_rect(int h, int w, int x, int y, int rw, int rh, double *u);
/* src/main.cpp:255-281 (returns a mean despite the name) */
double cvo_region_variance(const uint8_t *img, const double *u, int h, int w,
                           int region, double eps);
/* src/main.cpp:299-312 */
void cvo_variance_penalty(const uint8_t *channel, int h, int w, double c,
                          double lambda, double *out);
/* src/main.cpp:342-375 */
void cvo_curvature(const double *u, int h, int w, double *kappa);
/* src/ParallelPixelFunction.cpp:12-17 with func = delta (op 0), heaviside (1),
 * 1-heaviside (2) — flat range [start,end) of a w-wide CV_64FC1 matrix. */
void cvo_ppf_apply(double *data, int w, long start, long end, int op, double eps);
/* src/main.cpp:950-960 */
double cvo_stop_condition(const uint8_t *const *channels, int nof_channels, int h,
                          int w, double tol);
/* One pass of the loop body src/main.cpp:965-994; updates u in place, returns
 * u_diff_norm; c1/c2 (length nof_channels) receive the region means used. */
double cvo_csv_step(const uint8_t *const *channels, int nof_channels, int h, int w,
                    const cvo_params *p, double *u, double *c1, double *c2);
/* Exact-sum adjudicator (beside the restatement, not part of it): the region means of :272-280 from the same
 * per-pixel terms, added with Neumaier-compensated long double sums (accumulation error ~1e-19 instead of the
 * ~1e-12 of 16.7 M sequential double additions).  out = {sum H, sum fl(I H), sum fl(1-H), sum fl(I fl(1-H))}. */
void cvo_region_sums_exact(const uint8_t *img, const double *u, int h, int w, double eps,
                           long double out[4]);
void cvo_region_means_exact(const uint8_t *img, const double *u, int h, int w, double eps,
                            double *c1, double *c2);
/* cvo_csv_step with the region means taken from cvo_region_means_exact (everything else identical). */
double cvo_csv_step_exact(const uint8_t *const *channels, int nof_channels, int h, int w,
                          const cvo_params *p, double *u, double *c1, double *c2);
/* src/main.cpp:950-1001. trace (may be NULL) receives per iteration
 * [c1_0..c1_{C-1}, c2_0..c2_{C-1}, norm] for at most trace_cap iterations.
 * Returns the number of iterations executed (t at break, or max_steps). */
int cvo_csv_run(const uint8_t *const *channels, int nof_channels, int h, int w,
                const cvo_params *p, int max_steps, double *u, double *last_norm,
                double *trace, int trace_cap);
/* trip count of `for (double t = 0; t < T; t += L)` src/main.cpp:498 */
int cvo_pm_trip_count(double L, double T);
/* src/main.cpp:478-560, one channel; out is the CV_8UC1 result (:551,554).
 * state_out (may be NULL) receives the final double image I_prev. */
void cvo_perona_malik_channel(const uint8_t *in, int h, int w, double K, double L,
                              double T, uint8_t *out, double *state_out);
/* src/main.cpp:395-400 mask = ((float)u > 0), optional invert */
void cvo_mask(const double *u, int h, int w, int invert, uint8_t *mask);
/* src/VideoWriterManager.cpp:60-74 draw_contour: 1 where the frame gets the contour colour.
 * convertTo(CV_8U) + threshold + cv::findContours(RETR_TREE, CHAIN_APPROX_SIMPLE) + drawContours
 * (1 px, 8-connected), restated from Suzuki-Abe border following as OpenCV 2.4 applies it
 * (outer ring of the mask cleared; border point of an 8-connected component = 1-pixel with a
 * 0-pixel among its 4 neighbours).  Parity unpinned: no OpenCV in this image. */
void cvo_video_contour(const double *u, int h, int w, uint8_t *contour);
/* src/main.cpp:386-405: interleaved 3-channel img (h*w*3) -> selection (h*w*3) */
void cvo_separate(const uint8_t *img3, const double *u, int h, int w, int invert,
                  uint8_t *selection3);

#ifdef __cplusplus
}
#endif
#endif
