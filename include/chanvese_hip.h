/*
 * chanvese_hip.h — C ABI of the MI355X-native Chan-Sandberg-Vese / Perona-Malik hot path.
 *
 * This is the drop-in boundary for ktht/chan_vese: plain C, opaque context, caller-owned
 * host buffers, library-owned device buffers, one HIP stream per context, no exceptions
 * across the boundary (every entry point returns a cvh_status; cvh_last_error() gives the
 * text).  Each entry point names the reference code it replaces (file:line relative to
 * the reference repo).  The reference-side binding is shown in INTEGRATION.md.
 *
 * Layouts (identical to the reference): level set u = h*w IEEE doubles, row-major,
 * contiguous (src/main.cpp:225-227 treats u.data as double[h*w]); image channels = C
 * separate planes of h*w uint8, row-major (src/main.cpp:269-270, after cv::split :936).
 * Channel order is whatever the caller split (BGR for cv::imread colour, :879).
 *
 * Threading: calls on one context are not re-entrant; different contexts (different
 * images / GPUs) may be driven from different host threads.
 */
#ifndef CHANVESE_HIP_H
#define CHANVESE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVH_MAX_CHANNELS 3

typedef enum cvh_status {
  CVH_OK = 0,
  CVH_ERR_ARG = 1,    /* bad argument (null pointer, non-positive size, channels not 1/3 ...) */
  CVH_ERR_HIP = 2,    /* a HIP runtime call failed; see cvh_last_error() */
  CVH_ERR_STATE = 3,  /* call sequence error (e.g. run before set_image / set_levelset) */
  CVH_ERR_NOMEM = 4
} cvh_status;

/* Pixel functions the ParallelPixelFunction operator can run on the device.  The
 * reference passes an opaque std::function (include/ParallelPixelFunction.hpp:26-28);
 * its only use is delta_eps (src/main.cpp:988-989), the Heaviside pair is what
 * region_variance evaluates per pixel (src/main.cpp:265-267). */
typedef enum cvh_pixel_op {
  CVH_OP_DELTA = 0,               /* regularized_delta      src/main.cpp:204-210 */
  CVH_OP_HEAVISIDE = 1,           /* regularized_heaviside  src/main.cpp:188-194 */
  CVH_OP_ONE_MINUS_HEAVISIDE = 2  /* the Outside lambda     src/main.cpp:267     */
} cvh_pixel_op;

/* Arithmetic flavour of the fused level-set kernel (both keep FP64 state). */
typedef enum cvh_math_mode {
  CVH_MATH_DEFAULT = 0, /* library default */
  CVH_MATH_STRICT = 1,  /* IEEE sqrt and divide, no FMA contraction: every per-pixel
                           operation rounds as the reference's -O3 x86-64 code does; only
                           atan (device libm) and the summation order differ */
  CVH_MATH_FAST = 2     /* rsqrt/rcp + Newton, FMA: <= 2 ulp per operation */
} cvh_math_mode;

/* Free parameters of the CSV iteration: src/main.cpp:731-734, defaults :759-765,:812-829 */
typedef struct cvh_params {
  double mu;                        /* length penalty, default 0.5 */
  double nu;                        /* area penalty, default 0 */
  double dt;                        /* time step, default 1 */
  double eps;                       /* Heaviside/delta smoothing, default 1 */
  double tol;                       /* stop tolerance, default 0.001 */
  double lambda1[CVH_MAX_CHANNELS]; /* inside penalties, default 1 */
  double lambda2[CVH_MAX_CHANNELS]; /* outside penalties, default 1 */
} cvh_params;

typedef struct cvh_context cvh_context;

/* Fills *p with the reference defaults (src/main.cpp:759-765, :812-813, :828-829). */
void cvh_default_params(cvh_params *p);

/* Number of HIP devices visible to this process. */
int cvh_device_count(int *count);

/* Creates a context for one h x w image with `channels` (1 = -g grayscale path, 3 =
 * colour; src/main.cpp:893) on HIP device `device`.  Allocates the device planes, the
 * level-set ping-pong pair and the reduction workspace.  On failure *out is NULL and
 * cvh_last_error(NULL) describes why. */
int cvh_create(cvh_context **out, int h, int w, int channels, const cvh_params *p,
               int device);
void cvh_destroy(cvh_context *ctx);

/* Text of the last error on this context (or of the last failed cvh_create when ctx is
 * NULL).  Never NULL.  Replaces the reference's msg_exit text, src/main.cpp:173-178. */
const char *cvh_last_error(const cvh_context *ctx);

/* Replaces the parameter block (may be called between runs). */
int cvh_set_params(cvh_context *ctx, const cvh_params *p);

/* Tuning / behaviour knobs, by name (defaults are what bench.py measures):
 *   "math_mode"      cvh_math_mode
 *   "finalize"       0 = region means reduced by the last-arriving workgroup inside the step
 *                    kernel (default), 1 = by a separate one-workgroup kernel
 *   "trace"          capacity (iterations) of the per-iteration trace, 0 = off
 *   "sync_every"     iterations enqueued between host polls of the stop flag (default 32)
 *   "graph"          1 = runs of 16 steps are replayed as one hipGraph (default), 0 = plain launches
 *   "kernel"         data flow of the CSV step: -1 auto (3 where it applies -- width a multiple of 16 and >= 144, from 0.6 Mpixel; three
 *                    channels in FAST arithmetic only -- else 2; 0 from 2^28 pixels),
 *                    0 LDS tile, 2 wave-streaming, 3 wave-streaming with 2 pixels per lane (w % 16 == 0, w >= 144; other shapes
 *                    fall back to 2).  (1, the streaming-strip kernel of round 1, was removed in round 4: CVH_ERR_ARG)
 *   "resident"       -1 auto (default), 0 off, 1 on: planes whose level set fits the LDS of the chip (1 channel, FAST, even width,
 *                    at most one 128 x 128 tile per CU: up to 2048 x 2048 on an MI355X) iterate IN LDS -- one cooperative launch per
 *                    chunk of iterations, one workgroup per tile, a grid barrier per iteration, the stop rule inside the kernel at the
 *                    reference's iteration (csv_resident_kernel.hip).  Auto steps aside when "kernel", "strip_rows", "strips" or
 *                    "graph" were set (the caller asked for a per-launch flow), when "state" is 32, and -- enqueue by enqueue -- when
 *                    other co-resident contexts live on the device (a batch: cooperative launches of different contexts serialise, and
 *                    interleaved per-launch flows are twice as fast in short chunks) UNLESS the plane is large and the enqueue long:
 *                    from 48 iterations at >= 3.6 Mpixel, 72 at >= 2.9, 100 at >= 2.2 (cvh_run: chunks of up to 1024) one cooperative
 *                    launch after the other wins (eight 2048 x 2048 planes: 12.1-14.7 against 16.2 us per image-iteration).  The two
 *                    flows continue each other on one context and agree to 1e-9, not bit for bit: set 0 or 1 for the same bits
 *                    whatever the chunking
 *   "wave_pol"       cache policy of the streamed level-set rows: -1 auto (write-through stores while the ping-pong pairs and planes of
 *                    ALL contexts on the device that hold an image and a level set fit the Infinity Cache, <= 300 MB together; decided
 *                    when a run's first iteration is enqueued, kept for the run), 0 plain, 1 write-through
 *   "co_resident"    1 (default): this context streams beside the others on its GPU and counts in their automatic choices ("wave_pol",
 *                    "resident"); 0: a scratch / warm-up context that is idle while the others run
 *   "state"          64 (default): the level set lives in HBM as double -- the reference's CV_64FC1 (src/main.cpp:225), the parity mode.
 *                    32: a DECLARED fast mode that deliberately departs from the reference's type: float in HBM (9 instead of 17 bytes
 *                    per pixel-iteration; 11 instead of 19 with three channels), every new value rounded to float; arithmetic, tables
 *                    and the fixed-point sums unchanged.  2-pixel wave kernel only: FAST arithmetic, width a multiple of 16 and >= 144,
 *                    fewer than 2^28 pixels (CVH_ERR_ARG otherwise).  cvh_set_levelset / cvh_get_levelset keep exchanging doubles.
 *                    Parity bar: mask IoU >= 0.999 and median |du| / max|u| <= 1e-4 against the FP64 path
 *   "near_switch"    1 (default): a wave whose strip / band starts where most pixels are below the far-field threshold of H_eps (32 eps)
 *                    evaluates the table form of H_eps on every pixel of that strip (one form per pixel: a level set that is near
 *                    everywhere, e.g. dt << 1); 0: the far-field series with the per-group correction everywhere
 *   "tile_rows"      tile kernel: rows per tile (0 auto, 14/16)
 *   "strip_rows"     wave kernels: rows per strip (0 auto)
 *   "lut"            1 = region term from a per-launch 256-entry table (FAST, default; tile and 1-pixel wave kernels: 0 = computed)
 *   "dma"            tile kernel: 1 = global->LDS DMA loader (slower on MI355X, default 0)
 *   "pm_kernel"      Perona-Malik data flow: -1 auto (= 4 where the plane qualifies, else 3), 0 LDS tile,
 *                    1 wave-streaming, 3 wave-streaming with TWO time steps per launch (an odd last step runs flavour 1),
 *                    (2, a 2-pixel-per-lane 1-step kernel, was removed in round 4: CVH_ERR_ARG)
 *                    4 resident plane: the FP64 state of a channel stays in the LDS of the CUs for all time steps, one
 *                    cooperative launch per channel (even width, >= 16 rows and columns, <= 128 rows x 128 columns per CU:
 *                    up to 2048 x 2048 on MI355X; CVH_ERR_ARG if asked for a plane that does not qualify; auto steps
 *                    aside when "pm_strip_rows" was set)
 *   "pm_strip_rows"  Perona-Malik wave kernel: rows per strip (0 auto)
 * (Ablation / diagnostic knobs of the wave kernels are not part of this interface: they are listed in
 * chan_vese_amd/csrc/cvh_internal.h.)  Unknown keys and out-of-range values return CVH_ERR_ARG. */
int cvh_set_option(cvh_context *ctx, const char *key, long value);

/* Uploads the C channel planes (what cv::split produced, src/main.cpp:934-937) and takes their sums on the device
 * (sum I_k for the region means; the tol-free stop norm of :950-959 — exact integers for one channel, the reference's
 * serial order on the host for three). */
int cvh_set_image(cvh_context *ctx, const uint8_t *const *planes);
/* Downloads the planes (after cvh_perona_malik they hold the smoothed 8-bit image that
 * the reference writes as <stem>_pm, src/main.cpp:943-946). */
int cvh_get_image(cvh_context *ctx, uint8_t *const *planes);

/* Level set in / out (src/main.cpp:898-923 produce it, :1004-1005 consume it). */
int cvh_set_levelset(cvh_context *ctx, const double *u);
int cvh_get_levelset(cvh_context *ctx, double *u);
/* levelset_checkerboard, src/main.cpp:221-233.  The h + w sine factors are evaluated on the HOST
 * with libm sin (the sign on every fifth row/column is rounding noise that only the host libm
 * reproduces) and uploaded; the sign of their product — one IEEE multiplication — is taken on the
 * device, bit-identical to cvh_levelset_checkerboard_host without 8 bytes per pixel over PCIe. */
int cvh_init_checkerboard(cvh_context *ctx);
/* Host-only helper with the same arithmetic, for callers that keep u themselves. */
void cvh_levelset_checkerboard_host(int h, int w, double *u);

/* The timestep loop, src/main.cpp:963-1001: runs until `max_steps` further iterations
 * are done or the stop rule ||u_diff||_2 <= tol*||mean_k I_k||_2 fires (checked after the
 * update, :994-1000).  max_steps < 0 means unlimited (:890).  steps_done / last_norm
 * (may be NULL) receive the iterations executed by this call and the last ||u_diff||_2.
 * One fused HIP kernel per iteration: curvature (:342-375), region terms (:255-312,
 * :965-985), delta_eps map (ParallelPixelFunction, :988-992), update and norm (:993-994),
 * plus the Heaviside-weighted sums that give the next iteration's c1/c2 (:973-974). */
int cvh_run(cvh_context *ctx, int max_steps, int *steps_done, double *last_norm);

/* Asynchronous halves of cvh_run, for interleaving several contexts (a batch of
 * independent images) on one GPU: enqueue `nsteps` iterations on the context's stream,
 * later wait for them.  `stopped` reports whether the stop rule fired. */
int cvh_enqueue_steps(cvh_context *ctx, int nsteps);
/* Optional: does the one-off host work of an upcoming cvh_enqueue_steps(ctx, nsteps) now (strip table,
 * capture + instantiation of the 16-step hipGraph of the position the chunk starts at), so that a caller
 * timing the enqueue/sync pair with its own clock does not see it.  cvh_run and cvh_enqueue_steps do the
 * same work themselves before they open cvh_last_run_ms's interval. */
int cvh_warm(cvh_context *ctx, int nsteps);
int cvh_sync(cvh_context *ctx, int *steps_done_total, double *last_norm, int *stopped);
/* Clears the iteration counter and the stop flag (cvh_run does this itself). */
int cvh_reset_run(cvh_context *ctx);

/* Region means of the current level set (what the next iteration will use),
 * region_variance src/main.cpp:255-281; c1/c2 have `channels` entries. */
int cvh_get_means(cvh_context *ctx, double *c1, double *c2);
/* Per-iteration trace rows [c1_0..c1_{C-1}, c2_0..c2_{C-1}, norm] recorded when the
 * "trace" option is on; *rows receives the number of valid rows copied (<= max_rows). */
int cvh_get_trace(cvh_context *ctx, double *out, int max_rows, int *rows);
/* tol * || (sum_k I_k)/C ||_2, src/main.cpp:950-959 (valid after set_image). */
int cvh_get_stop_condition(cvh_context *ctx, double *stop_cond);

/* mask = ((float)u > 0), optionally 1 - mask: src/main.cpp:395-400. */
int cvh_get_mask(cvh_context *ctx, uint8_t *mask, int invert);
/* Contour map of the reference's video frame, VideoWriterManager::draw_contour
 * src/VideoWriterManager.cpp:60-74: 1 where a frame pixel is painted in the contour colour.
 * The frame's mask rule differs from cvh_get_mask: uint8(round(u)) > 0.
 * cv::findContours/drawContours are restated (see misc_kernels.hip); parity unpinned. */
int cvh_get_contour(cvh_context *ctx, uint8_t *contour);
/* separate(), src/main.cpp:386-405: img3 and selection3 are interleaved h*w*3 uint8
 * (the reference's CV_8UC3); white canvas with img3 copied where the mask is set. */
int cvh_separate(cvh_context *ctx, const uint8_t *img3, int invert, uint8_t *selection3);

/* perona_malik, src/main.cpp:478-560, applied in place to every device plane:
 * trip count from `for (double t = 0; t < T; t += L)` (:498), FP64 state, final
 * round-half-even to uint8 (:551).  Invalidates the stop condition (recomputed from the
 * smoothed planes, :950). */
int cvh_perona_malik(cvh_context *ctx, double K, double L, double T);
/* Trip count of that loop (host arithmetic). */
int cvh_pm_trip_count(double L, double T);

/* Device time (HIP events on the context's stream) of the last cvh_run / of the span
 * from the first cvh_enqueue_steps after a cvh_sync to that next cvh_sync; and of the
 * last cvh_perona_malik. */
int cvh_last_run_ms(cvh_context *ctx, float *ms);
int cvh_last_pm_ms(cvh_context *ctx, float *ms);
/* Which kernel the library launches, as "key=value" text written by the launch sites themselves (no counterpart in
 * the reference: the kernel is an implementation detail; bench.py names it in its roofline object and profiles/ are
 * matched against it).  phase 0: the CSV step as the next cvh_run / cvh_enqueue_steps launches it with the current
 * options -- kernel=<instantiation as rocprofv3 prints it> grid= block= lds_bytes= strips= strip_rows= chain= wave_pol=
 * math= steps_per_graph=; phase 1: what the last cvh_perona_malik launched (CVH_ERR_STATE before the first).
 * Launches nothing.  Truncates to cap - 1 characters. */
int cvh_launch_info(cvh_context *ctx, int phase, char *buf, int cap);

/* ParallelPixelFunction::operator()(cv::Range(start,end)) with a tagged function,
 * src/ParallelPixelFunction.cpp:12-17 — host buffer form: data is a w-wide CV_64FC1
 * matrix, elements [start,end) of its flat index range are replaced by f(x) in place
 * (copied to the device, mapped by a HIP kernel, copied back). */
int cvh_ppf_apply(double *data, int w, long start, long end, int op, double eps,
                  int device);
/* Same on a device pointer the caller owns (hip stream as void*, NULL = default). */
int cvh_ppf_apply_device(double *d_data, long n, int op, double eps, void *stream);

/* Library version string, e.g. "chanvese_hip 0.1 (gfx950)". */
const char *cvh_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CHANVESE_HIP_H */
