// api.hip — host side of the C ABI declared in include/chanvese_hip.h.
// Owns the device buffers, the stream and the launch sequence of one context; never throws.
#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "cvh_internal.h"

constexpr int kGraphSteps = 16;   // steps per captured graph (even: the ping-pong parity repeats)
struct StepGraph { hipGraphExec_t exec = nullptr; CvhStepArgs key[4]; int kind = -1, flavour = -1; };

struct cvh_context {
  int h = 0, w = 0, C = 0, device = 0;
  size_t n = 0;
  cvh_params p{};
  hipStream_t stream = nullptr;
  uint8_t *d_img[CVH_MAX_CHANNELS] = {nullptr, nullptr, nullptr};   // planes inside d_img_slab, img_stride bytes apart
  uint8_t *d_img_slab = nullptr;
  size_t img_stride = 0;
  double *d_u[2] = {nullptr, nullptr};
  void *d_u_slab = nullptr;
  // option "state" = 32 (declared FP32-state mode): the iteration kernels read and write d_uf[]; d_u[] stays the exchange format of
  // set / get / mask / contour / selection and of the initial sums -- a mirror, refreshed lazily (ensure_f64_mirror)
  int state_bits = 64;
  float *d_uf[2] = {nullptr, nullptr};
  void *d_uf_slab = nullptr;
  bool mirror_valid = true;     // d_u[current] holds the level set (always true with 64-bit state)
  // automatic cache policy of a run ("wave_pol" = -1): decided when the run's first iteration is enqueued, from the footprint of EVERY
  // context on this device that holds an image and a level set (live_footprint), and kept until the run counter is reset
  int co_resident = 1;          // option "co_resident": 0 = a scratch / warm-up context that does not stream beside the others
  mutable int run_pol = -1;     // the decision of the current run (-1: not taken yet)
  mutable int run_alone = -1;   // 1: no other co-resident context on the device when the run started (automatic resident flow allowed)
  int run_chunk = -1;           // iterations of the enqueue at hand (cvh_enqueue_steps / cvh_warm: their argument; cvh_run: its chunk) -- how long a cooperative launch would be (-1: nothing announced yet)
  CvhState *d_state = nullptr;
  CvhState *h_state = nullptr;  // pinned, four slots for pipelined polling
  double *d_partials = nullptr;
  int partial_rows = 0;
  double *d_trace = nullptr;
  int trace_cap = 0;
  double *d_pm[2] = {nullptr, nullptr};
  uint8_t *d_mask = nullptr;
  bool have_image = false, have_u = false, sums_valid = false, stop_valid = false;
  double stop_norm = 0.0;  // || (sum_k I_k)/C ||_2
  double stop_cond_h = 0.0; // tol * stop_norm of the current run (a launch argument)
  int math_mode = CVH_MATH_DEFAULT, finalize_mode = 0, sync_every = 32;
  int tile_rows = 0 /* auto */, use_lut = 1, use_dma = 0;
  int kernel = -1;      // -1 auto, 0 tile kernel, 2 wave kernel, 3 wave kernel with 2 pixels per lane
  int pm_kernel = -1;   // -1 auto (4 where the plane and the run qualify, else 3), 0 tile kernel, 1 wave kernel, 3 two time steps per launch, 4 resident plane
  int pm_strip_rows = 0;
  int wave_minw = 5, wave_lds_cap = 0, wave_prio = 1, wave_sync = -1 /* auto: 1 channel 1, 3 channels 0 */, wave_imgv = 1, wave_depth = 4;
  int res_prio = 1;     // option "res_prio": resident kernels, priority by quarters of a wave's band (csv_resident_kernel.hip)
  int res_go_share = 5;  // option "res_go_share": log2 of the tiles of an XCD that share one release line of the resident kernel (0: a line per tile, 5: a line per XCD, 6: one line)
  int near_switch = 1;  // option "near_switch": per-wave, per-group choice of the form of H_eps (csv_wave2_kernel.hip); 0 = far form + correction always
  double *d_dummy = nullptr;
  int wave_rev = 0, wave_xcd = 1;
  int use_graph = 1;
  StepGraph graphs[4];          // by the chain-mode sum set of the first step, (chain_pb + enqueued) & 3; the ping-pong parity
                                // follows it (cur_base == chain_pb mod 2: set_levelset / init_checkerboard keep that invariant)
  char pm_desc[256] = {0};      // what the last cvh_perona_malik launched (cvh_launch_info)
  hipGraphExec_t pm_graph = nullptr;   // 16 Perona-Malik steps starting from d_pm[0]
  CvhPmArgs pm_graph_key{};
  int pm_graph_kind = -1;
  int wave_skew = 0;            // per-mille: older workgroups get longer strips (see upload_strip_bounds)
  // chain mode of the 2-pixel wave kernel (cvh_internal.h, CvhChainAcc)
  CvhChainAcc *d_chain = nullptr;
  int chain_opt = 1;            // option "chain"
  int chain_pb = 0;             // sum set that belongs to the level set at run-counter 0
  bool chain_pending = false;   // chain launches enqueued since the last flush
  bool chain_acc_valid = false; // the fixed-point sets hold the sums of the current level set
  // resident kernel (csv_resident_kernel.hip): cache-resident planes iterate in LDS, one cooperative launch per chunk
  CvhResident *d_resident = nullptr;
  double *d_res_halo = nullptr;
  double *d_pm_halo = nullptr;   // pm_resident_kernel's border entries {value, tag}: its own buffer (tags must never meet foreign data)
  int *h_resident = nullptr;     // pinned: {arrive, error} of the last launch
  int res_straight = 1;          // diagnostic option "res_straight": 0 = the generic march of csv_resident_kernel whatever the tile height
  int pm_resident_cap = -1;      // workgroups of pm_resident_kernel the device holds at once (-1: not asked yet)
  unsigned pm_res_serial = 0;    // launches of pm_resident_kernel so far (tag of the border entries; 0 = the cleared buffer)
  int resident_opt = -1;         // option "resident": -1 auto (on where it applies, unless a per-launch knob was set), 0 off, 1 on where it applies
  int resident_cap = -1;         // workgroups the device holds at once (-1: not asked yet, 0: unavailable)
  bool resident_used = false;    // a resident launch since the last sync: its error word is checked there
  int far_terms = 5;            // terms of the far-field series of H_eps (5: valid from 32 eps, 4: from 64 eps)
  int wave_pol = -1;            // option "wave_pol": cache policy of the 2-pixel kernel's rows (-1 auto by footprint, 0 plain, 1 write-through)
  int wave_cls = 1;             // 2-pixel wave kernel: class-major workgroup numbering (dispatch rounds)
  int wave_cskew = 500;         // per-mille strip-length skew between dispatch rounds (see upload_strip_bounds); measured
                                // in one process at 4096^2: 0 -> 61.1, 300 -> 59.3, 500 -> 58.7, 750 -> 58.5, 900 -> 59.2 us
  int *d_bounds = nullptr;      // wave kernel: first row of every strip, [tiles_y + 1]
  int bounds_key[4] = {-1, -1, -1, -1};
  int *h_status = nullptr;  // pinned + mapped: {steps_done, stopped} written by the device
  unsigned long long *d_isums = nullptr, *h_isums = nullptr;   // image_sums_kernel: {sum p, sum p^2} per plane (device / pinned)
  int strip_rows = 0;   // 0 auto
  int strips = 0;       // 2-pixel kernel: exact number of strips (0 auto); rows are dealt by cumulative weight, so any count works
  int num_cus = 256;
  double *d_atan = nullptr;
  unsigned long long *d_dbg = nullptr;  // diagnostic stamps (option "debug_times")
  size_t dbg_words = 0;
  double sum_img[CVH_MAX_CHANNELS] = {0, 0, 0};
  int tiles_x = 0, tiles_y = 0;
  int cur_base = 0;   // buffer that held u when the run counter was last reset
  int enqueued = 0;   // steps enqueued since then
  int steps_done = 0; // as of the last sync
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evp[4] = {nullptr, nullptr, nullptr, nullptr};
  bool timing_open = false;
  float last_run_ms = 0.f, last_pm_ms = 0.f;
  char err[512] = {0};
};

static char g_create_err[512] = "no error";

static int fail(cvh_context *ctx, int code, const char *fmt, ...)
{
  char *dst = ctx ? ctx->err : g_create_err;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCHK(ctx, call)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail((ctx), CVH_ERR_HIP, "HIP error %d (%s) in %s", (int)e_, hipGetErrorString(e_), \
                  #call);                                                                      \
  } while (0)

static bool use_fast(const cvh_context *c)
{
  const int m = c->math_mode == CVH_MATH_DEFAULT ? CVH_MATH_FAST : c->math_mode;
  return m == CVH_MATH_FAST;
}

extern "C" const char *cvh_version(void) { return "chanvese_hip 0.1 (gfx950)"; }

void cvh_fill_note(CvhLaunchNote *note, unsigned grid, unsigned block, size_t lds, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(note->name, sizeof(note->name), fmt, ap);
  va_end(ap);
  note->grid = grid; note->block = block; note->lds = (unsigned)lds;
}

extern "C" void cvh_default_params(cvh_params *p)
{
  if (!p) return;
  p->mu = 0.5; p->nu = 0.0; p->dt = 1.0; p->eps = 1.0; p->tol = 0.001;  // src/main.cpp:759-765
  for (int k = 0; k < CVH_MAX_CHANNELS; ++k) { p->lambda1[k] = 1.0; p->lambda2[k] = 1.0; }
}

extern "C" int cvh_device_count(int *count)
{
  if (!count) return CVH_ERR_ARG;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(nullptr, CVH_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *count = n;
  return CVH_OK;
}

extern "C" const char *cvh_last_error(const cvh_context *ctx) { return ctx ? ctx->err : g_create_err; }

static int check_params(cvh_context *ctx, const cvh_params *p, int C)
{
  // the reference's range checks, src/main.cpp:795-830
  if (!(p->dt > 0)) return fail(ctx, CVH_ERR_ARG, "Cannot have negative or zero timestep: %f.", p->dt);
  if (p->mu < 0) return fail(ctx, CVH_ERR_ARG, "Length penalty parameter cannot be negative: %f.", p->mu);
  for (int k = 0; k < C; ++k) {
    if (p->lambda1[k] < 0) return fail(ctx, CVH_ERR_ARG, "The value of lambda1 cannot be negative.");
    if (p->lambda2[k] < 0) return fail(ctx, CVH_ERR_ARG, "The value of lambda2 cannot be negative.");
  }
  return CVH_OK;
}

// Every live context of the process (cvh_create .. cvh_destroy).  Several contexts on one GPU share its 256 MiB Infinity Cache: what one
// context's footprint suggests (write-through stores while its ping-pong pair fits the cache) is wrong when eight of them stream side by
// side -- the batch BASELINE configs[4] describes.  Round 3 left that to the caller ("wave_pol" = 0); now the automatic choice looks here.
static std::mutex g_live_mu;
static std::vector<cvh_context *> g_live;
static double live_footprint(const cvh_context *c);

// bytes per pixel-iteration pair (level-set ping-pong + planes) of every context on c's device that holds an image and a level set and
// streams beside the others ("co_resident")
static int live_contexts(const cvh_context *c)
{
  std::lock_guard<std::mutex> lk(g_live_mu);
  int k = 0;
  for (const cvh_context *o : g_live)
    if (o->device == c->device && o->co_resident && ((o->have_u && o->have_image) || o == c)) ++k;
  return k;
}

// Is this context the only co-resident one on its device?  Decided when a run's first iteration is enqueued, kept for the run.
static bool run_is_alone(const cvh_context *c)
{
  if (c->run_alone < 0 || c->enqueued == 0) c->run_alone = live_contexts(c) <= 1 ? 1 : 0;
  return c->run_alone != 0;
}

static double live_footprint(const cvh_context *c)
{
  std::lock_guard<std::mutex> lk(g_live_mu);
  double sum = 0.0;
  for (const cvh_context *o : g_live)
    if (o->device == c->device && o->co_resident && ((o->have_u && o->have_image) || o == c))
      sum += (double)o->n * (2.0 * (o->state_bits / 8) + o->C);
  return sum;
}

extern "C" void cvh_destroy(cvh_context *c)
{
  if (!c) return;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live.erase(std::remove(g_live.begin(), g_live.end(), c), g_live.end());
  }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->d_img_slab) (void)hipFree(c->d_img_slab);
  if (c->d_u_slab) (void)hipFree(c->d_u_slab);
  if (c->d_uf_slab) (void)hipFree(c->d_uf_slab);
  for (int k = 0; k < 2; ++k) if (c->d_pm[k]) (void)hipFree(c->d_pm[k]);
  if (c->d_state) (void)hipFree(c->d_state);
  if (c->h_state) (void)hipHostFree(c->h_state);
  if (c->d_partials) (void)hipFree(c->d_partials);
  if (c->d_trace) (void)hipFree(c->d_trace);
  if (c->d_mask) (void)hipFree(c->d_mask);
  if (c->d_atan) (void)hipFree(c->d_atan);
  if (c->d_dbg) (void)hipFree(c->d_dbg);
  for (int k = 0; k < 4; ++k) if (c->graphs[k].exec) (void)hipGraphExecDestroy(c->graphs[k].exec);
  if (c->pm_graph) (void)hipGraphExecDestroy(c->pm_graph);
  if (c->d_dummy) (void)hipFree(c->d_dummy);
  if (c->d_chain) (void)hipFree(c->d_chain);
  if (c->d_resident) (void)hipFree(c->d_resident);
  if (c->d_res_halo) (void)hipFree(c->d_res_halo);
  if (c->d_pm_halo) (void)hipFree(c->d_pm_halo);
  if (c->h_resident) (void)hipHostFree(c->h_resident);
  if (c->d_bounds) (void)hipFree(c->d_bounds);
  if (c->h_status) (void)hipHostFree(c->h_status);
  if (c->d_isums) (void)hipFree(c->d_isums);
  if (c->h_isums) (void)hipHostFree(c->h_isums);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (int k = 0; k < 4; ++k) if (c->evp[k]) (void)hipEventDestroy(c->evp[k]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

static int create_impl(cvh_context *c)
{
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0) c->num_cus = cus;
  }
  // one slab for the planes: the 1-pixel wave kernel fetches a group's image pieces of all channels with one instruction
  c->img_stride = (c->n + 255) & ~(size_t)255;
  HIPCHK(c, hipMalloc((void **)&c->d_img_slab, c->img_stride * c->C));
  for (int k = 0; k < c->C; ++k) c->d_img[k] = c->d_img_slab + (size_t)k * c->img_stride;
  {
    // one slab for the ping-pong pair (64 doubles of slack behind each buffer: the wave kernels park the stores of lanes
    // that own no pixel there).  Skewing the second buffer against the first by 256 B .. 1 MiB was measured: no effect.
    const size_t each = (((c->n + 64) * sizeof(double)) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    HIPCHK(c, hipMalloc((void **)&c->d_u_slab, 2 * each));
    c->d_u[0] = (double *)c->d_u_slab;
    c->d_u[1] = (double *)((char *)c->d_u_slab + each);
  }
  HIPCHK(c, hipMalloc((void **)&c->d_state, sizeof(CvhState)));
  HIPCHK(c, hipMemset(c->d_state, 0, sizeof(CvhState)));
  HIPCHK(c, hipHostMalloc((void **)&c->h_state, 4 * sizeof(CvhState), hipHostMallocDefault));
  memset(c->h_state, 0, 4 * sizeof(CvhState));
  int step_blocks = cvh_step_max_blocks(c->h, c->w);
  {
    const int wave_blocks = (((c->w + cvh_wave_cols() - 1) / cvh_wave_cols() + 3) / 4) * c->h + 1;  // strip_rows >= 1
    if (wave_blocks > step_blocks) step_blocks = wave_blocks;
  }
  {
    // atan(i/128) and pi/2 - atan(i/128), rounded once from long double
    double tab[2 * CVH_ATAN_N + CVH_ATAN2_N];
    const long double pil = 3.14159265358979323846264338327950288L;
    for (int i = 0; i < CVH_ATAN_N; ++i) {
      const long double at = atanl((long double)i / (CVH_ATAN_N - 1));
      tab[i] = (double)at;
      tab[CVH_ATAN_N + i] = (double)(pil / 2 - at);
    }
    for (int j = 0; j < CVH_ATAN2_N; ++j)  // (pi/4 + atan((j-128)/128)) / pi
      tab[2 * CVH_ATAN_N + j] = (double)((pil / 4 + atanl((long double)(j - 128) / 128)) / pil);
    HIPCHK(c, hipMalloc((void **)&c->d_atan, sizeof(tab)));
    HIPCHK(c, hipMemcpy(c->d_atan, tab, sizeof(tab), hipMemcpyHostToDevice));
  }
  const int init_blocks = cvh_init_sum_blocks(c->h, c->w);
  c->partial_rows = step_blocks > init_blocks ? step_blocks : init_blocks;
  HIPCHK(c, hipMalloc((void **)&c->d_partials, (size_t)c->partial_rows * cvh_nsums(c->C) * sizeof(double)));
  HIPCHK(c, hipMalloc((void **)&c->d_dummy, (size_t)(c->w > 64 ? c->w : 64) * sizeof(double)));
  HIPCHK(c, hipMalloc((void **)&c->d_bounds, (size_t)(c->h + 2) * sizeof(int)));
  HIPCHK(c, hipMalloc((void **)&c->d_chain, sizeof(CvhChainAcc)));
  HIPCHK(c, hipMemset(c->d_chain, 0, sizeof(CvhChainAcc)));
  HIPCHK(c, hipHostMalloc((void **)&c->h_status, 64, hipHostMallocMapped));
  c->h_status[0] = 0; c->h_status[1] = 0;
  HIPCHK(c, hipMalloc((void **)&c->d_isums, 8 * sizeof(unsigned long long)));
  HIPCHK(c, hipHostMalloc((void **)&c->h_isums, 8 * sizeof(unsigned long long), hipHostMallocDefault));
  HIPCHK(c, hipEventCreate(&c->ev0));
  HIPCHK(c, hipEventCreate(&c->ev1));
  for (int k = 0; k < 4; ++k) HIPCHK(c, hipEventCreateWithFlags(&c->evp[k], hipEventDisableTiming));
  snprintf(c->err, sizeof(c->err), "no error");
  return CVH_OK;
}

extern "C" int cvh_create(cvh_context **out, int h, int w, int channels, const cvh_params *p, int device)
{
  if (!out) return fail(nullptr, CVH_ERR_ARG, "cvh_create: out is NULL");
  *out = nullptr;
  if (h <= 0 || w <= 0) return fail(nullptr, CVH_ERR_ARG, "cvh_create: image size must be positive (got %d x %d)", h, w);
  if (channels != 1 && channels != 3)
    return fail(nullptr, CVH_ERR_ARG, "cvh_create: channels must be 1 (grayscale) or 3 (colour), got %d", channels);
  cvh_params def;
  cvh_default_params(&def);
  if (!p) p = &def;
  int rc = check_params(nullptr, p, channels);
  if (rc != CVH_OK) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, CVH_ERR_HIP, "cvh_create: no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(nullptr, CVH_ERR_ARG, "cvh_create: device %d out of range [0,%d)", device, ndev);
  cvh_context *c = new (std::nothrow) cvh_context();
  if (!c) return fail(nullptr, CVH_ERR_NOMEM, "cvh_create: out of host memory");
  c->h = h; c->w = w; c->C = channels; c->device = device; c->n = (size_t)h * w; c->p = *p;
  rc = create_impl(c);
  if (rc != CVH_OK) {
    snprintf(g_create_err, sizeof(g_create_err), "%s", c->err);
    cvh_destroy(c);
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live.push_back(c);
  }
  *out = c;
  return CVH_OK;
}

extern "C" int cvh_set_params(cvh_context *c, const cvh_params *p)
{
  if (!c || !p) return CVH_ERR_ARG;
  int rc = check_params(c, p, c->C);
  if (rc != CVH_OK) return rc;
  if (p->eps != c->p.eps) c->sums_valid = false;
  c->p = *p;
  return CVH_OK;
}

static int sync_impl(cvh_context *c);

static int ensure_f64_mirror(cvh_context *c);
static int adopt_f32_state(cvh_context *c);

extern "C" int cvh_set_option(cvh_context *c, const char *key, long value)
{
  if (!c || !key) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->timing_open || c->chain_pending) { const int rc = sync_impl(c); if (rc != CVH_OK) return rc; }  // options apply between runs
  if (!strcmp(key, "math_mode")) {
    if (value < 0 || value > 2) return fail(c, CVH_ERR_ARG, "math_mode must be 0, 1 or 2");
    c->math_mode = (int)value;
  } else if (!strcmp(key, "finalize")) {
    if (value != 0 && value != 1) return fail(c, CVH_ERR_ARG, "finalize must be 0 or 1");
    c->finalize_mode = (int)value;
  } else if (!strcmp(key, "tile_rows")) {
    if (value != 0 && value != 12 && value != 14 && value != 16) return fail(c, CVH_ERR_ARG, "tile_rows must be 0 (auto), 12, 14 or 16");
    c->tile_rows = (int)value;
  } else if (!strcmp(key, "kernel")) {
    if (value < -1 || value > 3 || value == 1)      // (1 was the strip kernel of round 1: never chosen, no fallback -- tools/experiments/pruned_flavours/)
      return fail(c, CVH_ERR_ARG, "kernel must be -1 (auto), 0 (tile), 2 (wave) or 3 (wave, 2 pixels per lane)");
    c->kernel = (int)value;
  } else if (!strcmp(key, "pm_kernel")) {
    if (value < -1 || value > 4 || value == 2)     // (2 was a 2-pixel-per-lane 1-step kernel: never chosen -- tools/experiments/pruned_flavours/)
      return fail(c, CVH_ERR_ARG, "pm_kernel must be -1 (auto), 0 (tile), 1 (wave), 3 (wave, 2 time steps per launch) or 4 (resident plane)");
    c->pm_kernel = (int)value;
  } else if (!strcmp(key, "res_straight")) {
    c->res_straight = value ? 1 : 0;
  } else if (!strcmp(key, "pm_strip_rows")) {
    if (value < 0) return fail(c, CVH_ERR_ARG, "pm_strip_rows must be >= 0");
    c->pm_strip_rows = (int)value;
  } else if (!strcmp(key, "wave_occupancy")) {
    if (value < 3 || value > 5) return fail(c, CVH_ERR_ARG, "wave_occupancy must be 3..5 (more waves per SIMD would spill registers)");
    c->wave_minw = (int)value;
  } else if (!strcmp(key, "debug_times")) {
    // diagnostic: per-wave start/end stamps of the wave kernel, read back with cvh_debug_read
    if (c->d_dbg) { HIPCHK(c, hipFree(c->d_dbg)); c->d_dbg = nullptr; c->dbg_words = 0; }
    if (value > 0) {
      c->dbg_words = (size_t)c->partial_rows * 20 + 16;
      if (c->dbg_words < (size_t)CVH_RESIDENT_MAX_TILES * 12 + 64) c->dbg_words = (size_t)CVH_RESIDENT_MAX_TILES * 12 + 64;   // resident kernel: 12 words per tile + 64 of the master
      HIPCHK(c, hipMalloc((void **)&c->d_dbg, c->dbg_words * 8));
      HIPCHK(c, hipMemset(c->d_dbg, 0, c->dbg_words * 8));
    }
  } else if (!strcmp(key, "graph")) {
    c->use_graph = value != 0;
  } else if (!strcmp(key, "wave_xcd")) {
    c->wave_xcd = value != 0;
  } else if (!strcmp(key, "wave_rev")) {
    c->wave_rev = value != 0;
  } else if (!strcmp(key, "wave_skew")) {
    if (value < 0 || value > 500) return fail(c, CVH_ERR_ARG, "wave_skew must be 0..500 (per mille)");
    c->wave_skew = (int)value;
  } else if (!strcmp(key, "chain")) {
    c->chain_opt = value != 0;
  } else if (!strcmp(key, "resident")) {
    if (value < -1 || value > 1) return fail(c, CVH_ERR_ARG, "resident must be -1 (auto), 0 or 1");
    c->resident_opt = (int)value;
  } else if (!strcmp(key, "far_terms")) {
    if (value != 4 && value != 5) return fail(c, CVH_ERR_ARG, "far_terms must be 4 or 5");
    c->far_terms = (int)value;
  } else if (!strcmp(key, "wave_pol")) {
    if (value < -1 || value > 2) return fail(c, CVH_ERR_ARG, "wave_pol must be -1 (auto), 0, 1 or 2 (diagnostic: non-temporal loads)");
    c->wave_pol = (int)value;
  } else if (!strcmp(key, "wave_cls")) {
    if (value < 0 || value > 2) return fail(c, CVH_ERR_ARG, "wave_cls must be 0 (off), 1 (2-pixel kernel) or 2 (1-pixel kernel too)");
    c->wave_cls = (int)value;
  } else if (!strcmp(key, "wave_cskew")) {
    if (value < 0 || value > 900) return fail(c, CVH_ERR_ARG, "wave_cskew must be 0..900 (per mille)");
    c->wave_cskew = (int)value;
  } else if (!strcmp(key, "wave_depth")) {
    if (value != 4 && value != 8) return fail(c, CVH_ERR_ARG, "wave_depth must be 4 or 8");
    c->wave_depth = (int)value;
  } else if (!strcmp(key, "wave_imgv")) {
    c->wave_imgv = value != 0;
  } else if (!strcmp(key, "wave_sync")) {
    c->wave_sync = value < 0 ? -1 : (value != 0);
  } else if (!strcmp(key, "near_switch")) {
    c->near_switch = value != 0;
  } else if (!strcmp(key, "res_prio")) {
    c->res_prio = value != 0;
  } else if (!strcmp(key, "res_go_share")) {
    if (value < 0 || value > 6) return fail(c, CVH_ERR_ARG, "res_go_share must be 0 .. 6");
    c->res_go_share = (int)value;
  } else if (!strcmp(key, "co_resident")) {
    c->co_resident = value != 0;
  } else if (!strcmp(key, "state")) {
    // 64 (default): the level set lives in HBM as double, the reference's CV_64FC1 (src/main.cpp:225) -- the parity mode.
    // 32: DECLARED fast mode -- float in HBM (9 instead of 17 bytes per pixel-iteration), arithmetic and sums unchanged; every new value is
    // rounded to float.  2-pixel wave kernel only (FAST arithmetic, width a multiple of 16 and >= 144, < 2^28 pixels).
    if (value != 32 && value != 64) return fail(c, CVH_ERR_ARG, "state must be 64 or 32");
    if ((int)value != c->state_bits) {
      if (value == 32) {
        if (c->w % 16 != 0 || c->w < 144 || c->n >= ((size_t)1 << 28))
          return fail(c, CVH_ERR_ARG, "state 32 needs a width that is a multiple of 16 and >= 144, and fewer than 2^28 pixels (the 2-pixel wave kernel)");
        if (c->have_u) { const int rc = ensure_f64_mirror(c); if (rc != CVH_OK) return rc; }
        c->state_bits = 32;
        if (c->have_u) { const int rc = adopt_f32_state(c); if (rc != CVH_OK) return rc; }
      } else {
        if (c->have_u) { const int rc = ensure_f64_mirror(c); if (rc != CVH_OK) return rc; }
        c->state_bits = 64;
        c->mirror_valid = true;
      }
      c->sums_valid = false;
    }
  } else if (!strcmp(key, "wave_prio")) {
    if (value < 0 || value > 4) return fail(c, CVH_ERR_ARG, "wave_prio must be 0..4");
    c->wave_prio = (int)value;
  } else if (!strcmp(key, "wave_lds_cap")) {
    c->wave_lds_cap = value != 0;
  } else if (!strcmp(key, "strip_rows")) {
    if (value < 0) return fail(c, CVH_ERR_ARG, "strip_rows must be >= 0");
    c->strip_rows = (int)value;
  } else if (!strcmp(key, "strips")) {
    if (value < 0 || value > c->h) return fail(c, CVH_ERR_ARG, "strips must be 0 (auto) .. h");
    c->strips = (int)value;
  } else if (!strcmp(key, "lut")) {
    c->use_lut = value != 0;
  } else if (!strcmp(key, "dma")) {
    c->use_dma = value != 0;
  } else if (!strcmp(key, "sync_every")) {
    if (value < 1) return fail(c, CVH_ERR_ARG, "sync_every must be >= 1");
    c->sync_every = (int)(value > 1000000 ? 1000000 : value);
  } else if (!strcmp(key, "trace")) {
    if (value < 0) return fail(c, CVH_ERR_ARG, "trace capacity must be >= 0");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_trace) { HIPCHK(c, hipFree(c->d_trace)); c->d_trace = nullptr; }
    c->trace_cap = 0;
    if (value > 0) {
      const size_t bytes = (size_t)value * (2 * c->C + 1) * sizeof(double);
      HIPCHK(c, hipMalloc((void **)&c->d_trace, bytes));
      HIPCHK(c, hipMemset(c->d_trace, 0, bytes));
      c->trace_cap = (int)value;
    }
  } else {
    return fail(c, CVH_ERR_ARG, "unknown option \"%s\"", key);
  }
  return CVH_OK;
}

// tol-free part of the stop condition, src/main.cpp:950-959 (zero-initialised accumulator, channels added serially in k,
// scaled by 1/C, L2 norm with four squares per step added left to right).  (sum_k I_k)/C squared takes one of 255 C + 1
// values: the table keeps the reference's rounding and summation order while the loop is integer adds and lookups.
static double stop_norm_host(const std::vector<const uint8_t *> &planes, size_t n)
{
  const int C = (int)planes.size();
  const double inv = 1.0 / C;
  double sq[CVH_MAX_CHANNELS * 255 + 1];
  for (int t = 0; t <= 255 * C; ++t) { const double v = (double)t * inv; sq[t] = v * v; }
  auto at = [&](size_t q) {
    int t = 0;
    for (int k = 0; k < C; ++k) t += planes[k][q];
    return sq[t];
  };
  double s = 0;
  size_t i = 0;
  for (; i + 4 <= n; i += 4) s += ((at(i) + at(i + 1)) + at(i + 2)) + at(i + 3);
  for (; i < n; ++i) s += at(i);
  return sqrt(s);
}

// Sums of the planes resident on the device: sum(I_k) for the region means and, for one channel, the stop norm (exact
// integers, image_sums_kernel).  Three channels round (sum_k I_k)/3 per pixel, so their norm needs the reference's serial
// order: `host_planes` (the caller's buffers, or nullptr to fetch the planes) feed stop_norm_host.
static int image_stats(cvh_context *c, const uint8_t *const *host_planes)
{
  HIPCHK(c, hipMemsetAsync(c->d_isums, 0, 8 * sizeof(unsigned long long), c->stream));
  HIPCHK(c, cvh_launch_image_sums(c->d_img, c->C, c->n, c->d_isums, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_isums, c->d_isums, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  const bool exact_on_device = c->C == 1 && c->n < ((size_t)1 << 36);   // 2^36 * 255^2 < 2^53
  std::vector<std::vector<uint8_t>> fetched;
  std::vector<const uint8_t *> pl;
  if (!exact_on_device) {
    if (host_planes) pl.assign(host_planes, host_planes + c->C);
    else {
      try { fetched.assign(c->C, std::vector<uint8_t>(c->n)); } catch (...) { return fail(c, CVH_ERR_NOMEM, "out of host memory"); }
      for (int k = 0; k < c->C; ++k) {
        HIPCHK(c, hipMemcpyAsync(fetched[k].data(), c->d_img[k], c->n, hipMemcpyDeviceToHost, c->stream));
        pl.push_back(fetched[k].data());
      }
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < c->C; ++k) c->sum_img[k] = (double)c->h_isums[2 * k];   // exact: < 2^53
  c->stop_norm = exact_on_device ? sqrt((double)c->h_isums[1]) : stop_norm_host(pl, c->n);
  c->stop_valid = true;
  return CVH_OK;
}

extern "C" int cvh_set_image(cvh_context *c, const uint8_t *const *planes)
{
  if (!c || !planes) return CVH_ERR_ARG;
  for (int k = 0; k < c->C; ++k) if (!planes[k]) return fail(c, CVH_ERR_ARG, "cvh_set_image: plane %d is NULL", k);
  HIPCHK(c, hipSetDevice(c->device));
  if (c->timing_open || c->chain_pending) { const int rc = sync_impl(c); if (rc != CVH_OK) return rc; }
  for (int k = 0; k < c->C; ++k)
    HIPCHK(c, hipMemcpyAsync(c->d_img[k], planes[k], c->n, hipMemcpyHostToDevice, c->stream));
  const int rc = image_stats(c, planes);
  if (rc != CVH_OK) return rc;
  c->have_image = true;
  c->sums_valid = false;
  return CVH_OK;
}

extern "C" int cvh_get_image(cvh_context *c, uint8_t *const *planes)
{
  if (!c || !planes) return CVH_ERR_ARG;
  if (!c->have_image) return fail(c, CVH_ERR_STATE, "cvh_get_image: no image set");
  HIPCHK(c, hipSetDevice(c->device));
  for (int k = 0; k < c->C; ++k) {
    if (!planes[k]) return fail(c, CVH_ERR_ARG, "cvh_get_image: plane %d is NULL", k);
    HIPCHK(c, hipMemcpyAsync(planes[k], c->d_img[k], c->n, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CVH_OK;
}

static int current_buffer(const cvh_context *c) { return (c->cur_base + c->steps_done) & 1; }

// FP32 state: the float buffers (lazily allocated) take over the level set that d_u[current] holds -- rounded to float, and d_u[current]
// is rewritten with the rounded values, so that whatever reads the mirror (initial sums, mask, get) sees what the kernels iterate on.
static int adopt_f32_state(cvh_context *c)
{
  if (!c->d_uf_slab) {
    const size_t each = (((c->n + 64) * sizeof(float)) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    HIPCHK(c, hipMalloc((void **)&c->d_uf_slab, 2 * each));
    c->d_uf[0] = (float *)c->d_uf_slab;
    c->d_uf[1] = (float *)((char *)c->d_uf_slab + each);
  }
  const int cur = current_buffer(c);
  HIPCHK(c, cvh_launch_state_narrow(c->d_u[cur], c->d_uf[cur], c->n, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->mirror_valid = true;
  c->sums_valid = false;
  return CVH_OK;
}

// FP32 state: d_u[current] = (double) d_uf[current] if launches have run since the mirror was last refreshed (call behind a sync).
static int ensure_f64_mirror(cvh_context *c)
{
  if (c->state_bits != 32 || c->mirror_valid) return CVH_OK;
  if (c->timing_open || c->chain_pending) { const int rc = sync_impl(c); if (rc != CVH_OK) return rc; }
  const int cur = current_buffer(c);
  HIPCHK(c, cvh_launch_state_widen(c->d_uf[cur], c->d_u[cur], c->n, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->mirror_valid = true;
  return CVH_OK;
}

static int reset_run_impl(cvh_context *c)
{
  if (c->timing_open || c->chain_pending) { const int rc = sync_impl(c); if (rc != CVH_OK) return rc; }
  // new run: counter and stop flag cleared; the buffer holding u becomes the base, and so does the chain-mode sum set
  // that belongs to it (the set after it may hold the sums of an iteration computed past a stop: cleared)
  c->cur_base = current_buffer(c);
  c->chain_pb = (c->chain_pb + c->steps_done) & 3;
  c->steps_done = 0;
  c->enqueued = 0;
  c->run_pol = -1; c->run_alone = -1; c->run_chunk = -1;   // the automatic choices of a run are taken again (live-context registry)
  static const int zeros[4] = {0, 0, 0, 0};   // steps_done, stopped, ticket, pending
  HIPCHK(c, hipMemcpyAsync(&c->d_state->steps_done, zeros, sizeof(zeros), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(&c->d_chain->v[(c->chain_pb + 1) & 3][0], 0, sizeof(c->d_chain->v[0]), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->h_status[0] = 0; c->h_status[1] = 0;
  return CVH_OK;
}

extern "C" int cvh_reset_run(cvh_context *c)
{
  if (!c) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  return reset_run_impl(c);
}

extern "C" int cvh_set_levelset(cvh_context *c, const double *u)
{
  if (!c || !u) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->timing_open || c->chain_pending) { const int rc = sync_impl(c); if (rc != CVH_OK) return rc; }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // the new level set goes into the buffer whose parity equals the chain-mode sum set's (chain_pb & 1): the ping-pong parity
  // and the sum-set phase of a launch then stay locked together, and a cached step graph of a phase is valid for every run
  c->cur_base = c->chain_pb & 1; c->steps_done = 0; c->enqueued = 0;
  HIPCHK(c, hipMemcpyAsync(c->d_u[c->cur_base], u, c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_u = true;
  c->sums_valid = false;
  c->mirror_valid = true;
  if (c->state_bits == 32) { const int rc = adopt_f32_state(c); if (rc != CVH_OK) return rc; }
  return reset_run_impl(c);
}

extern "C" void cvh_levelset_checkerboard_host(int h, int w, double *u)
{
  // src/main.cpp:226-231: sign(sin(pi*i/5) * sin(pi*j/5)), host libm, double
  const double pi = 3.14159265358979323846;
  std::vector<double> sj((size_t)w);
  for (int j = 0; j < w; ++j) sj[j] = sin(pi * j / 5);
  for (int i = 0; i < h; ++i) {
    const double si = sin(pi * i / 5);
    for (int j = 0; j < w; ++j) {
      const double z = si * sj[j];
      u[(size_t)i * w + j] = (z == 0) ? 0.0 : (z < 0 ? -1.0 : 1.0);
    }
  }
}

extern "C" int cvh_init_checkerboard(cvh_context *c)
{
  if (!c) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->timing_open || c->chain_pending) { const int rc = sync_impl(c); if (rc != CVH_OK) return rc; }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // the h + w sine factors from the host's libm (as cvh_levelset_checkerboard_host), staged in the idle buffer of the
  // ping-pong pair (h + w <= h w + 1 doubles); the sign of their product is taken on the device
  std::vector<double> sv((size_t)c->h + c->w);
  const double pi = 3.14159265358979323846;
  for (int i = 0; i < c->h; ++i) sv[i] = sin(pi * i / 5);
  for (int j = 0; j < c->w; ++j) sv[(size_t)c->h + j] = sin(pi * j / 5);
  c->cur_base = c->chain_pb & 1; c->steps_done = 0; c->enqueued = 0;   // see cvh_set_levelset
  HIPCHK(c, hipMemcpyAsync(c->d_u[c->cur_base ^ 1], sv.data(), sv.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, cvh_launch_checkerboard(c->d_u[c->cur_base ^ 1], c->d_u[c->cur_base], c->h, c->w, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_u = true;
  c->sums_valid = false;
  c->mirror_valid = true;
  if (c->state_bits == 32) { const int rc = adopt_f32_state(c); if (rc != CVH_OK) return rc; }
  return reset_run_impl(c);
}

extern "C" int cvh_get_levelset(cvh_context *c, double *u)
{
  if (!c || !u) return CVH_ERR_ARG;
  if (!c->have_u) return fail(c, CVH_ERR_STATE, "cvh_get_levelset: no level set");
  HIPCHK(c, hipSetDevice(c->device));
  { const int rc = ensure_f64_mirror(c); if (rc != CVH_OK) return rc; }
  HIPCHK(c, hipMemcpyAsync(u, c->d_u[current_buffer(c)], c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CVH_OK;
}

struct Geometry { int strip; int rows; int tiles_x, tiles_y, strip_rows, nblocks; };

// Which step kernel runs and on what grid (g.strip: 0 tile kernel, 2 wave kernel, 3 wave kernel with 2 pixels per lane).
static Geometry resolve_geometry(const cvh_context *c)
{
  Geometry g;
  g.strip = 0;
  // default: the wave kernel (any width; fastest measured); it addresses the level set through
  // buffer instructions with 32-bit byte offsets and marks dropped lanes with offset 2^31, so
  // images of 2^28 pixels (2 GiB of level set) or more use the tile kernel
  // auto: the 2-pixel kernel from 0.6 Mpixel up (end of round 2, one context per size, 2-pixel with its cache policy chosen by
  // footprint vs 1-pixel: 3000x4000 40.3 vs 43.8 us, 4096^2 59.4 vs 64.7, 4608^2 80.5 vs 87.2, 5120^2 93.1 vs 97.2, 6144^2 133.0 vs
  // 140.7, 4320x7680 122.7 vs 123.2, 8192^2 243.3 vs 245.5)
  // three channels: the 2-pixel flavour exists in FAST arithmetic only; with equal strips (no class skew) it is the faster one from
  // round 3 on (4096^2 x 3, one context, alternating: 73.3 vs 74.9 us; bench lines of one session: 72.9 / 74.3 vs 74.5 / 76.1 us)
  const bool two_px_c3 = c->C == 3 && use_fast(c) && (c->kernel == 3 || c->kernel == -1 || c->state_bits == 32);
  if ((c->kernel == 3 || c->state_bits == 32 || (c->kernel == -1 && c->n >= (size_t)600000)) && (c->C == 1 || two_px_c3) && c->w % 16 == 0 &&
      c->w >= 144 && c->n < ((size_t)1 << 28)) {
    // wave kernel with 2 pixels per lane: 126 output columns per wave; workgroup = 2 wave-columns x 2 strips;
    // one round of resident waves (3 or 4 per SIMD)
    g.strip = 3;
    g.rows = 4;
    g.tiles_x = (c->w + cvh_wave2_cols() - 1) / cvh_wave2_cols();
    const int nbc = (g.tiles_x + 1) / 2;
    int sr = c->strip_rows, small_exact = 0;
    if (sr <= 0) {
      const int occ = use_fast(c) ? (c->wave_minw == 4 ? 4 : 3) : 2;   // as compiled: cvh_launch_wave2
      int nstrips = 2 * ((c->num_cus * occ) / nbc);
      // small planes (a full round would mean strips of < 13 rows: 3 halo rows and a pipeline fill each): ~1.8 workgroups
      // per CU instead -- measured at 2048^2: 16 rows 23.2, 18 rows 24.1, 20 rows 21.4, 22 rows 22.7, 24 rows 22.8 us
      // (round 3, exact strip counts at 2048^2, one context: 56 strips 22.8 us, 84 21.2, 100 21.1, 104 20.8, 108 21.2, 112 20.7, 114 22.1 --
      // one workgroup more than two per CU --, 128 21.4, 140 21.1, 168 21.6: flat from 84 to 168 except just above a multiple of the CU
      // count; TWO workgroups per CU, never more)
      bool exact = false;
      if (nstrips > 160) { nstrips = 2 * ((2 * c->num_cus) / nbc); if (nstrips < 2) nstrips = 2; exact = true; }
      if (nstrips < 1) nstrips = 1;
      sr = (c->h + nstrips - 1) / nstrips;
      if (sr < 8) { sr = 8; exact = false; }
      if (exact && c->wave_cls && c->wave_xcd) small_exact = nstrips;   // the class-major table deals rows by weight: any count works
    }
    g.strip_rows = sr;
    g.tiles_y = small_exact ? small_exact : (c->h + sr - 1) / sr;
    if (c->strips > 0 && c->strip_rows <= 0 && c->wave_cls && c->wave_xcd) {   // exact count ("strips"): the class-major table deals rows by weight
      g.tiles_y = c->strips;
      g.strip_rows = (c->h + c->strips - 1) / c->strips;
      if (g.strip_rows < 8) { g.strip_rows = 8; g.tiles_y = (c->h + 7) / 8; }
    }
    g.nblocks = nbc * ((g.tiles_y + 1) / 2);
    return g;
  }
  if ((c->kernel == 2 || c->kernel == 3 || c->kernel == -1) && c->n < ((size_t)1 << 28)) {
    // wave kernel: 63 output columns per wave, strip_rows rows per wave, 4 waves per workgroup;
    // one round of resident waves (wave_minw per SIMD)
    g.strip = 2;
    g.rows = 4;
    g.tiles_x = (c->w + cvh_wave_cols() - 1) / cvh_wave_cols();
    int sr = c->strip_rows;
    if (sr <= 0) {
      // waves per SIMD the kernel flavour is compiled for (csv_wave_kernel.hip, launch_wave_c)
      const int occ = use_fast(c) ? (c->C == 3 ? 3 : c->wave_minw) : (c->C == 3 ? 2 : 3);
      int nstrips = (c->num_cus * occ) / ((g.tiles_x + 3) / 4);
      // Every strip re-reads 3 halo rows and fills its pipeline once: measured on MI355X (512^2 ..
      // 4096^2, tools/size_sweep.sh) a full round of resident waves is best at 4096^2 (75 strips) and
      // 64 strips wherever residency would allow many more (smaller images).
      if (nstrips > 80) nstrips = 64;
      if (nstrips < 1) nstrips = 1;
      sr = (c->h + nstrips - 1) / nstrips;
      if (sr < 8) sr = 8;  // shorter strips only pay prologue overhead
    }
    g.strip_rows = sr;
    g.tiles_y = (c->h + sr - 1) / sr;
    g.nblocks = ((g.tiles_x + 3) / 4) * g.tiles_y;  // 4 adjacent wave-columns per workgroup
    return g;
  }
  g.rows = c->tile_rows == 16 ? 16 : 14;  // auto = 14: keeps 4 workgroups per CU beside the tables
  cvh_step_grid(c->h, c->w, g.rows, &g.tiles_x, &g.tiles_y);
  g.strip_rows = g.rows;
  g.nblocks = g.tiles_x * g.tiles_y;
  return g;
}

static bool use_chain(const cvh_context *c, const Geometry &g)
{
  return (g.strip == 3 || g.strip == 2) && use_fast(c) && c->finalize_mode == 0 && c->chain_opt;
}

// Resident mode (csv_resident_kernel.hip): the plane is cut into tr x tc tiles of <= 128 x 128 pixels, one workgroup per tile, all
// co-resident (one per CU), the level set stays in LDS for a chunk of iterations.  Applies to 1 channel, FAST arithmetic, chain-mode
// sums, even widths, and planes that fit: tiles <= what the device holds, every tile 16 .. 128 rows.
struct ResidentGeom { int tr, tc, band; };
// tiles_y x tiles_x tiles of <= 128 rows x 128 columns, at most one per CU
static bool resident_tiles(const cvh_context *c, int cap_blocks, ResidentGeom *rg)
{
  const int tw = cvh_resident_tile_w(), thmax = cvh_resident_tile_hmax();
  const int tc = (c->w + tw - 1) / tw;
  int cap = cap_blocks < CVH_RESIDENT_MAX_TILES ? cap_blocks : CVH_RESIDENT_MAX_TILES;
  if (cap > c->num_cus) cap = c->num_cus;                      // one workgroup per CU: a second one on a CU would wait for its slot
  int tr = cap / tc;
  if (tr < 1) return false;
  if (tr > c->h / 16) tr = c->h / 16;                          // tiles of >= 16 rows (every wave's band >= 2 rows)
  if ((c->h + tr - 1) / tr > thmax) return false;              // does not fit the LDS of the CUs
  rg->tr = tr; rg->tc = tc; rg->band = 0;
  return true;
}

static bool resident_geometry(cvh_context *c, ResidentGeom *rg)
{
  if (!c->resident_opt || c->C != 1 || !use_fast(c) || !c->chain_opt || c->finalize_mode != 0 || (c->w & 1) || c->w < 16 || c->h < 16) return false;
  if (c->state_bits == 32) return false;      // the FP32-state mode is the 2-pixel per-launch kernel's
  if (!(c->kernel == -1 || c->kernel == 2 || c->kernel == 3)) return false;
  // auto: a caller who chose a per-launch data flow or tuned its geometry / launch path gets that flow (measured, one context per size,
  // resident vs per-launch: 128^2 7.4 vs 7.6 us, 256^2 6.9 vs 7.6, 768^2 8.5 vs 9.8, 1024x2048 11.5 vs 14.6, 1536^2 12.4 vs 16.2,
  // 1200x1920 12.0 vs 15.4, 2048^2 16.0 vs 20.9: ahead at every size that fits)
  if (c->resident_opt < 0 && (c->kernel != -1 || c->strip_rows != 0 || c->strips != 0 || !c->use_graph)) return false;
  // auto also steps aside when other contexts stream on this GPU (a batch): cooperative launches of different contexts serialise and cost
  // ~25 us each, while interleaved per-launch flows fill each other's gaps -- measured, eight images interleaved in chunks of 8 iterations
  // (tools/batch_probe.py, gpurun_out/r4s9): 2048^2 32.6 us per image-iteration resident vs 16.2 per launch (17.3 with chunks of 50);
  // 1024^2 22.3 vs 6.1 (10.6).  Decided when a run's first iteration is enqueued, kept for the run.
  // End of round 4, with the resident kernel a quarter faster (12.1 us per iteration at 2048^2): a batch of LARGE planes whose runs are enqueued in
  // LONG chunks is better off with one cooperative launch after the other -- eight planes, us per image-iteration, per-launch interleaved vs
  // resident in chunks of 50 / 100 / 400 (gpurun_out/r4s61, r4s62): 2048^2 16.2 vs 14.7 / 13.2 / 12.1; 1792^2 13.4 vs 13.8 / 12.5 / 11.6;
  // 1536^2 11.0 vs 12.2 / 10.8 / 9.9; 1280^2 8.2 vs 10.5 / 9.2 / 8.3; 1024^2 5.9 vs 8.2 / 6.9 / 6.1.  So in a batch an ENQUEUE takes the resident flow
  // when it is long enough for the plane's size (cvh_run: chunks of up to 1024 iterations) -- per enqueue, not per run: a long warm-up chunk
  // followed by chunks of 8 must not leave a run with 8-iteration cooperative launches (29 us per image-iteration at 2048^2).  The two flows
  // continue each other on one context (sum sets, stop rule, trace); their level sets agree to <= 1e-9, not bit for bit -- a caller who needs
  // the same bits whatever the chunking sets "resident" itself.
  if (c->resident_opt < 0 && !run_is_alone(c)) {
    const double px = (double)c->h * (double)c->w;
    const int need = px >= 3.6e6 ? 48 : px >= 2.9e6 ? 72 : px >= 2.2e6 ? 100 : INT_MAX;
    if (c->run_chunk < need) return false;
  }
  if (c->resident_cap < 0) {
    int coop = 0;
    c->resident_cap = 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, c->device) == hipSuccess && coop)
      c->resident_cap = cvh_resident_blocks_per_cu() * c->num_cus;
  }
  if (c->resident_cap <= 0) return false;
  return resident_tiles(c, c->resident_cap, rg);
}

// Perona-Malik on a resident plane (pm_resident_kernel.hip): any channel count (the planes are smoothed one after the other), both
// arithmetic flavours; the same tiles as the CSV kernel.
static bool pm_resident_geometry(cvh_context *c, ResidentGeom *rg)
{
  if ((c->w & 1) || c->w < 16 || c->h < 16) return false;
  if (c->pm_resident_cap < 0) {
    int coop = 0;
    c->pm_resident_cap = 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, c->device) == hipSuccess && coop)
      c->pm_resident_cap = cvh_pm_resident_blocks_per_cu() * c->num_cus;
  }
  if (c->pm_resident_cap <= 0) return false;
  int cap = c->pm_resident_cap < CVH_RESIDENT_MAX_TILES ? c->pm_resident_cap : CVH_RESIDENT_MAX_TILES;
  if (cap > c->num_cus) cap = c->num_cus;                      // one workgroup per CU
  const int tc = (c->w + cvh_resident_tile_w() - 1) / cvh_resident_tile_w();
  // tiles of 8 x band rows x 128 columns, every wave a band of exactly 2, 4, 8 or 16 rows: the shortest bands whose tiles the CUs hold at once
  // (more CUs at work); the last tile row of the image may be shorter, but holds at least the two rows a border piece needs
  for (int nr = 2; nr <= 16; nr *= 2) {
    const int th = 8 * nr, tr = (c->h + th - 1) / th;
    if (tr * tc > cap) continue;
    if (c->h - (tr - 1) * th < 2) continue;
    rg->tr = tr; rg->tc = tc; rg->band = nr;
    return true;
  }
  return false;
}

// `step` = index of the launch inside the run (c->enqueued when it is enqueued): selects the chain-mode sum set
static void fill_args(const cvh_context *c, CvhStepArgs *a, int in_buf, int step)
{
  memset(a, 0, sizeof(*a));
  a->u_in = c->d_u[in_buf];
  a->u_out = c->d_u[in_buf ^ 1];
  a->state32 = 0;
  if (c->state_bits == 32) {   // FP32 state: the step kernel's pair are the float buffers (prepare() points the initial sums at the mirror)
    a->u_in = reinterpret_cast<const double *>(c->d_uf[in_buf]);
    a->u_out = reinterpret_cast<double *>(c->d_uf[in_buf ^ 1]);
    a->state32 = 1;
  }
  for (int k = 0; k < c->C; ++k) a->img[k] = c->d_img[k];
  a->img_stride = (unsigned)c->img_stride;
  a->st = c->d_state;
  a->partials = c->d_partials;
  a->trace = c->d_trace;
  a->trace_cap = c->trace_cap;
  a->h = c->h; a->w = c->w;
  Geometry g = resolve_geometry(c);
  a->tiles_x = g.tiles_x; a->tiles_y = g.tiles_y;
  a->nparts = g.nblocks;
  a->tile_rows = g.rows;
  a->strip_rows = g.strip_rows;
  a->fused_finalize = c->finalize_mode == 0;
  // src/main.cpp:985: dt * (mu*kappa - nu + u_diff/N) evaluates as one addWeighted
  a->alpha = c->p.mu * c->p.dt;
  a->beta = (1.0 / c->C) * c->p.dt;
  a->gamma = -c->p.nu * c->p.dt;
  a->eps = c->p.eps;
  for (int k = 0; k < CVH_MAX_CHANNELS; ++k) { a->lambda1[k] = c->p.lambda1[k]; a->lambda2[k] = c->p.lambda2[k]; }
  const double pi = 3.14159265358979323846;
  a->atan_tab = c->d_atan;
  a->atan2_tab = c->d_atan + 2 * CVH_ATAN_N;
  a->wave_minw = c->wave_minw;
  a->wave_lds_cap = c->wave_lds_cap;
  a->wave_prio = c->wave_prio;
  // workgroup barrier per group of four rows: keeps a workgroup's waves on neighbouring rows (cache locality) -- worth it for one channel;
  // with three channels the barrier costs more than the locality returns (4096^2 x 3, one context: 73.0-74.4 -> 71.9-73.2 us)
  a->wave_sync = c->wave_sync >= 0 ? c->wave_sync : (c->C == 3 ? 0 : 1);
  a->near_switch = c->near_switch;
  a->res_prio = c->res_prio;
  a->res_go_shift = c->res_go_share;
  a->wave_depth = c->wave_depth;
  a->wave_imgv = c->wave_imgv;
  a->dummy = c->d_dummy;
  a->strip_bounds = c->d_bounds;
  a->wave_rev = c->wave_rev;
  a->wave_xcd = c->wave_xcd;
  if (use_chain(c, g)) {
    a->chain = c->d_chain;
    a->chain_pb = c->chain_pb;
    a->chain_phase = (c->chain_pb + step) & 3;
    a->chain_s4 = c->d_partials;   // [2][nparts] rows of sum u_diff^2 (the workspace holds far more)
    // |sum (H - 1/2)| <= N/2 and |sum I (H - 1/2)| <= 255 N / 2 for every subset of pixels: 62 - ceil(log2(bound + 1)) fraction bits
    const double bound[4] = {0.5 * (double)c->n, 127.5 * (double)c->n, 127.5 * (double)c->n, 127.5 * (double)c->n};
    for (int k = 0; k < 4; ++k) {
      int e = 0;
      while (ldexp(1.0, e) < bound[k] + 1.0) ++e;
      a->chain_scale[k] = ldexp(1.0, 62 - e);
      a->chain_inv[k] = ldexp(1.0, e - 62);
    }
  }
  // class-major numbering + class skew: the 2-pixel kernel by default (measured there: -2.3 us at 4096^2); the 1-pixel kernel only
  // on request ("wave_cls" = 2): measured neutral to slightly worse there (4096^2 x 3 channels: 77.2 plain, 77.3 class-major, 82.5 with
  // skew 500; 1 channel: 63.6 / 64.3)
  // write-through stores pay while the ping-pong pair and the planes (mostly) fit the 256 MiB Infinity Cache: up to ~300 MB of footprint
  if (c->wave_pol >= 0) a->wave_pol = c->wave_pol;
  else {   // auto, per run: taken (and re-taken, while nothing of the run is enqueued) from the device's live footprint, then kept
    if (c->run_pol < 0 || c->enqueued == 0) c->run_pol = live_footprint(c) <= 300e6 ? 1 : 0;
    a->wave_pol = c->run_pol;
  }
  a->wave_cls = (((g.strip == 3 && c->wave_cls) || (g.strip == 2 && c->wave_cls == 2)) && c->wave_xcd) ? (c->num_cus >= 8 ? c->num_cus / 8 : 1) : 0;
  a->host_status = c->h_status;
  a->dbg_times = c->d_dbg;
  a->inv_eps = 1.0 / c->p.eps;
  a->dk1 = pi / c->p.eps;
  a->dk2 = pi * c->p.eps;
  {
    const double e = c->p.eps, e2 = e * e;
    a->far_k[0] = e / pi; a->far_k[1] = -(e * e2) / (3.0 * pi);
    a->far_k[2] = (e * e2 * e2) / (5.0 * pi); a->far_k[3] = -(e * e2 * e2 * e2) / (7.0 * pi);
    // 5 terms (through t^9/9, t = eps/|u|): next term t^11/11 <= 2.5e-18 for |u| >= 32 eps.  With 4 terms the threshold
    // is 64 eps ("far_terms" = 4: far_k[4] = 0) -- the 4096^2 checkerboard run then spends iterations 3..13 in the near field
    // (|u| grows from 36 to 64 there), with 5 terms only iterations 1..2.
    a->far_k[4] = c->far_terms == 5 ? (e * e2 * e2 * e2 * e2) / (9.0 * pi) : 0.0;
    a->far_thr = (c->far_terms == 5 ? 32.0 : 64.0) * e;
  }
  a->stop_cond = c->stop_cond_h;
  a->npix = (double)c->n;
  for (int k = 0; k < CVH_MAX_CHANNELS; ++k) a->sum_img[k] = c->sum_img[k];
  a->derive_complement = use_fast(c) ? (g.strip >= 2 ? 2 : 1) : 0;  // 2: the wave kernels sum H - 1/2
  a->use_lut = c->use_lut;
  a->use_dma = c->use_dma;
}

// Host part of prepare(): the tol-free stop norm of planes that changed on the device.
static int prepare_host(cvh_context *c)
{
  if (!c->have_image) return fail(c, CVH_ERR_STATE, "no image set (call cvh_set_image first)");
  if (!c->have_u) return fail(c, CVH_ERR_STATE, "no level set (call cvh_set_levelset or cvh_init_checkerboard first)");
  if (!c->stop_valid) {   // planes changed on the device (Perona-Malik): src/main.cpp:950 uses the smoothed channels
    const int rc = image_stats(c, nullptr);
    if (rc != CVH_OK) return rc;
  }
  c->stop_cond_h = c->p.tol * c->stop_norm;  // :959 (a launch argument: part of the graph key)
  if (c->state_bits == 32 && (!use_fast(c) || resolve_geometry(c).strip != 3))
    return fail(c, CVH_ERR_ARG, "state 32 runs the 2-pixel wave kernel in FAST arithmetic only (math_mode, kernel)");
  return CVH_OK;
}

// Makes c1/c2 of the current level set and the stop condition valid on the device.
static int prepare(cvh_context *c)
{
  int rc0 = prepare_host(c);
  if (rc0 != CVH_OK) return rc0;
  // (the stop condition travels as a launch argument, CvhStepArgs::stop_cond: no per-enqueue upload inside the timed interval)
  const bool chain = use_chain(c, resolve_geometry(c));
  if (chain && !c->chain_acc_valid) c->sums_valid = false;   // the means exist only as doubles (another kernel ran): recompute
  if (!c->sums_valid) {
    CvhStepArgs a;
    fill_args(c, &a, current_buffer(c), c->enqueued);
    if (c->state_bits == 32) {   // the sums of the level set the run starts from are taken of its double mirror (the rounded values)
      const int rc = ensure_f64_mirror(c);
      if (rc != CVH_OK) return rc;
      a.u_in = c->d_u[current_buffer(c)];
    }
    int nparts = 0;
    HIPCHK(c, cvh_launch_init_sums(a, c->C, use_fast(c), &nparts, c->stream));
    a.nparts = nparts;
    HIPCHK(c, cvh_launch_finalize(a, c->C, 1, c->stream));   // chain mode: also seeds the fixed-point set of this step
    c->sums_valid = true;
    c->chain_acc_valid = chain;
  }
  return CVH_OK;
}

// Wave kernel: rows [bounds[k], bounds[k+1]) belong to strip k.  All waves start together, but at
// equal priority the SIMD arbiter favours the OLDEST wave, i.e. the lowest workgroup index, and
// equal strips then finish up to 10 us apart inside one SIMD (tools/wave_timeline.py) -- the tail
// runs at 1-2 waves per SIMD.  wave_skew = 1000 alpha makes the strip length fall linearly from
// (1 + alpha) to (1 - alpha) times the mean with the strip index, so they finish together.
// First row of every strip of the wave kernels, b[0 .. S] (pure host arithmetic: also exported for the CPU tests).
//   kind 3: 2-pixel kernel (a workgroup is 2 wave-columns of 2 strips), kind 2: 1-pixel kernel (4 wave-columns of ONE strip)
//   cls > 0: class-major workgroup numbering with `cls` workgroups per XCD per dispatch round; cskew = per-mille skew between rounds
//   cls == 0: equal strips of strip_rows rows (skew: the 1-pixel kernel's legacy linear skew)
static void compute_strip_bounds(int kind, int h, int tiles_x, int S, int strip_rows, int nblocks, int cls, int cskew, int skew,
                                 std::vector<int> &b)
{
  b.assign((size_t)S + 1, 0);
  if (cls) {
    // Class-major numbering (csv_wave2_kernel.hip): the hardware deals workgroup b to XCD b % 8 and, inside an XCD, the first
    // `cls` workgroups to distinct CUs, the next `cls` to the same CUs again, ... (measured, tools/wave_timeline.py: a CU holds
    // workgroups j, j + 32, j + 64 of its XCD, in wave slots 0, 1, 2).  At equal priority the SIMD arbiter serves the OLDEST
    // wave first, so round 0 finishes 4 us before round 1 and 8 us before round 2 (53 / 57 / 61 us) and the tail of every launch
    // runs at 2, then 1 wave per SIMD.  The class-major numbering makes the strips of one round contiguous, and cskew = 1000 a
    // gives the rounds (1 + a), 1, (1 - a) times the mean strip length.  Rows are dealt by cumulative weight: no short last strip.
    const int spw = kind == 3 ? 2 : 1;
    const int nbc = kind == 3 ? (tiles_x + 1) / 2 : (tiles_x + 3) / 4, nb = nblocks, q = nb >> 3, r = nb & 7;
    const int npairs = (S + spw - 1) / spw;
    int ncls = 0;
    std::vector<long> K;                       // K[k] = workgroups in rounds 0..k
    for (;; ++ncls) {
      long tot = 0;
      for (int x = 0; x < 8; ++x) { const int nx = q + (x < r ? 1 : 0); const long lim = (long)(ncls + 1) * cls; tot += nx < lim ? nx : lim; }
      K.push_back(tot);
      if (tot >= nb) { ++ncls; break; }
    }
    const double a_ = cskew / 1000.0, mid = (ncls - 1) / 2.0;
    std::vector<double> wgt((size_t)S);
    double total = 0;
    for (int sp = 0; sp < npairs; ++sp) {
      const long rank = (long)sp * nbc + nbc / 2;
      int k = 0;
      while (k < ncls - 1 && rank >= K[k]) ++k;
      const double wv = 1.0 + a_ * (mid - k) / (mid > 0 ? mid : 1.0);
      for (int t = 0; t < spw && spw * sp + t < S; ++t) { wgt[spw * sp + t] = wv; total += wv; }
    }
    double cum = 0;
    for (int k = 0; k < S; ++k) { b[k] = (int)((double)h * (cum / total) + 0.5); cum += wgt[k]; }
    for (int k = 1; k < S; ++k) if (b[k] < b[k - 1]) b[k] = b[k - 1];
  } else {
    const double alpha = skew / 1000.0;
    for (int k = 0; k <= S; ++k) {
      long v;
      if (skew == 0) v = (long)k * strip_rows;
      else { const double x = (double)k / S; v = (long)((double)h * (x + alpha * x * (1.0 - x))); }
      b[k] = (int)(v < h ? v : h);
    }
  }
  b[S] = h;
}

// Diagnostic (not part of include/chanvese_hip.h): the strip table for a geometry, without a device.  out needs S + 1 ints.
extern "C" int cvh_debug_strip_bounds(int kind, int h, int tiles_x, int S, int strip_rows, int nblocks, int cls, int cskew, int skew, int *out)
{
  if (!out || S < 1 || h < 1 || (kind != 2 && kind != 3)) return CVH_ERR_ARG;
  std::vector<int> b;
  compute_strip_bounds(kind, h, tiles_x, S, strip_rows, nblocks, cls, cskew, skew, b);
  memcpy(out, b.data(), b.size() * sizeof(int));
  return CVH_OK;
}

// Diagnostic (not part of include/chanvese_hip.h): which per-launch data flow resolve_geometry() picks for a shape and option set, without a
// device -- 0 tile kernel, 2 wave kernel, 3 wave kernel with 2 pixels per lane -- and its grid (tests/test_host_geometry.py pins the dispatch,
// e.g. the tile kernel from 2^28 pixels on, which no GPU test launches).
extern "C" int cvh_debug_data_flow(int h, int w, int channels, int math_mode, int kernel, int state_bits, int num_cus, int *flow, int *tiles_x,
                                   int *tiles_y, int *strip_rows)
{
  if (h < 1 || w < 1 || (channels != 1 && channels != 3) || !flow) return CVH_ERR_ARG;
  cvh_context c;
  c.h = h; c.w = w; c.C = channels; c.n = (size_t)h * (size_t)w;
  c.math_mode = math_mode; c.kernel = kernel; c.state_bits = state_bits; c.num_cus = num_cus > 0 ? num_cus : 256;
  const Geometry g = resolve_geometry(&c);
  *flow = g.strip;
  if (tiles_x) *tiles_x = g.tiles_x;
  if (tiles_y) *tiles_y = g.tiles_y;
  if (strip_rows) *strip_rows = g.strip_rows;
  return CVH_OK;
}

static int upload_strip_bounds(cvh_context *c, const Geometry &g)
{
  const int cls = (((g.strip == 3 && c->wave_cls) || (g.strip == 2 && c->wave_cls == 2)) && c->wave_xcd) ? (c->num_cus >= 8 ? c->num_cus / 8 : 1) : 0;
  const bool alone = run_is_alone(c);
  const int key[4] = {g.tiles_y, g.strip_rows, c->wave_skew + 1000 * (cls ? c->wave_cskew + 1 : 0) + 10000000 * g.strip + (c->state_bits == 32 ? 500000000 : 0) + (alone ? 0 : 250000000), c->h};
  if (!memcmp(key, c->bounds_key, sizeof(key))) return CVH_OK;
  std::vector<int> b;
  // The skew pays for short strips only (one process, 2-pixel kernel: 4096^2, 46 rows: 61.1 -> 58.7 us; 6144^2, 102 rows: 140.9 ->
  // 140.6; 8192^2 forced onto this kernel, 178 rows: 244 -> 285 us): full below 46 rows, fading to none at 128.
  int cskew = c->wave_cskew;
  if (g.strip_rows > 46) cskew = g.strip_rows >= 128 ? 0 : (int)(cskew * (128.0 - g.strip_rows) / (128.0 - 46.0));
  // three channels: round 3 measured equal strips best (73.3 vs 74.1 us with the full skew, with the workgroup barrier per group); without that
  // barrier (their default since) the wave timeline shows the staircase again -- strips of dispatch round 0 end at 65 us, of round 2 at 74-76 -- and
  // a skew of 0.425 wins: five alternations in one context (round 4, gpurun_out/r4s17) 73.00 (equal) / 71.69 (0.35) / 70.88 (0.425) / 71.74 (0.5 +
  // priority scheme 2) us.  Any other "wave_cskew" applies as given.
  // (the FP32-state flavour of three channels, compute-bound, still prefers equal strips: 52.5 vs 54.6 us, gpurun_out/r4s19)
  if (c->C == 3 && c->wave_cskew == 500) cskew = c->state_bits == 32 ? 0 : 425;
  // a batch: launches of several contexts interleave on the CUs, the staircase of ONE launch's dispatch rounds is not what ends a launch any more --
  // equal strips (8 interleaved 4096^2 images, `bench.py --config C5`: 298.1 k against 293.2-293.4 k Mpixel-iterations/s, gpurun_out/r4s23)
  if (!alone && c->wave_cskew == 500) cskew = 0;
  compute_strip_bounds(g.strip, c->h, g.tiles_x, g.tiles_y, g.strip_rows, g.nblocks, cls, cskew, c->wave_skew, b);
  HIPCHK(c, hipStreamSynchronize(c->stream));  // launches already enqueued read the old table
  HIPCHK(c, hipMemcpy(c->d_bounds, b.data(), b.size() * sizeof(int), hipMemcpyHostToDevice));
  memcpy(c->bounds_key, key, sizeof(key));
  return CVH_OK;
}

// `capturing`: the launch is recorded into a stream capture, nothing reaches the GPU -- the context's bookkeeping of what is
// in flight (chain_pending, chain_acc_valid) is updated by the caller when the graph is really launched
static int launch_one_step(cvh_context *c, int in_buf, int step, bool capturing = false, CvhLaunchNote *note = nullptr)
{
  CvhStepArgs a;
  fill_args(c, &a, in_buf, step);
  a.note = note;
  const int kind = resolve_geometry(c).strip;
  if (kind == 3) HIPCHK(c, cvh_launch_wave2(a, c->C, use_fast(c), c->stream));
  else if (kind == 2) HIPCHK(c, cvh_launch_wave(a, c->C, use_fast(c), c->stream));
  else HIPCHK(c, cvh_launch_step(a, c->C, use_fast(c), c->stream));
  if (note) return CVH_OK;
  if (c->finalize_mode == 1) HIPCHK(c, cvh_launch_finalize(a, c->C, 0, c->stream));
  if (!capturing) {
    if (a.chain) c->chain_pending = true;
    else c->chain_acc_valid = false;   // the means now live in the state block only
  }
  return CVH_OK;
}

// Chain mode: the last launch's iteration has no successor to book it -- one small kernel does (norm, stop rule,
// trace row) and writes the region means of the current level set into the state block.
static int chain_flush(cvh_context *c)
{
  if (!c->chain_pending) return CVH_OK;
  CvhStepArgs a;
  fill_args(c, &a, 0, 0);
  if (!a.chain) return fail(c, CVH_ERR_STATE, "chain-mode launches are pending but the context no longer selects chain mode");
  HIPCHK(c, cvh_launch_chain_flush(a, c->C, c->stream));
  c->chain_pending = false;
  return CVH_OK;
}

// A run of kGraphSteps consecutive steps as one hipGraph (launch arguments differ between steps only
// in the ping-pong parity; step counter, trace row and stop flag live on the device).  Measured on
// MI355X: back-to-back launches on a stream cost 2.8 us each, graph nodes 1.6 us (tools/launch_probe.hip).
// The instantiated graph is kept per start parity and rebuilt when any launch argument changed.
static int ensure_step_graph(cvh_context *c, int parity)
{
  // One slot per sum-set phase of the first step: a chunk size that is not a multiple of 4 (sync_every = 18, repeated
  // cvh_enqueue_steps(18), a run that stopped early) cycles through the phases, and each keeps its instantiated graph.
  StepGraph &g = c->graphs[(c->chain_pb + c->enqueued) & 3];
  // the arguments of consecutive steps differ in the ping-pong parity and the chain-mode sum set: period 4
  CvhStepArgs key[4];
  for (int s = 0; s < 4; ++s) fill_args(c, &key[s], parity ^ (s & 1), c->enqueued + s);
  const int kind = resolve_geometry(c).strip, flavour = (use_fast(c) ? 1 : 0) | (c->finalize_mode << 1);
  if (g.exec && g.kind == kind && g.flavour == flavour && !memcmp(key, g.key, sizeof(key))) return CVH_OK;
  if (g.exec) {   // an argument changed: the old exec may still have launches in flight (cvh_run queues chunks ahead)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipGraphExecDestroy(g.exec);
    g.exec = nullptr;
  }
  HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int rc = CVH_OK;
  for (int s = 0; s < kGraphSteps && rc == CVH_OK; ++s) rc = launch_one_step(c, parity ^ (s & 1), c->enqueued + s, true);
  hipGraph_t graph = nullptr;
  const hipError_t e_end = hipStreamEndCapture(c->stream, &graph);
  if (rc != CVH_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e_end != hipSuccess || !graph) return fail(c, CVH_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e_end));
  const hipError_t e_inst = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e_inst != hipSuccess) { g.exec = nullptr; return fail(c, CVH_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e_inst)); }
  memcpy(g.key, key, sizeof(key));
  g.kind = kind; g.flavour = flavour;
  return CVH_OK;
}

// One-off HOST work of a run: the strip table and, when the run is long enough to use them, the
// instantiated graph of the run's ping-pong parity (stream capture + hipGraphInstantiate cost about a
// millisecond each while the GPU idles).  Called before the timed interval opens, so that neither
// cvh_last_run_ms nor a caller's wall clock around cvh_enqueue_steps / cvh_sync is charged with it.
static int warm_impl(cvh_context *c, long nsteps)
{
  if (nsteps > 0) c->run_chunk = (int)(nsteps < 1024 ? nsteps : 1024);   // (the caller announces its next enqueue)
  { ResidentGeom rg; if (resident_geometry(c, &rg)) return CVH_OK; }   // one cooperative launch per chunk: nothing to capture
  const Geometry g = resolve_geometry(c);
  if (g.strip >= 2) { const int rc = upload_strip_bounds(c, g); if (rc != CVH_OK) return rc; }
  if (c->use_graph && nsteps >= kGraphSteps) {
    // every graph launch of one call starts on the same ping-pong parity / sum-set phase (kGraphSteps is a multiple of 4),
    // after the nsteps % kGraphSteps plain launches that enqueue_impl() issues first
    const int ahead = (int)(nsteps % kGraphSteps);
    c->enqueued += ahead;
    const int rc = ensure_step_graph(c, (c->cur_base + c->enqueued) & 1);
    c->enqueued -= ahead;
    if (rc != CVH_OK) return rc;
  }
  return CVH_OK;
}

// Synchronisation words and border buffer of the resident kernels (csv_resident_kernel.hip, pm_resident_kernel.hip), pinned error word.
static int ensure_resident_buffers(cvh_context *c)
{
  if (c->d_resident) return CVH_OK;
  const int halo = cvh_resident_halo_doubles();
  // (fine-grained and uncached device memory -- hipExtMallocWithFlags -- for these lines and buffers were tried: no difference,
  // profiles/r04_C4/resident_memory_kinds.txt)
  HIPCHK(c, hipMalloc((void **)&c->d_resident, sizeof(CvhResident)));
  HIPCHK(c, hipMalloc((void **)&c->d_res_halo, (size_t)2 * CVH_RESIDENT_MAX_TILES * halo * sizeof(double)));
  HIPCHK(c, hipHostMalloc((void **)&c->h_resident, 64, hipHostMallocDefault));
  memset(c->h_resident, 0, 64);
  return CVH_OK;
}

// One cooperative launch per chunk of iterations (csv_resident_kernel.hip).
static int launch_resident(cvh_context *c, const ResidentGeom &rg, int nsteps, CvhLaunchNote *note)
{
  const int ntiles = rg.tr * rg.tc;
  if (!note) { const int rc = ensure_resident_buffers(c); if (rc != CVH_OK) return rc; }
  constexpr int kMaxPerLaunch = 4096;
  for (int s = 0; s < nsteps || note;) {
    const int n = nsteps - s < kMaxPerLaunch ? nsteps - s : kMaxPerLaunch;
    CvhStepArgs a;
    fill_args(c, &a, (c->cur_base + c->enqueued) & 1, c->enqueued);
    if (!a.chain) return fail(c, CVH_ERR_STATE, "resident mode needs chain-mode sums");
    a.tiles_x = rg.tc; a.tiles_y = rg.tr; a.nparts = ntiles;
    {   // every tile 16, 32, 64 or 128 rows: the straight-line flavour of the march
      const int th = c->h % rg.tr == 0 ? c->h / rg.tr : 0;
      a.res_band_rows = (c->res_straight && (th == 16 || th == 32 || th == 64 || th == 128)) ? th / 8 : 0;
    }
    a.resident = c->d_resident;
    a.res_halo = c->d_res_halo;
    a.res_steps = n;
    a.res_t0 = c->enqueued;
    a.res_poll_cap = 2000000;      // seconds of polling before a wait gives up (the grid always drains)
    a.note = note;
    if (note) { HIPCHK(c, cvh_launch_resident(a, c->stream)); return CVH_OK; }
    HIPCHK(c, hipMemsetAsync(c->d_resident, 0, sizeof(CvhResident), c->stream));
    HIPCHK(c, cvh_launch_resident(a, c->stream));
    c->chain_pending = true;       // the flush kernel writes c1 / c2 of the final level set into the state block at the next sync
    c->resident_used = true;
    c->enqueued += n;
    s += n;
  }
  return CVH_OK;
}

static int enqueue_impl(cvh_context *c, int nsteps)
{
  if (c->state_bits == 32 && nsteps > 0) c->mirror_valid = false;
  if (nsteps > 0) c->run_chunk = nsteps;        // (resident_geometry's rule for a batch looks at the length of THIS enqueue)
  {
    ResidentGeom rg;
    if (resident_geometry(c, &rg)) return launch_resident(c, rg, nsteps, nullptr);
    const Geometry g = resolve_geometry(c);
    if (g.strip >= 2) { const int rc = upload_strip_bounds(c, g); if (rc != CVH_OK) return rc; }
  }
  // The odd-sized part goes FIRST as plain launches: from an idle stream they reach the GPU within 3-5 us, while the first
  // hipGraph replay takes 10-16 us; the graphs (runs of kGraphSteps) follow.  warm_impl() builds the graph for that position.
  int s = 0;
  const int plain = (c->use_graph && nsteps >= kGraphSteps) ? nsteps % kGraphSteps : nsteps;
  for (; s < plain; ++s) {
    const int rc = launch_one_step(c, (c->cur_base + c->enqueued) & 1, c->enqueued);
    if (rc != CVH_OK) return rc;
    c->enqueued++;
  }
  while (nsteps - s >= kGraphSteps) {
    const int parity = (c->cur_base + c->enqueued) & 1;
    const int rc = ensure_step_graph(c, parity);
    if (rc != CVH_OK) return rc;
    const StepGraph &sg = c->graphs[(c->chain_pb + c->enqueued) & 3];
    HIPCHK(c, hipGraphLaunch(sg.exec, c->stream));
    if (sg.key[0].chain) c->chain_pending = true; else c->chain_acc_valid = false;
    c->enqueued += kGraphSteps;
    s += kGraphSteps;
  }
  return CVH_OK;
}

extern "C" int cvh_enqueue_steps(cvh_context *c, int nsteps)
{
  if (!c || nsteps < 0) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = prepare_host(c);
  if (rc != CVH_OK) return rc;
  rc = warm_impl(c, nsteps);   // graph build etc. stays outside the timed interval
  if (rc != CVH_OK) return rc;
  if (!c->timing_open) {
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    c->timing_open = true;
  }
  rc = prepare(c);
  if (rc != CVH_OK) return rc;
  return enqueue_impl(c, nsteps);
}

extern "C" int cvh_warm(cvh_context *c, int nsteps)
{
  if (!c || nsteps < 0) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = prepare_host(c);
  if (rc != CVH_OK) return rc;
  return warm_impl(c, nsteps);
}

static int absorb_state(cvh_context *c, const CvhState *hs)
{
  c->steps_done = hs->steps_done;
  if (hs->stopped) c->enqueued = hs->steps_done;  // launches past the stop were no-ops
  return CVH_OK;
}

static int sync_impl(cvh_context *c)
{
  const bool via_flush = c->chain_pending;   // the flush kernel writes {steps_done, stopped, norm} into the pinned host block itself
  int rc = chain_flush(c);
  if (rc != CVH_OK) return rc;
  if (c->timing_open) HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  if (!via_flush) HIPCHK(c, hipMemcpyAsync(&c->h_state[0], c->d_state, sizeof(CvhState), hipMemcpyDeviceToHost, c->stream));
  if (c->resident_used) HIPCHK(c, hipMemcpyAsync(c->h_resident, c->d_resident, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->resident_used) {
    c->resident_used = false;
    if (c->h_resident[0]) {
      c->timing_open = false;
      return fail(c, CVH_ERR_HIP, "the resident step kernel gave up waiting for a workgroup: the level set of this run is invalid");
    }
  }
  if (via_flush) {
    c->h_state[0].steps_done = c->h_status[0];
    c->h_state[0].stopped = c->h_status[1];
    memcpy(&c->h_state[0].norm, &c->h_status[2], sizeof(double));
  }
  if (c->timing_open) {
    HIPCHK(c, hipEventElapsedTime(&c->last_run_ms, c->ev0, c->ev1));
    c->timing_open = false;
  }
  absorb_state(c, &c->h_state[0]);
  if (!c->h_state[0].stopped) c->enqueued = c->steps_done;
  return CVH_OK;
}

extern "C" int cvh_sync(cvh_context *c, int *steps_done_total, double *last_norm, int *stopped)
{
  if (!c) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const int rc = sync_impl(c);
  if (rc != CVH_OK) return rc;
  if (steps_done_total) *steps_done_total = c->h_state[0].steps_done;
  if (last_norm) *last_norm = c->h_state[0].norm;
  if (stopped) *stopped = c->h_state[0].stopped;
  return CVH_OK;
}

extern "C" int cvh_run(cvh_context *c, int max_steps, int *steps_done, double *last_norm)
{
  if (!c) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->have_image) return fail(c, CVH_ERR_STATE, "no image set (call cvh_set_image first)");
  if (!c->have_u) return fail(c, CVH_ERR_STATE, "no level set (call cvh_set_levelset or cvh_init_checkerboard first)");
  int rc = reset_run_impl(c);   // settles whatever is in flight first
  if (rc != CVH_OK) return rc;
  long remaining = max_steps < 0 ? (long)INT_MAX : (long)max_steps;  // src/main.cpp:890
  rc = prepare_host(c);  // one-off host work (src/main.cpp:950-959) stays outside the device timing
  if (rc != CVH_OK) return rc;
  rc = warm_impl(c, remaining < c->sync_every ? remaining : (long)c->sync_every);  // so do the strip table and the graph of the first chunk
  if (rc != CVH_OK) return rc;
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  rc = prepare(c);
  if (rc != CVH_OK) return rc;
  // Chunks of sync_every launches.  The finalising workgroup of every step stores
  // {steps_done, stopped} into pinned host memory; the host reads those two words between
  // chunks (no copy, no synchronisation) and never runs more than kAhead chunks in front of
  // the device.  Launches queued behind a fired stop are no-ops on the device (sticky flag).
  constexpr int kAhead = 4;
  volatile int *hs = c->h_status;
  bool stopped = false;
  int queued = 0;
  // resident mode: a chunk is ONE launch that loads the tiles, iterates and stores them; the stop rule ends it inside the kernel at the
  // reference's iteration, so chunks can be long (the tile load / store of a 2048^2 plane is worth ~0.4 us per iteration at 32)
  int chunk_len = c->sync_every;
  c->run_chunk = (int)(remaining < 1024 ? remaining : 1024);     // (what a chunk is if the run takes the resident flow: resident_geometry's rule for a batch)
  { ResidentGeom rg; if (resident_geometry(c, &rg) && chunk_len < 1024) chunk_len = 1024; else c->run_chunk = chunk_len; }
  while (remaining > 0 && !stopped) {
    while (queued - hs[0] > kAhead * chunk_len && !hs[1]) {
      if (hipStreamQuery(c->stream) == hipSuccess) break;  // everything queued has run
    }
    if (hs[1]) { stopped = true; break; }
    const int chunk = (int)(remaining < chunk_len ? remaining : chunk_len);
    rc = enqueue_impl(c, chunk);
    if (rc != CVH_OK) return rc;
    remaining -= chunk;
    queued += chunk;
  }
  c->timing_open = true;   // sync_impl closes the interval opened at ev0 (after the chain-mode flush)
  rc = sync_impl(c);
  if (rc != CVH_OK) return rc;
  c->enqueued = c->steps_done;
  if (steps_done) *steps_done = c->h_state[0].steps_done;
  if (last_norm) *last_norm = c->h_state[0].norm;
  return CVH_OK;
}

extern "C" int cvh_get_means(cvh_context *c, double *c1, double *c2)
{
  if (!c || !c1 || !c2) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->timing_open) return fail(c, CVH_ERR_STATE, "cvh_get_means: steps in flight, call cvh_sync first");
  int rc = prepare(c);
  if (rc != CVH_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(&c->h_state[0], c->d_state, sizeof(CvhState), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < c->C; ++k) { c1[k] = c->h_state[0].c1[k]; c2[k] = c->h_state[0].c2[k]; }
  return CVH_OK;
}

extern "C" int cvh_get_trace(cvh_context *c, double *out, int max_rows, int *rows)
{
  if (!c || !out || !rows || max_rows < 0) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  int n = c->steps_done < c->trace_cap ? c->steps_done : c->trace_cap;
  if (n > max_rows) n = max_rows;
  *rows = n;
  if (n > 0) {
    HIPCHK(c, hipMemcpyAsync(out, c->d_trace, (size_t)n * (2 * c->C + 1) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return CVH_OK;
}

extern "C" int cvh_get_stop_condition(cvh_context *c, double *stop_cond)
{
  if (!c || !stop_cond) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = prepare(c);
  if (rc != CVH_OK) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *stop_cond = c->p.tol * c->stop_norm;
  return CVH_OK;
}

extern "C" int cvh_get_mask(cvh_context *c, uint8_t *mask, int invert)
{
  if (!c || !mask) return CVH_ERR_ARG;
  if (!c->have_u) return fail(c, CVH_ERR_STATE, "cvh_get_mask: no level set");
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->d_mask) HIPCHK(c, hipMalloc((void **)&c->d_mask, c->n));
  { const int rc = ensure_f64_mirror(c); if (rc != CVH_OK) return rc; }
  HIPCHK(c, cvh_launch_mask(c->d_u[current_buffer(c)], c->d_mask, c->n, invert, c->stream));
  HIPCHK(c, hipMemcpyAsync(mask, c->d_mask, c->n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CVH_OK;
}

extern "C" int cvh_get_contour(cvh_context *c, uint8_t *contour)
{
  if (!c || !contour) return CVH_ERR_ARG;
  if (!c->have_u) return fail(c, CVH_ERR_STATE, "cvh_get_contour: no level set");
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->d_mask) HIPCHK(c, hipMalloc((void **)&c->d_mask, c->n));
  { const int rc = ensure_f64_mirror(c); if (rc != CVH_OK) return rc; }
  HIPCHK(c, cvh_launch_contour(c->d_u[current_buffer(c)], c->d_mask, c->h, c->w, c->stream));
  HIPCHK(c, hipMemcpyAsync(contour, c->d_mask, c->n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CVH_OK;
}

extern "C" int cvh_separate(cvh_context *c, const uint8_t *img3, int invert, uint8_t *selection3)
{
  if (!c || !img3 || !selection3) return CVH_ERR_ARG;
  if (!c->have_u) return fail(c, CVH_ERR_STATE, "cvh_separate: no level set");
  HIPCHK(c, hipSetDevice(c->device));
  { const int rc = ensure_f64_mirror(c); if (rc != CVH_OK) return rc; }
  uint8_t *d_in = nullptr, *d_out = nullptr;
  HIPCHK(c, hipMalloc((void **)&d_in, c->n * 3));
  hipError_t e = hipMalloc((void **)&d_out, c->n * 3);
  if (e != hipSuccess) { (void)hipFree(d_in); return fail(c, CVH_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
  int rc = CVH_OK;
  do {
    if ((e = hipMemcpyAsync(d_in, img3, c->n * 3, hipMemcpyHostToDevice, c->stream)) != hipSuccess) break;
    if ((e = cvh_launch_separate(d_in, c->d_u[current_buffer(c)], d_out, c->n, invert, c->stream)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(selection3, d_out, c->n * 3, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) break;
    e = hipStreamSynchronize(c->stream);
  } while (0);
  if (e != hipSuccess) rc = fail(c, CVH_ERR_HIP, "cvh_separate: %s", hipGetErrorString(e));
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return rc;
}

extern "C" int cvh_pm_trip_count(double L, double T)
{
  int n = 0;
  for (double t = 0; t < T; t += L) {  // src/main.cpp:498: the counter itself is a double
    if (++n == INT_MAX) break;
    if (!(L > 0)) break;               // L == 0 would never terminate; one step is what T >= L allows
  }
  return n;
}

// "pm_kernel" = -1: the resident kernel pays ~25 us per cooperative launch that the per-launch flow does not, and gains 0.9 us per step
// on small planes, 1.6 at 1024^2, 4 at 2048^2 (tools/pm_flows.py): runs shorter than this keep the per-launch flow
static int pm_resident_min_trips(size_t n) { return n >= ((size_t)3 << 20) ? 8 : (n >= ((size_t)1 << 20) ? 16 : 32); }

// Perona-Malik with the plane resident in LDS: per channel uint8 -> FP64 plane, ONE cooperative launch per chunk of time steps,
// FP64 -> uint8 (round-half-even, :551) behind the last step.
static int pm_run_resident(cvh_context *c, const CvhPmArgs &base, const ResidentGeom &rg, int trips)
{
  { const int rc = ensure_resident_buffers(c); if (rc != CVH_OK) return rc; }
  CvhPmArgs a = base;
  a.tiles_x = rg.tc; a.tiles_y = rg.tr; a.res_band_rows = rg.band;
  a.res_prio = c->res_prio;
  a.resident = c->d_resident;
  if (!c->d_pm_halo) {
    const size_t bytes = (size_t)2 * CVH_RESIDENT_MAX_TILES * cvh_pm_resident_halo_doubles() * sizeof(double);
    HIPCHK(c, hipMalloc((void **)&c->d_pm_halo, bytes));
    HIPCHK(c, hipMemset(c->d_pm_halo, 0, bytes));            // tag 0: matches no launch
  }
  a.res_halo = c->d_pm_halo;
  a.res_poll_cap = 2000000;
  a.dbg_times = c->d_dbg;
  constexpr int kMaxPerLaunch = 1 << 16;
  {
    CvhLaunchNote nb{};
    CvhPmArgs pa = a; pa.note = &nb; pa.res_steps = trips;
    (void)cvh_launch_pm_resident(pa, c->stream);
    snprintf(c->pm_desc, sizeof(c->pm_desc), "kernel=%s grid=%u block=%u lds_bytes=%u steps_per_launch=%d tiles_y=%d tiles_x=%d launches=%d graph_launches=0 trips=%d planes=%d",
             nb.name, nb.grid, nb.block, nb.lds, trips < kMaxPerLaunch ? trips : kMaxPerLaunch, rg.tr, rg.tc, (trips + kMaxPerLaunch - 1) / kMaxPerLaunch, trips, c->C);
  }
  HIPCHK(c, hipMemsetAsync(c->d_resident, 0, sizeof(CvhResident), c->stream));
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  for (int k = 0; k < c->C; ++k) {
    HIPCHK(c, cvh_launch_pm_load(c->d_img[k], c->d_pm[0], c->n, c->stream));
    int cur = 0;
    for (int t = 0; t < trips;) {
      const int n = trips - t < kMaxPerLaunch ? trips - t : kMaxPerLaunch;
      CvhPmArgs pa = a;
      pa.in = c->d_pm[cur]; pa.out = c->d_pm[cur ^ 1]; pa.res_steps = n;
      pa.res_serial = ++c->pm_res_serial;        // border entries carry {serial, step}: nothing an earlier launch left can match
      HIPCHK(c, cvh_launch_pm_resident(pa, c->stream));
      cur ^= 1;
      t += n;
    }
    HIPCHK(c, cvh_launch_pm_store(c->d_pm[cur], c->d_img[k], c->n, c->stream));
  }
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_resident, c->d_resident, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipEventElapsedTime(&c->last_pm_ms, c->ev0, c->ev1));
  c->stop_valid = false;
  c->sums_valid = false;
  if (c->h_resident[0]) {
    c->h_resident[0] = 0;
    return fail(c, CVH_ERR_HIP, "cvh_perona_malik: a wait of the resident kernel gave up (a workgroup was not resident, or a fault); the planes are undefined");
  }
  return CVH_OK;
}

extern "C" int cvh_perona_malik(cvh_context *c, double K, double L, double T)
{
  if (!c) return CVH_ERR_ARG;
  if (!c->have_image) return fail(c, CVH_ERR_STATE, "cvh_perona_malik: no image set");
  // src/main.cpp:863-867
  if (L > 0.25 || L < 0)
    return fail(c, CVH_ERR_ARG, "The Laplacian coefficient in Perona-Malik segmentation must be between 0 and 0.25.");
  if (T < L)
    return fail(c, CVH_ERR_ARG, "The segmentation duration must exceed the value of Laplacian coefficient, %f.", L);
  if (K == 0) return fail(c, CVH_ERR_ARG, "cvh_perona_malik: edge coefficient K must be non-zero");
  HIPCHK(c, hipSetDevice(c->device));
  // CSV work that was enqueued and never synchronised is closed first, as cvh_set_image does: the resident Perona-Malik flow clears the
  // shared CvhResident block (the error word of an unsynchronised csv_resident_kernel launch with it) and reuses ev0 / ev1.
  if (c->timing_open || c->chain_pending || c->resident_used) { const int rc = sync_impl(c); if (rc != CVH_OK) return rc; }
  const int trips = cvh_pm_trip_count(L, T);
  for (int k = 0; k < 2; ++k)
    if (!c->d_pm[k]) HIPCHK(c, hipMalloc((void **)&c->d_pm[k], c->n * sizeof(double)));
  CvhPmArgs a;
  memset(&a, 0, sizeof(a));
  a.h = c->h; a.w = c->w; a.K2 = K * K; a.L = L;
  a.invK2 = 1.0 / (K * K); a.L4 = L / 4; a.fast = use_fast(c) ? 1 : 0;
  a.pol = (c->wave_pol >= 0 ? (c->wave_pol == 1) : ((double)c->n * 16.0 <= 300e6 ? 1 : 0));
  // A plane whose FP64 state fits the chip's LDS stays there for the whole run (pm_resident_kernel.hip): one cooperative launch per
  // channel, the tiles' borders cross workgroups, nothing else moves.
  {
    ResidentGeom rg;
    const bool want = c->pm_kernel == 4 || (c->pm_kernel == -1 && c->pm_strip_rows == 0 && trips >= pm_resident_min_trips(c->n));
    if (want && trips > 0 && pm_resident_geometry(c, &rg)) return pm_run_resident(c, a, rg, trips);
    if (c->pm_kernel == 4 && trips > 0)
      return fail(c, CVH_ERR_ARG, "pm_kernel 4 (resident plane) needs an even width, >= 16 rows and columns, and a plane that fits the LDS of the CUs");
  }
  // two time steps per launch (pm_wave_k2_kernel.hip): planes that fit the caches, where a step is launch / latency bound
  // (a 2-pixel-per-lane 1-step kernel was the default from 12 Mpixel on in round 1: 48.4 us/step at 4096^2 against 39.3 for the 2-step
  // kernel -- tools/experiments/pruned_flavours/pm_wave2_kernel.hip)
  const bool pm_k2 = trips >= 2 && c->n < ((size_t)1 << 28) && (c->pm_kernel == 3 || c->pm_kernel == -1);
  const bool pm_wave = c->pm_kernel != 0;
  CvhPmArgs a2 = a;      // geometry of the 2-step kernel (the odd last step runs the 1-step wave kernel)
  if (pm_k2) {
    a2.tiles_x = (c->w + cvh_pm_wave_k2_cols() - 1) / cvh_pm_wave_k2_cols();
    int sr = c->pm_strip_rows;
    if (sr <= 0) {
      // ~51 strips whatever the size (measured, us/step: 1024^2: 16 rows 6.3, 24 rows 6.05, 32 rows 6.7; 2048^2: 16 13.0, 24 13.2,
      // 32 13.3, 40 12.55, 48 13.5, 64 15.5; 4096^2: 48 40.1, 64 39.8, 80 38.2-39.3, 104 38.6, 128 41.9, 160 39.9, 200 44.6)
      sr = (c->h + 50) / 51;
      sr = ((sr + 4) / 8) * 8;   // nearest multiple of 8: the row loop is unrolled by 8
      if (sr < 16) sr = 16;      // every strip pays 5 extra stage-1 rows
    }
    a2.strip_rows = sr;
  }
  if (pm_wave) {
    a.tiles_x = (c->w + cvh_pm_wave_cols() - 1) / cvh_pm_wave_cols();
    int sr = c->pm_strip_rows;
    if (sr <= 0) {  // ~3 waves per SIMD resident
      int nstrips = (c->num_cus * 3) / ((a.tiles_x + 3) / 4);
      if (nstrips < 1) nstrips = 1;
      sr = (c->h + nstrips - 1) / nstrips;
      sr = ((sr + 3) / 8) * 8;  // nearest multiple of the 8-row loop body; measured best: 8 / 8-16 / 24 rows at 512^2 / 1024^2 / 2048^2
      if (sr < 8) sr = 8;
    }
    a.strip_rows = sr;
  } else {
    cvh_pm_grid(c->h, c->w, &a.tiles_x, &a.tiles_y);
  }
  const int kind = pm_wave ? 1 : 0;
  auto launch_pm = [&](const CvhPmArgs &pa) -> hipError_t {
    return pm_wave ? cvh_launch_pm_wave(pa, c->stream) : cvh_launch_pm_step(pa, c->stream);
  };
  const int per_launch = pm_k2 ? 2 : 1;   // time steps per launch of the bulk kernel
  auto launch_bulk = [&](int from, CvhLaunchNote *note = nullptr) -> hipError_t {
    if (!pm_k2) { CvhPmArgs pa = a; pa.in = c->d_pm[from]; pa.out = c->d_pm[from ^ 1]; pa.note = note; return launch_pm(pa); }
    CvhPmArgs pa = a2; pa.in = c->d_pm[from]; pa.out = c->d_pm[from ^ 1]; pa.note = note;
    return cvh_launch_pm_wave_k2(pa, c->stream);
  };
  {   // what this call launches, for cvh_launch_info (filled by the launch sites themselves)
    CvhLaunchNote nb{}, no{};
    (void)launch_bulk(0, &nb);
    const bool odd = trips % per_launch != 0;
    if (odd) { CvhPmArgs pa = a; pa.in = c->d_pm[0]; pa.out = c->d_pm[1]; pa.note = &no; (void)launch_pm(pa); }
    const bool graphed = c->use_graph && trips >= kGraphSteps * per_launch;
    snprintf(c->pm_desc, sizeof(c->pm_desc),
             "kernel=%s grid=%u block=%u steps_per_launch=%d strip_rows=%d launches=%d%s%s graph_launches=%d trips=%d planes=%d",
             nb.name, nb.grid, nb.block, per_launch, pm_k2 ? a2.strip_rows : a.strip_rows, trips / per_launch,
             odd ? " last_step_kernel=" : "", odd ? no.name : "", graphed ? kGraphSteps : 0, trips, c->C);
  }
  // kGraphSteps steps as one hipGraph, as for the CSV step: a graph node costs 1.6 us against 2.8 us for a stream launch
  // (tools/launch_probe.hip) and a 2048^2 step is only ~13 us.  The graph always starts from d_pm[0] (16 is even).
  if (c->use_graph && trips >= kGraphSteps * per_launch) {
    CvhPmArgs key = pm_k2 ? a2 : a;
    key.in = c->d_pm[0]; key.out = c->d_pm[1];
    if (!c->pm_graph || c->pm_graph_kind != kind + 10 * pm_k2 || memcmp(&key, &c->pm_graph_key, sizeof(key))) {
      if (c->pm_graph) { (void)hipGraphExecDestroy(c->pm_graph); c->pm_graph = nullptr; }
      HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      hipError_t e = hipSuccess;
      for (int t = 0; t < kGraphSteps && e == hipSuccess; ++t) e = launch_bulk(t & 1);
      hipGraph_t graph = nullptr;
      const hipError_t e_end = hipStreamEndCapture(c->stream, &graph);
      if (e != hipSuccess || e_end != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        return fail(c, CVH_ERR_HIP, "cvh_perona_malik: graph capture failed (%s)", hipGetErrorString(e != hipSuccess ? e : e_end));
      }
      const hipError_t e_inst = hipGraphInstantiate(&c->pm_graph, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e_inst != hipSuccess) { c->pm_graph = nullptr; return fail(c, CVH_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e_inst)); }
      c->pm_graph_key = key; c->pm_graph_kind = kind + 10 * pm_k2;
    }
  }
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  if (trips > 0) {
    for (int k = 0; k < c->C; ++k) {
      HIPCHK(c, cvh_launch_pm_load(c->d_img[k], c->d_pm[0], c->n, c->stream));
      int cur = 0, t = 0;
      for (; c->use_graph && c->pm_graph && trips - t >= kGraphSteps * per_launch; t += kGraphSteps * per_launch) HIPCHK(c, hipGraphLaunch(c->pm_graph, c->stream));
      for (; trips - t >= per_launch; t += per_launch) { HIPCHK(c, launch_bulk(cur)); cur ^= 1; }
      for (; t < trips; ++t) {   // the odd last step of the 2-step flavour
        a.in = c->d_pm[cur]; a.out = c->d_pm[cur ^ 1];
        HIPCHK(c, launch_pm(a));
        cur ^= 1;
      }
      HIPCHK(c, cvh_launch_pm_store(c->d_pm[cur], c->d_img[k], c->n, c->stream));
    }
  }
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipEventElapsedTime(&c->last_pm_ms, c->ev0, c->ev1));
  c->stop_valid = false;
  c->sums_valid = false;
  return CVH_OK;
}

extern "C" int cvh_last_run_ms(cvh_context *c, float *ms)
{
  if (!c || !ms) return CVH_ERR_ARG;
  *ms = c->last_run_ms;
  return CVH_OK;
}

extern "C" int cvh_last_pm_ms(cvh_context *c, float *ms)
{
  if (!c || !ms) return CVH_ERR_ARG;
  *ms = c->last_pm_ms;
  return CVH_OK;
}

extern "C" int cvh_ppf_apply_device(double *d_data, long n, int op, double eps, void *stream)
{
  if (!d_data || n < 0 || op < 0 || op > 2) return fail(nullptr, CVH_ERR_ARG, "cvh_ppf_apply_device: bad argument");
  hipError_t e = cvh_launch_ppf(d_data, (size_t)n, op, eps, (hipStream_t)stream);
  if (e != hipSuccess) return fail(nullptr, CVH_ERR_HIP, "cvh_ppf_apply_device: %s", hipGetErrorString(e));
  return CVH_OK;
}

extern "C" int cvh_ppf_apply(double *data, int w, long start, long end, int op, double eps, int device)
{
  // data.at<double>(i / w, i % w) of a continuous w-wide matrix is data[i]
  if (!data || w <= 0 || start < 0 || end < start || op < 0 || op > 2)
    return fail(nullptr, CVH_ERR_ARG, "cvh_ppf_apply: bad argument");
  if (end == start) return CVH_OK;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, CVH_ERR_HIP, "cvh_ppf_apply: hipSetDevice: %s (no CPU fallback)", hipGetErrorString(e));
  const size_t n = (size_t)(end - start);
  // OpenCV's backend calls the operator once per sub-range from several host threads: the staging buffer comes from the
  // device's stream-ordered memory pool (cached between calls; no device-wide synchronisation as with hipMalloc / hipFree)
  // and everything runs on a stream of its own, so concurrent sub-ranges do not serialise on the null stream.
  hipStream_t st = nullptr;
  if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess)
    return fail(nullptr, CVH_ERR_HIP, "cvh_ppf_apply: hipStreamCreate: %s", hipGetErrorString(e));
  double *d = nullptr;
  bool pooled = true;
  if (hipMallocAsync((void **)&d, n * sizeof(double), st) != hipSuccess) {
    (void)hipGetLastError();
    pooled = false;
    if ((e = hipMalloc((void **)&d, n * sizeof(double))) != hipSuccess) {
      (void)hipStreamDestroy(st);
      return fail(nullptr, CVH_ERR_HIP, "cvh_ppf_apply: hipMalloc: %s", hipGetErrorString(e));
    }
  }
  do {
    if ((e = hipMemcpyAsync(d, data + start, n * sizeof(double), hipMemcpyHostToDevice, st)) != hipSuccess) break;
    if ((e = cvh_launch_ppf(d, n, op, eps, st)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(data + start, d, n * sizeof(double), hipMemcpyDeviceToHost, st)) != hipSuccess) break;
    e = hipStreamSynchronize(st);
  } while (0);
  if (pooled) { (void)hipFreeAsync(d, st); (void)hipStreamSynchronize(st); } else (void)hipFree(d);
  (void)hipStreamDestroy(st);
  if (e != hipSuccess) return fail(nullptr, CVH_ERR_HIP, "cvh_ppf_apply: %s", hipGetErrorString(e));
  return CVH_OK;
}

extern "C" int cvh_launch_info(cvh_context *c, int phase, char *buf, int cap)
{
  if (!c || !buf || cap < 1 || (phase != 0 && phase != 1)) return CVH_ERR_ARG;
  if (phase == 1) {
    if (!c->pm_desc[0]) return fail(c, CVH_ERR_STATE, "cvh_launch_info: cvh_perona_malik has not run on this context");
    snprintf(buf, (size_t)cap, "%s", c->pm_desc);
    return CVH_OK;
  }
  // the CSV step as the next cvh_run / cvh_enqueue_steps would launch it: the launcher itself describes it (CVH_LAUNCH)
  CvhLaunchNote note{};
  {
    ResidentGeom rg;
    if (resident_geometry(c, &rg)) {
      const int rc = launch_resident(c, rg, 1, &note);
      if (rc != CVH_OK) return rc;
      snprintf(buf, (size_t)cap, "kernel=%s grid=%u block=%u lds_bytes=%u data_flow=4 tiles_y=%d tiles_x=%d tile_rows=%d chain=1 math=fast "
               "steps_per_launch=chunk", note.name, note.grid, note.block, note.lds, rg.tr, rg.tc, (c->h + rg.tr - 1) / rg.tr);
      return CVH_OK;
    }
  }
  const int rc = launch_one_step(c, current_buffer(c), c->enqueued, true, &note);
  if (rc != CVH_OK) return rc;
  const Geometry g = resolve_geometry(c);
  CvhStepArgs a;
  fill_args(c, &a, current_buffer(c), c->enqueued);
  snprintf(buf, (size_t)cap, "kernel=%s grid=%u block=%u lds_bytes=%u data_flow=%d wave_columns=%d strips=%d strip_rows=%d chain=%d "
           "wave_pol=%d math=%s steps_per_graph=%d", note.name, note.grid, note.block, note.lds, g.strip, g.tiles_x, g.tiles_y,
           g.strip_rows, a.chain ? 1 : 0, a.wave_pol, use_fast(c) ? "fast" : "strict", c->use_graph ? kGraphSteps : 0);
  return CVH_OK;
}

// Diagnostic (not part of include/chanvese_hip.h): the synchronisation words of the last resident launch: {error, 0, generation of
// the arrival line of tile 0 .. n-1, generation of the release line of tile 0 .. n-1}.
extern "C" int cvh_debug_resident_read(cvh_context *c, unsigned *out, int ngo)
{
  if (!c || !out || ngo < 0 || ngo > CVH_RESIDENT_MAX_TILES) return CVH_ERR_ARG;
  if (!c->d_resident) return fail(c, CVH_ERR_STATE, "no resident launch yet");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<unsigned> tmp(sizeof(CvhResident) / sizeof(unsigned));
  HIPCHK(c, hipMemcpy(tmp.data(), c->d_resident, sizeof(CvhResident), hipMemcpyDeviceToHost));
  out[0] = tmp[0]; out[1] = tmp[1] | (tmp[2] << 12) | (tmp[3] << 24);   // error; t_first | nit << 12 | steps_done as the kernel read it << 24
  for (int i = 0; i < ngo; ++i) { out[2 + i] = tmp[16 + (size_t)i * 16]; out[2 + ngo + i] = tmp[16 + (size_t)CVH_RESIDENT_MAX_TILES * 16 + (size_t)i * 16]; }
  return CVH_OK;
}

// Diagnostic (not part of include/chanvese_hip.h): copies the stamp buffer of "debug_times".
extern "C" int cvh_debug_read(cvh_context *c, unsigned long long *out, long max_words, long *words, int *nblocks)
{
  if (!c || !out || !words) return CVH_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  long n = (long)c->dbg_words < max_words ? (long)c->dbg_words : max_words;
  *words = n;
  if (nblocks) *nblocks = resolve_geometry(c).nblocks;
  if (n > 0 && c->d_dbg) HIPCHK(c, hipMemcpy(out, c->d_dbg, (size_t)n * 8, hipMemcpyDeviceToHost));
  return CVH_OK;
}
