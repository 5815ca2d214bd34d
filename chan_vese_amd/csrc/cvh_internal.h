// cvh_internal.h — shared declarations of the HIP implementation behind include/chanvese_hip.h.
// gfx950 (CDNA4, wave64) only.  Citations are file:line in the reference repository.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chanvese_hip.h"

#define CVH_BLOCK 256  // 4 waves of 64

// Ablation / diagnostic keys of cvh_set_option (not part of include/chanvese_hip.h; defaults are what ships):
//   "wave_occupancy" waves per SIMD the 1-pixel wave kernel's grid is sized for (3..5; default 5; the 2-pixel kernel is built for 3),
//   "wave_depth" rows per group (4, 8), "wave_prio" s_setprio progress equalisation (0 off, 1 quarters,
//   2-4 thresholds crowded to the end, 5 clock-paced), "wave_sync" workgroup barrier per group (1),
//   "wave_imgv" 16-byte image pieces (1), "wave_xcd" XCD-contiguous workgroup numbering (1),
//   "wave_skew" per-mille strip-length skew of the 1-pixel kernel (0), "wave_cls" class-major workgroup numbering (0 off,
//   1 = 2-pixel kernel (default), 2 = 1-pixel kernel too), "wave_cskew" per-mille strip-length skew between dispatch rounds (500),
//   "wave_pol" = 2 (diagnostic value of the public option: plain stores with non-temporal loads),
//   "chain" fixed-point chained sums + deferred bookkeeping in the wave kernels (1), "far_terms" terms of the far-field series (5; 4), "wave_lds_cap", "wave_rev", "debug_times" (per-wave stamps
//   read with cvh_debug_read, tools/wave_timeline.py), "res_straight" (1; 0 = csv_resident_kernel's generic march whatever the tile height).

// Device-resident scalar state of one context.  Written only by the finalising workgroup
// of a kernel and read by the next kernel on the same stream (kernel boundary = visibility).
struct CvhState {
  double c1[CVH_MAX_CHANNELS];  // region means of the current u, used by the next step (:973)
  double c2[CVH_MAX_CHANNELS];  // (:974)
  double norm;                  // ||u_diff||_2 of the last executed step (:993)
  double stop_cond;             // (unused: the stop condition is a launch argument, CvhStepArgs::stop_cond)
  int steps_done;               // iterations executed since cvh_reset_run
  int stopped;                  // sticky: stop rule fired (:1000); later launches are no-ops
  unsigned ticket;              // arrival counter of the in-kernel finalisation
  int pending;                  // chain mode: the iteration of the last launch still awaits its bookkeeping (norm, stop test)
};

// Chain mode of the 2-pixel wave kernel (csv_wave2_kernel.hip): the two sums the NEXT iteration needs, sum (H - 1/2)
// and sum I (H - 1/2), are accumulated as 64-bit FIXED-POINT integers with agent-scope atomic adds -- integer addition
// is associative, so the result is bitwise reproducible whatever the arrival order -- into one of four rotating sets
// of 16 or 32 shards per sum (same-address atomics serialise at ~10 ns each; 765 workgroups over 32 shards do not queue).  Launch e
// reads set (e mod 4), adds into set (e+1 mod 4) and clears set (e+2 mod 4); nothing waits for a last workgroup.
// A set holds 64 integers: (1 + C) sums x 64 / (1 + C) shards (chain_device.h).
constexpr int CVH_CHAIN_SETS = 4;
struct CvhChainAcc { long long v[CVH_CHAIN_SETS][64]; };

// Resident kernel (csv_resident_kernel.hip): the words its workgroups synchronise on, zeroed before every launch.
constexpr int CVH_RESIDENT_MAX_TILES = 256;     // the master's eight waves watch 32 arrival lines each (one tile per CU: 256 on MI355X)
struct CvhResident {
  int error;            // a bounded wait gave up (a workgroup was not resident, or a fault): the launch drains, the host reports it
  unsigned pad[15];
  // three 16-byte pieces per tile {generation, 0, payload}, written with agent-scope stores: sum u_diff^2 of the tile (double), then its
  // fixed-point sums of H - 1/2 and I (H - 1/2) (chain_device.h's integers): the tile has finished iteration generation - 1 of the launch.
  // PIECE-MAJOR -- piece p of tile t at entry p * CVH_RESIDENT_MAX_TILES + t -- so that a poll instruction of the master reads 64 neighbouring
  // entries (1 KiB) and not 16 bytes of 64 different lines.  The master adds the integers itself (exact, order-free): no atomics inside the
  // launch, and the arrival does not wait for the border stores (arrivals on distinct addresses do not serialise)
  unsigned flag[CVH_RESIDENT_MAX_TILES * 16];
  // one 64-byte line per tile, written by the master as two 16-byte stores {generation, leave, c1} {generation, leave, c2}: the release
  // behind iteration generation - 1 and the region means of the level set it produced
  unsigned go[CVH_RESIDENT_MAX_TILES * 16];
  // one 64-byte line per tile {generation}: the tile's borders of iteration generation - 1 have reached memory (its neighbours wait for this)
  unsigned hflag[CVH_RESIDENT_MAX_TILES * 16];
};

// Sums carried per workgroup and reduced in a fixed order (deterministic):
//   [0] sum H(u)  [1] sum (1-H(u))  [2..2+C) sum I_k H  [2+C..2+2C) sum I_k (1-H)  [2+2C] sum u_diff^2
__host__ __device__ constexpr int cvh_nsums(int C) { return 3 + 2 * C; }

// What a launcher WOULD launch (cvh_launch_info, include/chanvese_hip.h): filled at the launch site itself, by the
// same code path that selects the kernel flavour and the grid (CVH_LAUNCH below) -- nothing re-derives the choice.
struct CvhLaunchNote { char name[112]; unsigned grid, block, lds; };
void cvh_fill_note(CvhLaunchNote *note, unsigned grid, unsigned block, size_t lds, const char *fmt, ...);

struct CvhStepArgs {
  const double *u_in;
  double *u_out;
  const uint8_t *img[CVH_MAX_CHANNELS];
  unsigned img_stride;           // the planes are img[0] + k * img_stride (one slab)
  CvhState *st;
  double *partials;  // [nblocks][nsums]
  double *trace;     // [trace_cap][2C+1] or null
  int trace_cap;
  int h, w;
  int tiles_x, tiles_y;
  int nparts;        // number of partial rows the finaliser must add
  int fused_finalize;
  double alpha, beta, gamma;  // dt*mu, dt*(1/C), -nu*dt : addWeighted form of :985
  double eps;
  double lambda1[CVH_MAX_CHANNELS], lambda2[CVH_MAX_CHANNELS];
  // FAST flavour only
  const double *atan_tab;        // [2][CVH_ATAN_N]: atan(i/(N-1)) and pi/2 - atan(i/(N-1))
  double inv_eps;                // 1/eps
  double dk1, dk2;               // pi/eps, pi*eps : delta_eps(u) = 1 / (dk1*u^2 + dk2)
  double npix;                   // h*w
  double sum_img[CVH_MAX_CHANNELS];  // exact integer sums of the planes (complements are derived)
  double stop_cond;              // tol * ||mean_k I_k||_2 (:959), the value also held in CvhState
  double far_k[5], far_thr;      // wave kernels, FAST: far-field series of atan(eps/u)/pi (5 terms) and its threshold (32 eps)
  int derive_complement;         // sums [1] and [2+C..] are N - sum H, sum I - sum I H; 2: sums [0], [2..] are of H - 1/2
  int tile_rows;                 // rows per tile of the step kernel
  int use_lut;
  int use_dma;                   // LDS-DMA tile loader (even widths)
  int strip_rows;                // rows per workgroup (strip kernel) / per wave (wave kernel)
  const double *atan2_tab;       // [CVH_ATAN2_N]: (pi/4 + atan((j-128)/128)) / pi
  int wave_minw;                 // waves per SIMD the wave kernel is compiled for (5..8)
  unsigned long long *dbg_times; // diagnostic: per-wave {start, end, hw id} stamps (100 MHz), or null
  int *host_status;              // pinned host memory {steps_done, stopped}: written by the finaliser, polled by the host
  double *dummy;                 // >= max(w, 64) doubles that nobody reads: target of masked-off lanes' stores
  int wave_imgv;                 // wave kernel: 16-byte image pieces through LDS (w % 16 == 0)
  int wave_xcd;                  // wave kernel: XCD-contiguous workgroup numbering
  int wave_rev;                  // diagnostic: oldest workgroups take the bottom strips
  const int *strip_bounds;       // wave kernel: first row of each strip, [tiles_y + 1]
  int wave_depth;                // wave kernel: rows of u kept in flight per lane (4 or 8)
  int wave_sync;                 // wave kernel: workgroup barrier every 4 rows
  int wave_prio;                 // progress-based s_setprio in the wave kernel
  int state32;                   // 2-pixel wave kernel: the level set lives in HBM as float (option "state" = 32; u_in / u_out then point at floats)
  int near_switch;               // FAST wave / resident kernels: a wave that met near-field pixels runs its next group of rows in the table form of H_eps
  int wave_lds_cap;              // pad the LDS request so that at most wave_minw workgroups fit a CU
  CvhChainAcc *chain;            // chain mode (2-pixel wave kernel, FAST): fixed-point sum sets, or null
  int chain_phase;               // set this launch reads
  int chain_pb;                  // set that held the sums of u when the run counter was last reset (flush: set = pb + steps_done)
  double chain_scale[4], chain_inv[4];   // powers of two: fixed-point scale of sum (H-1/2), sum I_k (H-1/2) and their inverses
  double *chain_s4;              // [2][nparts] per-workgroup sum u_diff^2 rows, by launch parity
  int wave_pol;                  // 2-pixel wave kernel: cache policy of the level-set rows (wave2_device.h): 1 write-through stores, 0 plain
  int wave_cls;                  // 2-pixel wave kernel: workgroups per XCD per dispatch round (= CUs per XCD); > 0 numbers the
                                 // workgroups class-major (round 0 of every XCD first), 0 = plain XCD-contiguous numbering
  CvhLaunchNote *note;           // host only: non-null = describe the launch instead of issuing it (CVH_LAUNCH)
  // resident kernel (csv_resident_kernel.hip): tiles_x x tiles_y tiles, one workgroup each
  CvhResident *resident;         // synchronisation words
  double *res_halo;              // [2][tiles][6 * 128] border rows / columns of every tile, by iteration parity
  int res_steps, res_poll_cap;   // iterations in this launch; polls before a wait gives up
  int res_t0;                    // index of the launch's first iteration inside the run (= iterations enqueued before it)
  int res_prio;                  // resident kernels: a wave lowers its priority with every quarter of its band (the two waves of a SIMD finish together)
  int res_band_rows;             // rows per wave when every tile has 8 x that many rows (2, 4, 8, 16: straight-line march), else 0
  int res_go_shift;              // 2^shift tiles of one XCD share a release line (csv_resident_kernel.hip, go_line)
};

// The ONE way a step / Perona-Malik kernel is launched: KERNEL may be a parenthesised template-id; the trailing
// arguments are a printf format + values that spell the instantiation as rocprofv3 prints it.
#define CVH_LAUNCH(KERNEL, GRID, LDS, S, A, ...)                                                  \
  do {                                                                                             \
    if ((A).note) cvh_fill_note((A).note, (unsigned)(GRID), CVH_BLOCK, (size_t)(LDS), __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(CVH_BLOCK), (LDS), (S), (A));                 \
  } while (0)
#define CVH_TF(b) ((b) ? "true" : "false")

#define CVH_ATAN_N 129
#define CVH_ATAN2_N 257

struct CvhPmArgs {
  const double *in;
  double *out;
  int h, w;
  int tiles_x, tiles_y;
  double K2;  // K*K (:520)
  double L;
  double invK2, L4;  // FAST flavour: 1/K^2, L/4
  int fast;
  int strip_rows;    // wave kernel: rows per wave
  int pol;           // 2-step wave kernel: 1 = write-through stores (the state planes fit the Infinity Cache), 0 = plain
  // resident kernel (pm_resident_kernel.hip): tiles_y x tiles_x tiles, one workgroup each, res_steps time steps in one launch
  CvhResident *resident;         // the error word
  double *res_halo;              // 2 x tiles x 1024 entries of 16 bytes {value, tag}: the tiles' borders, double-buffered by step parity
  unsigned res_serial;           // tag of this launch (entries left by earlier launches never match)
  int res_steps;
  int res_band_rows;             // rows per wave: 2, 4, 8 or 16 (tiles of 8 x that many rows)
  int res_prio;                  // a wave lowers its priority with every quarter of its band (csv_resident_kernel.hip, quarter_prio)
  int res_poll_cap;              // polls before a wait gives up
  unsigned long long *dbg_times; // diagnostic (option "debug_times", tools/pm_resident_timeline.py): 12 stamps per workgroup, or null
  CvhLaunchNote *note;   // host only: describe the launch instead of issuing it (CVH_LAUNCH)
};

// ---- launchers (csv_kernels.hip / pm_kernels.hip / misc_kernels.hip) ----
// rows-per-tile options of the step kernel
void cvh_step_grid(int h, int w, int tile_rows, int *tiles_x, int *tiles_y);
int cvh_step_max_blocks(int h, int w);
hipError_t cvh_launch_step(const CvhStepArgs &a, int channels, int fast, hipStream_t s);
int cvh_wave2_cols();
hipError_t cvh_launch_wave2(const CvhStepArgs &a, int channels, int fast, hipStream_t s);
hipError_t cvh_launch_chain_flush(const CvhStepArgs &a, int channels, hipStream_t s);
size_t cvh_pm_resident_lds_bytes();
int cvh_pm_resident_halo_doubles();
int cvh_pm_resident_blocks_per_cu();
hipError_t cvh_launch_pm_resident(const CvhPmArgs &a, hipStream_t s);
size_t cvh_resident_lds_bytes();
int cvh_resident_tile_w();
int cvh_resident_tile_hmax();
int cvh_resident_halo_doubles();
int cvh_resident_blocks_per_cu();
hipError_t cvh_launch_resident(const CvhStepArgs &a, hipStream_t s);   // cooperative launch, a.res_steps iterations in LDS
hipError_t cvh_launch_wave(const CvhStepArgs &a, int channels, int fast, hipStream_t s);
int cvh_wave_cols();
hipError_t cvh_launch_init_sums(const CvhStepArgs &a, int channels, int fast, int *nparts_out,
                                hipStream_t s);
hipError_t cvh_launch_finalize(const CvhStepArgs &a, int channels, int is_init, hipStream_t s);
int cvh_init_sum_blocks(int h, int w);

hipError_t cvh_launch_pm_load(const uint8_t *plane, double *state, size_t n, hipStream_t s);
hipError_t cvh_launch_pm_step(const CvhPmArgs &a, hipStream_t s);
hipError_t cvh_launch_pm_wave(const CvhPmArgs &a, hipStream_t s);
int cvh_pm_wave_k2_cols();
hipError_t cvh_launch_pm_wave_k2(const CvhPmArgs &a, hipStream_t s);   // TWO time steps per launch
int cvh_pm_wave_cols();
hipError_t cvh_launch_pm_store(const double *state, uint8_t *plane, size_t n, hipStream_t s);
void cvh_pm_grid(int h, int w, int *tiles_x, int *tiles_y);

hipError_t cvh_launch_contour(const double *u, uint8_t *out, int h, int w, hipStream_t s);
hipError_t cvh_launch_checkerboard(const double *sv, double *u, int h, int w, hipStream_t s);
hipError_t cvh_launch_image_sums(const uint8_t *const *planes, int channels, size_t n, unsigned long long *out /* [2 * channels], zeroed */,
                                 hipStream_t s);
hipError_t cvh_launch_mask(const double *u, uint8_t *mask, size_t n, int invert, hipStream_t s);
hipError_t cvh_launch_state_narrow(double *u, float *uf, size_t n, hipStream_t s);       // uf = (float)u, u = (double)uf
hipError_t cvh_launch_state_widen(const float *uf, double *u, size_t n, hipStream_t s);  // u = (double)uf
hipError_t cvh_launch_ppf(double *data, size_t n, int op, double eps, hipStream_t s);
hipError_t cvh_launch_separate(const uint8_t *img3, const double *u, uint8_t *sel3, size_t n,
                               int invert, hipStream_t s);
