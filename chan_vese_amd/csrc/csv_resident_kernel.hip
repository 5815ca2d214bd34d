// csv_resident_kernel.hip — cache-resident planes (gfx950, wave64, 1 channel, FAST arithmetic, chain-mode sums): the level set
// lives in LDS for a whole chunk of iterations.
//
// A 2048^2 level set is 32 MiB; the 256 CUs of an MI355X hold 40 MiB of LDS.  One cooperative launch cuts the plane into
// tiles_y x tiles_x tiles of <= 128 rows x 128 columns, ONE workgroup (8 waves) per tile and per CU, loads every tile into its
// CU's LDS once, runs `res_steps` iterations of src/main.cpp:963-1001 on it in place, and writes the tiles back at the end.
// What a launch per iteration pays every iteration -- the launch gap, the dispatch ramp, 7 rows of loads per 21-row strip before
// the first row computes, the drain of the last waves, every level-set byte through the memory system twice -- is paid once per
// chunk; what crosses workgroups per iteration is
//   * the tile's border rows / columns (the 7-point cross reaches 2 up / left, 1 down / right: 6 x 128 doubles per tile), through a
//     double-buffered global halo buffer, written and read with sc1 (agent scope);
//   * ONE grid barrier with a master.  A tile ARRIVES with three 16-byte agent-scope stores into its own 64-byte line {generation, sum
//     u_diff^2} {generation, sum H'} {generation, sum I H'} -- the last two as chain mode's fixed-point integers (chain_device.h) -- right
//     behind its reduction, before its border stores (those get their own signal line: only the neighbours wait for them).  Workgroup
//     0 is the master: its eight waves watch 32 arrival lines each without workgroup barriers, wave 0 adds the integers (exact,
//     order-free: the totals the per-launch path's atomic adds produce), books the iteration (norm, stop rule src/main.cpp:1000,
//     trace row) and RELEASES everybody with one line per workgroup that carries the leave bit and the region means of the new level
//     set (every workgroup polls its own line: same-address polling does not scale, tools/experiments/persist/README.md).  The stop
//     rule therefore fires at the reference's iteration with no extra iteration computed, and no atomic is issued inside the launch.
// Every wait is a bounded poll: a workgroup that gives up raises CvhResident::error and leaves, and so does everybody waiting
// for it -- the grid always drains; the host reports the error at the next synchronisation.
//
// Inside a tile: wave v owns a band of rows and all 128 columns (lane l <-> columns 2l, 2l + 1), marches down its band with
// u(i-1), u(i) in registers and u(i+1) and the x-neighbours read from the LDS tile one row ahead, and writes row i IN PLACE once it
// is computed.  Bands only meet at their first / last rows: every wave reads the two rows above its band and the row below it
// into registers before any wave writes (one workgroup barrier).  Column -1's normalised x-gradient, which lane 0 needs for
// kappa_x of column 0 and no lane owns, is computed for all rows of the tile by one thread per row before the march.
// The arithmetic of a pixel is csv_wave2_kernel.hip's FAST flavour operation by operation.
#include "csv_device.h"
#include "buffer_ops.h"
#include "wave_math.h"
#include "chain_device.h"
#include <type_traits>

using namespace cvh_dev;

namespace {

constexpr int RT_W = 128;                 // tile width: 64 lanes x 2 pixels
constexpr int RT_HMAX = 128;              // most rows a tile may have (LDS)
constexpr int RT_PITCH = 132;             // doubles per LDS row: tile columns -2 .. 129
// Poll cadences and waves per workgroup were A/B build macros in round 3 (profiles/r03_C4/resident_poll_variants.txt: s_sleep 1 between the
// master's polls and 3 between a workgroup's polls of its release line are a local optimum; 12 / 16 waves per workgroup: 17.4 / 33 us).
constexpr int kMasterSleep = 1, kReleaseSleep = 3;
constexpr int RT_WAVES = 8, RT_THREADS = 64 * RT_WAVES;
constexpr int RT_HALO = 6 * RT_W;         // doubles a tile publishes per iteration: bottom 2 rows, top row, right 2 columns, left column


typedef double double2_t __attribute__((ext_vector_type(2)));

struct ResSmem {
  static constexpr int NS = cvh_nsums(1);
  static constexpr int off_u = 0;                                           // (RT_HMAX + 3) rows x RT_PITCH: tile rows -2 .. TH
  static constexpr int off_img = off_u + (RT_HMAX + 3) * RT_PITCH;          // RT_HMAX x 128 bytes
  static constexpr int off_lut = off_img + RT_HMAX * RT_W / 8;              // 256 x {term, I}
  static constexpr int off_atan = off_lut + 512;                            // CVH_ATAN2_N (+1 pad)
  static constexpr int off_nxl = off_atan + CVH_ATAN2_N + 1;                // RT_HMAX: normalised x-gradient of column -1
  static constexpr int off_red = off_nxl + RT_HMAX;                         // RT_WAVES x NS
  static constexpr int off_flag = off_red + RT_WAVES * NS;
  static constexpr int doubles = off_flag + 4 + (RT_WAVES + 1) / 2 + 1 + RT_WAVES / 2;   // 3 broadcast doubles, 12 ints, the master's RT_WAVES ints
  static constexpr size_t bytes = (size_t)doubles * sizeof(double);
};
static_assert(32 * RT_WAVES >= CVH_RESIDENT_MAX_TILES, "the master's waves watch 32 arrival lines each");
static_assert(ResSmem::bytes <= 160 * 1024, "the tile, its halo ring and the tables must fit one CU's LDS");

__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Sum over the workgroup's 8 waves in a fixed order; every thread returns the total of sum `s` it asked for (s < NS).
template <int NS>
__device__ __forceinline__ void block_reduce8(double (&acc)[NS], double *sred /*[8*NS]*/, double (&total)[NS])
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double v[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) v[s] = wave_sum(acc[s]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) sred[wave * NS + s] = v[s];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    double t = sred[s];
#pragma unroll
    for (int wv = 1; wv < RT_WAVES; ++wv) t += sred[wv * NS + s];   // fixed order
    total[s] = t;
  }
  __syncthreads();
}

// 16-byte agent-scope (sc1) accesses to the synchronisation lines: one lane, one transaction.
typedef unsigned int u32x4r_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4r_t ld_line16(const void *base, unsigned byte_off)
{
  return __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(base, 0x7fffffffu), byte_off, 0u, 16 /* sc1 */);
}
__device__ __forceinline__ void st_line16(void *base, unsigned byte_off, unsigned w0, unsigned w1, double d)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(d);
  __builtin_amdgcn_raw_buffer_store_b128(u32x4r_t{w0, w1, (unsigned)b, (unsigned)(b >> 32)}, make_rsrc(base, 0x7fffffffu), byte_off, 0u, 16 /* sc1 */);
}
__device__ __forceinline__ void st_line16_u64(void *base, unsigned byte_off, unsigned w0, unsigned w1, unsigned long long b)
{
  __builtin_amdgcn_raw_buffer_store_b128(u32x4r_t{w0, w1, (unsigned)b, (unsigned)(b >> 32)}, make_rsrc(base, 0x7fffffffu), byte_off, 0u, 16 /* sc1 */);
}
__device__ __forceinline__ long long line16_i64(u32x4r_t v) { return (long long)(((unsigned long long)v.w << 32) | v.z); }
__device__ __forceinline__ double line16_f64(u32x4r_t v) { return __longlong_as_double((long long)(((unsigned long long)v.w << 32) | v.z)); }

// Thread 0 polls this workgroup's release line until both halves carry generation >= `gen` (bounded); the workgroup meets at a
// barrier.  Returns the leave bit (or -1: gave up) and the region means the line carries.
__device__ __forceinline__ int wg_wait_go(const CvhResident *rs, int bid, int gen, const CvhStepArgs &a, double *s_bc /*[4]*/, double &c1, double &c2)
{
  if (threadIdx.x == 0) {
    int res = -1;
    double m1 = 0.0, m2 = 0.0;
    // (one poll in flight: two or four in flight sample the line more often but cost 0.5 / 0.7 us per iteration at 2048^2 -- the
    // polls of 256 workgroups compete with the arrivals and the release for the same fabric)
    for (int i = 0; i < a.res_poll_cap; ++i) {
      const u32x4r_t ga = ld_line16(rs->go, (unsigned)bid * 64u), gb = ld_line16(rs->go, (unsigned)bid * 64u + 16u);
      if (ga.x == gb.x && ga.x >= (unsigned)gen && ga.x != 0xffffffffu) { res = (int)(ga.y & 1u); m1 = line16_f64(ga); m2 = line16_f64(gb); break; }
      if ((i & 15) == 15 && ld_agent((const unsigned *)&rs->error) != 0u) break;
      if (i >= 64) __builtin_amdgcn_s_sleep(16); else if (i >= 2) __builtin_amdgcn_s_sleep(kReleaseSleep);
    }
    if (res < 0) st_agent(const_cast<int *>(&rs->error), 1);
    s_bc[0] = (double)res; s_bc[1] = m1; s_bc[2] = m2;
  }
  __syncthreads();
  const int res = (int)s_bc[0];
  c1 = s_bc[1]; c2 = s_bc[2];
  __syncthreads();
  return res;
}

// NRT: rows per wave when every tile has exactly 8 * NRT rows (2, 4, 8, 16: the march is straight-line code, row offsets are immediates, the
// band's last row is known at compile time); 0: any tile height (bands of TH / 8 rows, a loop over groups of four rows)
template <int NRT>
__global__ __launch_bounds__(RT_THREADS, 1) void csv_resident_kernel(const CvhStepArgs a)
{
  using L = ResSmem;
  constexpr int NS = L::NS;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *su = smem + L::off_u;
  unsigned char *simg = reinterpret_cast<unsigned char *>(smem + L::off_img);
  double *slut = smem + L::off_lut;
  double *satan = smem + L::off_atan;
  double *snxl = smem + L::off_nxl;
  double *sred = smem + L::off_red;
  double *s_bc = smem + L::off_flag;      // 4 doubles of broadcast scratch
  int *s_flag = (int *)(s_bc + 3);
  int *s_mflag = s_flag + 12;             // master workgroup: generation each wave's share of the arrivals is complete for
  constexpr unsigned kLutAddr = (unsigned)(L::off_lut * sizeof(double));   // LDS byte address of the region-term table (the dynamic block starts at 0)
  if (!lds_base_is_zero(smem)) __builtin_trap();                          // (folds away: no static LDS in this kernel)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < RT_WAVES) s_mflag[tid] = 0;
  const int h = a.h, w = a.w;
  CvhResident *const rs = a.resident;
  // sticky stop flag of an EARLIER launch (src/main.cpp:1000): read at agent scope -- every workgroup must see the same value, and a
  // cooperative launch is dispatched through its own queue (a cached copy of the word is not to be trusted here)
  if (tid == 0) s_flag[0] = __hip_atomic_load(&a.st->stopped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int stopped_before = s_flag[0];
  __syncthreads();
  if (stopped_before != 0) return;
  // index of this launch's first iteration inside the run: a launch ARGUMENT (the host's count of the iterations it has enqueued since
  // the run counter was reset), not read from the state block: a stale cached copy of that word would shift every trace row
  const int t_first = a.res_t0;

  // ---- this workgroup's tile
  const int tr = a.tiles_y, tc = a.tiles_x, ntiles = tr * tc;
  const int bid = (int)blockIdx.x;
  const int ty = bid / tc, tx = bid % tc;
  const int r0 = (int)(((long)h * ty) / tr), r1 = (int)(((long)h * (ty + 1)) / tr);
  const int TH = r1 - r0;                              // <= RT_HMAX (host)
  const int c0 = tx * RT_W;
  const int TWv = (w - c0) < RT_W ? (w - c0) : RT_W;   // even (host: w even)
  // LDS address of tile element (row r in -2 .. TH, column c in -2 .. 129)
  auto S = [&](int r, int c) -> double * { return su + (r + 2) * RT_PITCH + (c + 2); };

  // ---- once per launch: tables, image tile, level-set tile with its halo ring straight from the plane (clamped = BORDER_REPLICATE)
  for (int q = tid; q < CVH_ATAN2_N; q += RT_THREADS) satan[q] = a.atan2_tab[q];
  if ((w & 15) == 0) {                                                   // rows are 16-byte aligned and the tile's width is a multiple of 16
    for (int q = tid; q < TH * (RT_W / 16); q += RT_THREADS) {           // 16-byte pieces of the image tile
      const int r = q / (RT_W / 16), p = q % (RT_W / 16);
      const int col = c0 + 16 * p < w ? c0 + 16 * p : w - 16;            // pieces beyond the image: no lane reads them
      const uint4 v = *reinterpret_cast<const uint4 *>(a.img[0] + (size_t)(r0 + r) * w + col);
      *reinterpret_cast<uint4 *>(simg + r * RT_W + 16 * p) = v;
    }
  } else {
    for (int q = tid; q < TH * RT_W; q += RT_THREADS) {                  // other widths: byte by byte (once per launch)
      const int r = q / RT_W, c = q % RT_W;
      simg[q] = a.img[0][(size_t)(r0 + r) * w + clampi(c0 + c, 0, w - 1)];
    }
  }
  for (int q = tid; q < (TH + 3) * RT_PITCH; q += RT_THREADS) {
    const int r = q / RT_PITCH - 2, c = q % RT_PITCH - 2;
    const int gr = clampi(r0 + r, 0, h - 1), gc = clampi(c0 + c, 0, w - 1);
    su[q] = a.u_in[(size_t)gr * w + gc];
  }
  // Sum sets (chain_device.h): the launch reads set p0 (the sums of the level set it starts from) and leaves set p0 + executed filled and
  // set p0 + executed + 1 clear -- the per-launch invariant -- when it ends.  In between the sums do not touch the sets: every tile hands
  // its fixed-point integers to the master with its arrival line, and the master adds them (integer addition: exact, order-free, the
  // very totals the per-launch path's atomic adds produce).
  // region means of the level set the launch starts from (later iterations get theirs with the release)
  double c1, c2;
  {
    const long long entry = a.chain->v[a.chain_phase & 3][lane];
    double m1[1], m2[1];
    chain_means<1>(a, entry, m1, m2);
    c1 = m1[0]; c2 = m2[0];
  }
  __syncthreads();

  const double l1 = a.lambda1[0], l2 = a.lambda2[0];
  const double eps = a.eps, eps2 = eps * eps;
  const FarCoef fc = {a.far_k[0], a.far_k[1], a.far_k[2], a.far_k[3], a.far_k[4], a.far_thr};
  // this wave's band of tile rows
  const int rb0 = NRT ? NRT * wave : (TH * wave) / RT_WAVES, rb1 = NRT ? rb0 + NRT : (TH * (wave + 1)) / RT_WAVES;
  const int ca = 2 * lane;                                  // tile column of pixel a
  const bool lane_valid = ca < TWv;
  const double fxa = (c0 + ca <= 0) ? 0.0 : 1.0;            // kappa_x(i, 0) = 0 (src/main.cpp:371)
  double *const halo_mine[2] = {a.res_halo + (size_t)bid * RT_HALO, a.res_halo + ((size_t)ntiles + bid) * RT_HALO};
  auto norm = [&](double fwd, double bwd, double c) -> double { return normalised4(fwd, bwd, c + c); };

  // diagnostic stamps (option "debug_times", tools/resident_timeline.py): 12 words per workgroup, taken around iteration kStampIt
  constexpr int kStampIt = 3;
  auto stamp = [&](int it_now, int it_want, int slot) {
    if (a.dbg_times && it_now == it_want && tid == 0) a.dbg_times[(size_t)bid * 12 + slot] = __builtin_amdgcn_s_memrealtime();
  };
  int executed = 0;
  bool gave_up = false;
  const int nit = a.res_steps;
  if (bid == 0 && tid == 0) { rs->pad[0] = (unsigned)t_first; rs->pad[1] = (unsigned)nit; rs->pad[2] = (unsigned)a.st->steps_done; }   // (diagnostic record of the launch)
  for (int it = 0; it < nit; ++it) {
    const int phase = (a.chain_phase + it) & 3;
    // ---- the release behind iteration it - 1: leave bit and the region means of u(it); the halos were fetched while waiting
    if (it > 0) {
      const int go = wg_wait_go(rs, bid, it, a, s_bc, c1, c2);
      if (go < 0) { gave_up = true; break; }
      if (go & 1) break;
    }
    stamp(it, kStampIt, 0); stamp(it, kStampIt + 1, 8);       // released into this iteration
    // ---- table of the variance term (:307-310, :979, :985)
    if (tid < 256) {
      const double v = (double)tid;
      const double d1 = v - c1, d2 = v - c2;
      const double reg = (d2 * d2) * l2 - (d1 * d1) * l1;
      slut[2 * tid] = __builtin_fma(reg, a.beta, a.gamma);
      slut[2 * tid + 1] = v;
    }
    // ---- normalised x-gradient of tile column -1, one thread per row (nobody owns that column; lane 0 needs it for column 0)
    if (tid >= 256 && tid - 256 < TH) snxl[tid - 256] = norm(*S(tid - 256, 0), *S(tid - 256, -2), *S(tid - 256, -1));
    // ---- every wave takes the rows around its band into registers before any wave writes
    const double2_t um2_0 = *reinterpret_cast<const double2_t *>(S(rb0 - 2, ca));
    double2_t um = *reinterpret_cast<const double2_t *>(S(rb0 - 1, ca));
    double2_t u0 = *reinterpret_cast<const double2_t *>(S(rb0, ca));
    const double2_t ubot = *reinterpret_cast<const double2_t *>(S(rb1, ca));
    double uw = *S(rb0, ca - 1), ue = *S(rb0, ca + 2);
    const double2_t u1st = *reinterpret_cast<const double2_t *>(S(rb0 + 1 < rb1 ? rb0 + 1 : rb0, ca));   // row rb0 + 1 (own band, if it has one)
    __syncthreads();
    stamp(it, kStampIt, 1);                                    // table in LDS, band borders in registers

    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0;
    if (rb1 > rb0) {
      double nypa = norm(u0.x, um2_0.x, um.x), nypb = norm(u0.y, um2_0.y, um.y);   // ny at row rb0 - 1
      if (r0 + rb0 == 0) {   // kappa_y(0, .) = 0 (:372): ny_prev := row 0's own ny, the very expression the row uses
        const double2_t up0 = (rb0 + 1 < rb1) ? u1st : ubot;
        nypa = norm(up0.x, um.x, u0.x); nypb = norm(up0.y, um.y, u0.y);
      }
      auto pixel = [&](double c, double n_, double s_, double nx, double nxl, double fx, double &nyp, int byte, double &ud_out,
                       double &Ik_out) -> double {
        const double ny = norm(s_, n_, c);
        const double kappa = __builtin_fma(nx - nxl, fx, ny - nyp);
        const double2_t e = lds_read_d2(kLutAddr + (unsigned)byte);      // `byte`: the entry's byte offset (sample x 16)
        double ud = __builtin_fma(kappa, a.alpha, e.x);                  // :985
        const double qd = __builtin_fma(c, c, eps2) * a.dk1;             // 1 / delta_eps(u)
        const double q0 = __builtin_amdgcn_rcp(qd);
        const double er = __builtin_fma(-qd, q0, 1.0);
        ud = ud * __builtin_fma(__builtin_fma(er, er, er), q0, q0);      // :992
        nyp = ny;
        ud_out = ud; Ik_out = e.y;
        return c + ud;                                                   // :994
      };
      // One row: everything up to the new values and the far-field form of H - 1/2 on every lane (branch-free); the lanes near
      // the contour are corrected per group of four rows (csv_wave2_kernel.hip, DEFER)
      double2_t keep[4];
      int smp_keep[4];
      unsigned long long near_mask[4];       // lanes below the far-field threshold, per row of the current group (csv_wave2_kernel.hip: four independent masks are the fastest form)
      // (no branch inside a row: a group of four rows is one basic block and hipcc overlaps the rows' dependent chains)
      // (rel = row inside the band: a constant in the straight-line flavours, where every address below is base + immediate)
      double *const pb = S(rb0, ca);
      const unsigned char *const simg_b = simg + rb0 * RT_W + ca;
      const double *const snxl_b = snxl + rb0;
      // Which form of H_eps a BAND takes this iteration is decided per wave from its first row (csv_wave2_kernel.hip, near_strip): where
      // most of that row is below the far-field threshold the wave runs the copy of the march that takes the table form of every
      // pixel behind the rows of a group (valid for any u, nothing to correct) -- one form per pixel instead of three.
      const bool near_band = a.near_switch &&
          __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane_valid && (fabs(u0.x) < fc.thr || fabs(u0.y) < fc.thr))) >= 32;
      auto row = [&](int rel, int k, auto near_tag) {
        constexpr bool NEARFORM = decltype(near_tag)::value;
        const bool lastrow = rel + 1 >= (NRT ? NRT : rb1 - rb0);     // wave-uniform; a constant in the straight-line flavours
        double2_t up;
        double uw_n = 0.0, ue_n = 0.0;
        // below the band's last row: the copy taken before the march -- the band below may have rewritten its first row already
        if (NRT) {
          if (lastrow) up = ubot;
          else { up = *reinterpret_cast<const double2_t *>(pb + (rel + 1) * RT_PITCH); uw_n = pb[(rel + 1) * RT_PITCH - 1]; ue_n = pb[(rel + 1) * RT_PITCH + 2]; }
        } else {
          const double2_t up_l = *reinterpret_cast<const double2_t *>(pb + (rel + 1) * RT_PITCH);
          uw_n = pb[(rel + 1) * RT_PITCH - 1]; ue_n = pb[(rel + 1) * RT_PITCH + 2];   // (unused behind the band's last row)
          up = double2_t{lastrow ? ubot.x : up_l.x, lastrow ? ubot.y : up_l.y};
        }
        const int smp = (int)*reinterpret_cast<const unsigned short *>(simg_b + rel * RT_W);
        const int ba = (int)byte_x16<0>((unsigned)smp), bb = (int)byte_x16<1>((unsigned)smp);   // sample x 16 in one SDWA instruction each (wave_math.h)
        const double nxl0 = snxl_b[rel];
        // x-gradients first: nx(b) is the west gradient of lane + 1's a (DPP), nx(a) the west gradient of b
        const double nxa = norm(u0.y, uw, u0.x);
        const double nxb = norm(ue, u0.x, u0.y);
        const double nxla = dpp_from_left_or(nxl0, nxb);          // lane 0 has no lane to its left: it keeps the pre-pass's value
        double uda, udb, Ia, Ib;
        const double va = pixel(u0.x, um.x, up.x, nxa, nxla, fxa, nypa, ba, uda, Ia);
        const double vb = pixel(u0.y, um.y, up.y, nxb, nxa, 1.0, nypb, bb, udb, Ib);
        keep[k] = double2_t{va, vb};
        smp_keep[k] = smp;
        // in place: every reader of the old row i has it in registers.  Lanes beyond a ragged tile's width write cells nobody owns
        // (the halo column among them: it was read a row ahead and is refreshed before the next iteration)
        *reinterpret_cast<double2_t *>(pb + rel * RT_PITCH) = keep[k];
        if (!NEARFORM) {
          const double hva = heaviside_centred_far(va, fc), hvb = heaviside_centred_far(vb, fc);
          near_mask[k] = __builtin_amdgcn_ballot_w64(fabs(va) < fc.thr || fabs(vb) < fc.thr);
          acc[0] += hva; acc[0] += hvb;
          acc[2] = __builtin_fma(Ia, hva, acc[2]); acc[2] = __builtin_fma(Ib, hvb, acc[2]);
        }
        acc[4] = __builtin_fma(uda, uda, acc[4]); acc[4] = __builtin_fma(udb, udb, acc[4]);
        um = u0; u0 = up; uw = uw_n; ue = ue_n;
      };
      auto correct = [&](int k, auto near_tag) {
        constexpr bool NEARFORM = decltype(near_tag)::value;
        if (NEARFORM || near_mask[k] != 0ull) {
          const double xa = keep[k].x, xb = keep[k].y;
          double da, db;
          if (NEARFORM) {   // the rows added nothing for H
            da = heaviside_centred_near(xa, a.inv_eps, satan); db = heaviside_centred_near(xb, a.inv_eps, satan);
          } else {
            da = (fabs(xa) < fc.thr) ? heaviside_centred_near(xa, a.inv_eps, satan) - heaviside_centred_far(xa, fc) : 0.0;
            db = (fabs(xb) < fc.thr) ? heaviside_centred_near(xb, a.inv_eps, satan) - heaviside_centred_far(xb, fc) : 0.0;
          }
          acc[0] += da; acc[0] += db;
          acc[2] = __builtin_fma((double)(smp_keep[k] & 0xff), da, acc[2]);
          acc[2] = __builtin_fma((double)(smp_keep[k] >> 8), db, acc[2]);
        }
      };
      // Two waves share a SIMD and at equal priority the arbiter serves the OLDER one first: it is through its band after 7.3 us, the younger
      // then runs alone -- a single wave hides no latency -- until 12 us (profiles/r03_C4/resident_timeline_2048.txt).  A wave lowers its priority
      // with every quarter of its band (as the per-launch kernels do by quarters of their strips): whoever is AHEAD yields, the two leapfrog by
      // groups of rows and finish together, two waves overlapping to the end.  (option "res_prio", default on)
      auto quarter_prio = [&](int done, int of) {
        if (!a.res_prio) return;
        const int q = of >= 4 ? (4 * done) / of : done;      // quarters of the band behind this wave
        if (q <= 0) __builtin_amdgcn_s_setprio(3);
        else if (q == 1) __builtin_amdgcn_s_setprio(2);
        else if (q == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      };
      auto march = [&](auto near_tag) {
        constexpr bool NEARFORM = decltype(near_tag)::value;
        if (NRT >= 4) {
#pragma unroll
          for (int g = 0; g < NRT / 4; ++g) {
            quarter_prio(4 * g, NRT);
#pragma unroll
            for (int k = 0; k < 4; ++k) row(4 * g + k, k, near_tag);
            if (NEARFORM || (near_mask[0] | near_mask[1] | near_mask[2] | near_mask[3]) != 0ull) {
#pragma unroll
              for (int k = 0; k < 4; ++k) correct(k, near_tag);
            }
          }
        } else if (NRT == 2) {
          row(0, 0, near_tag); correct(0, near_tag);
          row(1, 0, near_tag); correct(0, near_tag);
        } else {
          int rel = 0;
          for (; rel + 4 <= rb1 - rb0; rel += 4) {
            quarter_prio(rel, rb1 - rb0);
#pragma unroll
            for (int k = 0; k < 4; ++k) row(rel + k, k, near_tag);
            if (NEARFORM || (near_mask[0] | near_mask[1] | near_mask[2] | near_mask[3]) != 0ull) {
#pragma unroll
              for (int k = 0; k < 4; ++k) correct(k, near_tag);
            }
          }
          for (; rel < rb1 - rb0; ++rel) { row(rel, 0, near_tag); correct(0, near_tag); }
        }
      };
      if (near_band) march(std::true_type{}); else march(std::false_type{});
      if (a.res_prio) __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int s = 0; s < NS; ++s) acc[s] = lane_valid ? acc[s] : 0.0;   // lanes beyond the image contribute nothing
    }
    stamp(it, kStampIt, 2);                                    // (thread 0's wave) march done
    double total[NS];
    block_reduce8<NS>(acc, sred, total);    // (its barriers also order the tile writes before the border reads below)
    stamp(it, kStampIt, 3);                                    // all waves done, sums reduced
    executed = it + 1;
    const unsigned gen = (unsigned)(it + 1);

    // ---- arrive: three 16-byte lines {generation, payload}: sum u_diff^2, and the fixed-point sums the next iteration's means come from
    // (distinct addresses: 256 arrivals on one counter serialise for 6 us).  The arrival does not wait for the border stores below: the
    // master needs the sums only, the neighbours get their own signal.
    if (tid < 3) {
      const unsigned long long payload = tid == 0 ? (unsigned long long)__double_as_longlong(total[4])
                                       : tid == 1 ? (unsigned long long)__double2ll_rn(total[0] * a.chain_scale[0])
                                                  : (unsigned long long)__double2ll_rn(total[2] * a.chain_scale[1]);
      st_line16_u64(rs->flag, (unsigned)bid * 64u + 16u * (unsigned)tid, gen, 0u, payload);
    }
    // ---- the tile's border for the neighbours, then the border signal
    {
      double *const hb = halo_mine[it & 1];
      for (int q = tid; q < 6 * RT_W; q += RT_THREADS) {
        const int piece = q / RT_W, k = q % RT_W;
        double v;
        if (piece < 2) v = *S(TH - 2 + piece, k);                    // bottom two rows   (TH >= 2: host)
        else if (piece == 2) v = *S(0, k);                           // top row
        else if (piece < 5) v = *S(k < TH ? k : 0, TWv - 5 + piece);  // right two columns: TWv - 2, TWv - 1
        else v = *S(k < TH ? k : 0, 0);                              // left column
        st_agent_f64(hb + q, v);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(it, kStampIt, 4);                                    // borders have reached memory
    if (tid == 0) st_line16_u64(rs->hflag, (unsigned)bid * 64u, gen, 0u, 0ull);

    // ---- workgroup 0 is the barrier's master.  Its eight waves watch 32 arrival lines each (lanes 0-31: the norm and sum-H' pieces,
    // lanes 32-63: the sum-I H' piece) WITHOUT workgroup barriers: a wave whose share is complete leaves its partial sums and the
    // generation in LDS and goes on; wave 0 collects the eight, books the iteration and releases everybody.  (Single-wave code is
    // latency-bound -- 8 cycles an instruction: everything here is the critical path of 255 waiting workgroups.)
    if (bid == 0) {
      const int b = wave * 32 + (lane & 31);                    // (host: ntiles <= 32 * RT_WAVES)
      const bool have = b < ntiles;
      bool done = false;
      u32x4r_t fa = {0u, 0u, 0u, 0u}, fb = {0u, 0u, 0u, 0u};
      // (one poll in flight: two in flight, half a round trip apart, sample a line twice as often and are SLOWER -- 2048^2 14.60 -> 15.04 us,
      // 1024^2 7.89 -> 8.19: reads of a line that is being written get in the way of the write)
      for (int round = 0; round < a.res_poll_cap; ++round) {
        if (have) {
          fa = ld_line16(rs->flag, (unsigned)b * 64u + (lane < 32 ? 0u : 32u));
          if (lane < 32) fb = ld_line16(rs->flag, (unsigned)b * 64u + 16u);
        }
        const bool ok = !have || (fa.x >= gen && fa.x != 0xffffffffu && (lane >= 32 || (fb.x >= gen && fb.x != 0xffffffffu)));
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) { done = true; break; }
        if ((round & 15) == 15 && ld_agent((const unsigned *)&rs->error) != 0u) break;
        __builtin_amdgcn_s_sleep(kMasterSleep);
      }
      if (done) {
        const double ws = wave_sum((have && lane < 32) ? line16_f64(fa) : 0.0);                       // fixed order: lane = tile
        const long long r = row16_sum_i64(!have ? 0ll : lane < 32 ? line16_i64(fb) : line16_i64(fa));
        const long long w0 = read_lane_i64(r, 0) + read_lane_i64(r, 16), w1 = read_lane_i64(r, 32) + read_lane_i64(r, 48);
        if (lane == 0) {
          sred[wave * NS] = ws;
          reinterpret_cast<long long *>(sred)[wave * NS + 1] = w0;
          reinterpret_cast<long long *>(sred)[wave * NS + 2] = w1;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __hip_atomic_store(&s_mflag[wave], (int)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else if (lane == 0) st_agent(&rs->error, 1);
      if (wave == 0 && done) {
        bool all = false;
        for (int round = 0; round < a.res_poll_cap; ++round) {
          const int f = lane < RT_WAVES ? __hip_atomic_load(&s_mflag[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : (int)gen;
          if (__builtin_amdgcn_ballot_w64(f != (int)gen) == 0ull) { all = true; break; }
          if ((round & 63) == 63 && ld_agent((const unsigned *)&rs->error) != 0u) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (all) {
          if (a.dbg_times && it == kStampIt && tid == 0) a.dbg_times[5] = __builtin_amdgcn_s_memrealtime();   // master: everybody has arrived
          // iteration `it` is complete everywhere: norm (fixed order), stop rule (src/main.cpp:993-1000), region means of u(it + 1) from
          // the integer totals
          const double *vred = sred;                               // (plain LDS reads: the acquire fence above orders them)
          const long long *vredq = reinterpret_cast<const long long *>(sred);
          double s4 = vred[0];
          long long q0 = vredq[1], q1 = vredq[2];
#pragma unroll
          for (int wv = 1; wv < RT_WAVES; ++wv) { s4 += vred[wv * NS]; q0 += vredq[wv * NS + 1]; q1 += vredq[wv * NS + 2]; }   // fixed order
          const double nrm = sqrt(s4);
          const int stop_now = nrm <= a.stop_cond;          // :1000, after the update
          // chain_means' formula (chain_device.h) on the totals
          const double sh = __builtin_fma((double)q0, a.chain_inv[0], 0.5 * a.npix);
          const double sih = __builtin_fma((double)q1, a.chain_inv[1], 0.5 * a.sum_img[0]);
          const double n1 = sih / sh, n2 = (a.sum_img[0] - sih) / (a.npix - sh);
          const unsigned leave = (stop_now || it + 1 >= nit) ? 1u : 0u;
          stamp(it, kStampIt, 6);                                  // master: norm and means known
          for (int i = lane; i < ntiles; i += 64) {
            st_line16(rs->go, (unsigned)i * 64u, gen, leave, n1);
            st_line16(rs->go, (unsigned)i * 64u + 16u, gen, leave, n2);
          }
          stamp(it, kStampIt, 7);                                  // master: release issued
          // everything else the master books comes AFTER the release (off the critical path of the other workgroups).
          // The last iteration of the launch leaves the sums where the per-launch path expects them: set p0 + executed filled (one
          // shard per sum), the set behind it clear
          if (leave) {
            __hip_atomic_store(&a.chain->v[(phase + 1) & 3][lane], lane == 0 ? q0 : lane == 32 ? q1 : 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.chain->v[(phase + 2) & 3][lane], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (tid == 0) {
            CvhState *st = a.st;
            const int t = t_first + it;
            if (a.trace && t < a.trace_cap) { a.trace[(size_t)t * 3] = c1; a.trace[(size_t)t * 3 + 1] = c2; a.trace[(size_t)t * 3 + 2] = nrm; }
            st->norm = nrm;
            st->steps_done = t + 1;
            st->pending = 0;
            if (stop_now) st->stopped = 1;
            if (a.host_status) {
              __hip_atomic_store(&a.host_status[1], stop_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              __hip_atomic_store(&a.host_status[0], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
          }
        } else if (lane == 0) st_agent(&rs->error, 1);
      }
      // (a master that gave up has raised the error word and released nobody: every wait below and in the other workgroups sees the
      // word and leaves)
    }

    // ---- while the barrier completes: the neighbours' borders of u(it + 1) (they exist as soon as the up-to-four neighbours have
    // arrived), into the halo ring; at the image's border: BORDER_REPLICATE from the tile's own edge (src/main.cpp:351-354)
    if (it + 1 < nit) {
      if (tid < 64) {
        const int nb = lane == 0 ? (ty > 0 ? bid - tc : -1) : lane == 1 ? (ty < tr - 1 ? bid + tc : -1) : lane == 2 ? (tx > 0 ? bid - 1 : -1)
                       : lane == 3 ? (tx < tc - 1 ? bid + 1 : -1) : -1;
        bool sat = nb < 0;
        int ok = 0;
        for (int i = 0; i < a.res_poll_cap; ++i) {
          if (!sat) { const u32x4r_t f = ld_line16(rs->hflag, (unsigned)(nb < 0 ? 0 : nb) * 64u); sat = f.x >= gen && f.x != 0xffffffffu; }
          if (__builtin_amdgcn_ballot_w64(!sat) == 0ull) { ok = 1; break; }
          if ((i & 15) == 15 && ld_agent((const unsigned *)&rs->error) != 0u) break;
          __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) { if (!ok) st_agent(&rs->error, 1); s_flag[0] = ok; }
      }
      __syncthreads();
      const int okn = s_flag[0];
      __syncthreads();
      if (!okn) { gave_up = true; break; }
      const double *const hbn = a.res_halo + (size_t)(it & 1) * ntiles * RT_HALO;
      for (int q = tid; q < 6 * RT_W; q += RT_THREADS) {
        const int piece = q / RT_W, k = q % RT_W;
        double v;
        double *dst;
        if (piece < 2) {            // top halo rows -2, -1 <- the tile above's bottom two rows
          dst = S(piece - 2, k);
          v = ty > 0 ? ld_agent_f64(hbn + (size_t)(bid - tc) * RT_HALO + piece * RT_W + k) : *S(0, k);
        } else if (piece == 2) {    // bottom halo row TH <- the tile below's top row
          dst = S(TH, k);
          v = ty < tr - 1 ? ld_agent_f64(hbn + (size_t)(bid + tc) * RT_HALO + 2 * RT_W + k) : *S(TH - 1, k);
        } else if (piece < 5) {     // left halo columns -2, -1 <- the left tile's right two columns
          dst = k < TH ? S(k, piece - 5) : nullptr;
          v = tx > 0 ? ld_agent_f64(hbn + (size_t)(bid - 1) * RT_HALO + piece * RT_W + k) : *S(k < TH ? k : 0, 0);
        } else {                    // right halo column TWv <- the right tile's left column
          dst = k < TH ? S(k, TWv) : nullptr;
          v = tx < tc - 1 ? ld_agent_f64(hbn + (size_t)(bid + 1) * RT_HALO + 5 * RT_W + k) : *S(k < TH ? k : 0, TWv - 1);
        }
        if (dst) *dst = v;
      }
      // (the barrier at the top of the next iteration orders these LDS writes before their readers)
    }
  }
  if (gave_up) return;
  // ---- leave: every workgroup waits for the release behind the last iteration it computed (the whole grid has then finished it),
  // then writes its tile back into the ping-pong buffer the per-launch path would hold the result in
  if (executed > 0) {
    double d1, d2;
    if (wg_wait_go(rs, bid, executed, a, s_bc, d1, d2) < 0) return;
  }
  // (an even count lands in the buffer the launch read from: every workgroup has long finished reading it -- the first grid
  // barrier lies behind all the tile loads)
  double *const dst = (executed & 1) ? a.u_out : const_cast<double *>(a.u_in);
  if (executed > 0) {
    for (int q = tid; q < TH * (RT_W / 2); q += RT_THREADS) {
      const int r = q / (RT_W / 2), c = 2 * (q % (RT_W / 2));
      if (c < TWv) *reinterpret_cast<double2_t *>(dst + (size_t)(r0 + r) * w + c0 + c) = *reinterpret_cast<const double2_t *>(S(r, c));
    }
  }
}

}  // namespace

size_t cvh_resident_lds_bytes() { return ResSmem::bytes; }
int cvh_resident_tile_w() { return RT_W; }
int cvh_resident_tile_hmax() { return RT_HMAX; }
int cvh_resident_halo_doubles() { return RT_HALO; }

namespace {
typedef void (*ResKernel)(const CvhStepArgs);
ResKernel res_kernel(int band_rows)
{
  switch (band_rows) {
    case 0: return csv_resident_kernel<0>;
    case 2: return csv_resident_kernel<2>;
    case 4: return csv_resident_kernel<4>;
    case 8: return csv_resident_kernel<8>;
    case 16: return csv_resident_kernel<16>;
  }
  return nullptr;
}
}  // namespace

// Workgroups of the resident kernel one CU holds (0: not launchable, e.g. the LDS request was refused): the least over the flavours.
int cvh_resident_blocks_per_cu()
{
  static int cached = -1;
  if (cached >= 0) return cached;
  int least = 1 << 30;
  for (int nr = 0; nr <= 16; nr = nr ? 2 * nr : 2) {
    const void *k = reinterpret_cast<const void *>(res_kernel(nr));
    int n = 0;
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ResSmem::bytes) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, RT_THREADS, ResSmem::bytes) != hipSuccess) {
      (void)hipGetLastError();
      return cached = 0;
    }
    if (n < least) least = n;
  }
  return cached = least;
}

hipError_t cvh_launch_resident(const CvhStepArgs &a, hipStream_t s)
{
  const ResKernel kern = res_kernel(a.res_band_rows);
  if (!kern) return hipErrorInvalidValue;
  if (a.note) {
    cvh_fill_note(a.note, (unsigned)(a.tiles_x * a.tiles_y), RT_THREADS, ResSmem::bytes, "csv_resident_kernel<%d>", a.res_band_rows);
    return hipSuccess;
  }
  CvhStepArgs copy = a;
  void *params[] = {&copy};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void *>(kern), dim3(a.tiles_x * a.tiles_y), dim3(RT_THREADS), params, (unsigned)ResSmem::bytes, s);
}
