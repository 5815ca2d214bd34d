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
//     double-buffered global halo buffer, written and read with sc1 (agent scope), two doubles per 16-byte store / load;
//   * ONE grid barrier with a master.  A tile ARRIVES with three 16-byte agent-scope stores into its own 64-byte line {generation, sum
//     u_diff^2} {generation, sum H'} {generation, sum I H'} -- the last two as chain mode's fixed-point integers (chain_device.h) -- right
//     behind its reduction, before its border stores (those get their own signal line: only the neighbours wait for them).  Workgroup
//     0 is the master: four of its waves watch 64 arrival lines each without workgroup barriers (the other four do the tile's own border
//     work meanwhile), wave 0 adds the integers (exact, order-free: the totals the per-launch path's atomic adds produce), books the
//     iteration (norm, stop rule src/main.cpp:1000, trace row) and RELEASES everybody with one line per XCD that carries the leave bit and
//     the region means of the new level set.  The stop rule therefore fires at the reference's iteration with no extra iteration computed,
//     and no atomic is issued inside the launch.  (What the barrier costs and how it got here: DESIGN.md 4.1b.)
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
static_assert(64 * (RT_WAVES / 2) >= CVH_RESIDENT_MAX_TILES, "the master's four polling waves watch 64 arrival lines each");
static_assert(RT_WAVES * ResSmem::NS >= 3 * RT_WAVES + 3 * (RT_WAVES / 2), "sred holds the tiles' 8 x 3 partial sums and the master's 4 x 3");
static_assert(ResSmem::bytes <= 160 * 1024, "the tile, its halo ring and the tables must fit one CU's LDS");

__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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

// (lds_barrier(), buffer_ops.h: the workgroup barrier that orders LDS only -- the master's wave 0 has its release and its bookkeeping
// stores under way on the path from one iteration into the next)

// Release lines are shared: workgroups are dealt to the 8 XCDs round-robin, and 2^shift tiles of one XCD read the same line (shift 0: a
// line per tile).  The master's release is then 2 * 256 / 2^shift sixteen-byte stores instead of 512 -- one wave needs 1.2 us to drain 512
// of them (the last tile saw its release 1.3 us after the first store was issued, profiles/r04_C4/resident_timeline_2048_master1.txt).
__device__ __forceinline__ unsigned go_line(int tile, int shift) { return shift >= 6 ? 0u : (unsigned)((tile & 7) + 8 * ((tile >> 3) >> shift)); }   // (6: one line for all)
__device__ __forceinline__ int go_lines(int ntiles, int shift) { return shift >= 6 ? 1 : 8 * ((((ntiles - 1) >> 3) >> shift) + 1); }

// Thread 0 polls this workgroup's release line until both halves carry generation >= `gen` (bounded); the workgroup meets at a
// barrier.  Returns the leave bit (or -1: gave up) and the region means the line carries.
// (`own`: the master workgroup wrote this very release itself -- thread 0 passes what it wrote, own.gen = its generation, and no line is polled)
struct OwnRelease { int gen, leave; double c1, c2; };
__device__ __forceinline__ int wg_wait_go(const CvhResident *rs, int bid, int gen, const CvhStepArgs &a, double *s_bc /*[4]*/, double &c1, double &c2,
                                          const OwnRelease &own)
{
  if (threadIdx.x == 0) {
    int res = -1;
    double m1 = 0.0, m2 = 0.0;
    if (own.gen == gen) { res = own.leave; m1 = own.c1; m2 = own.c2; }
    else
    // (one poll in flight: two or four in flight sample the line more often but cost 0.5 / 0.7 us per iteration at 2048^2 -- the
    // polls of 256 workgroups compete with the arrivals and the release for the same fabric)
    for (int i = 0; i < a.res_poll_cap; ++i) {
      const unsigned line = go_line(bid, a.res_go_shift) * 64u;
      const u32x4r_t ga = ld_line16(rs->go, line), gb = ld_line16(rs->go, line + 16u);
      if (ga.x == gb.x && ga.x >= (unsigned)gen && ga.x != 0xffffffffu) { res = (int)(ga.y & 1u); m1 = line16_f64(ga); m2 = line16_f64(gb); break; }
      if ((i & 15) == 15 && ld_agent((const unsigned *)&rs->error) != 0u) break;
      if (i >= 64) __builtin_amdgcn_s_sleep(16); else if (i >= 2) __builtin_amdgcn_s_sleep(kReleaseSleep);
    }
    if (res < 0) st_agent(const_cast<int *>(&rs->error), 1);
    s_bc[0] = (double)res; s_bc[1] = m1; s_bc[2] = m2;
  }
  lds_barrier();
  const int res = (int)s_bc[0];
  c1 = s_bc[1]; c2 = s_bc[2];
  lds_barrier();
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
  if (tid == 0) s_flag[9] = 0;             // master workgroup: waves whose border stores are acknowledged, counted over the launch
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
  int it = 0;
  // The master's books of one iteration (thread 0 of workgroup 0): trace row, state block and -- when the launch ends or the stop rule
  // fired -- the two words the host polls in pinned memory.  (Inside a launch the host does not need them: it runs at most four launches
  // ahead of the count the LAST word of a launch reports; a store to host memory is acknowledged after 1.5 us, and a wave cannot wait for
  // anything else of its own without waiting for that.)
  auto book = [&](double m1, double m2, double nrm, int stop_now, bool last) {
    if (tid != 0) return;
    CvhState *st = a.st;
    const int t = t_first + it;
    if (a.trace && t < a.trace_cap) { a.trace[(size_t)t * 3] = m1; a.trace[(size_t)t * 3 + 1] = m2; a.trace[(size_t)t * 3 + 2] = nrm; }
    st->norm = nrm;
    st->steps_done = t + 1;
    st->pending = 0;
    if (stop_now) st->stopped = 1;
    if (a.host_status && last) {
      __hip_atomic_store(&a.host_status[1], stop_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&a.host_status[0], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };
  // (master workgroup: the release it wrote itself, handed from thread 0 to the workgroup without a poll)
  OwnRelease own = {-1, 0, 0.0, 0.0};
  bool book_pending = false;   // (wave 0 of the master) iteration `it` is released but not yet booked
  double book_nrm = 0.0;
  bool have_go = false;      // (workgroup-uniform) the release into the next iteration was handed over inside the workgroup
  int go_known = -1;
  if (bid == 0 && tid == 0) { rs->pad[0] = (unsigned)t_first; rs->pad[1] = (unsigned)nit; rs->pad[2] = (unsigned)a.st->steps_done; }   // (diagnostic record of the launch)
  // ---- what an iteration reads before any wave writes and that does not wait for the release (c1 / c2): every wave takes the rows around
  // its band into registers, one thread per row computes the normalised x-gradient of tile column -1 (nobody owns that column; lane 0 needs
  // it for column 0), and ny of the row above the band.  Called when the tile and its halo ring are complete: before the loop, and at the end
  // of an iteration behind the fetch of the neighbours' borders -- off the path from the release into the march.
  double2_t p_um = {0.0, 0.0}, p_u0 = {0.0, 0.0}, p_ubot = {0.0, 0.0};
  double p_uw = 0.0, p_ue = 0.0, p_nypa = 0.0, p_nypb = 0.0;
  auto pre_reads = [&]() {
    if (tid >= 256 && tid - 256 < TH) snxl[tid - 256] = norm(*S(tid - 256, 0), *S(tid - 256, -2), *S(tid - 256, -1));
    const double2_t um2_0 = *reinterpret_cast<const double2_t *>(S(rb0 - 2, ca));
    p_um = *reinterpret_cast<const double2_t *>(S(rb0 - 1, ca));
    p_u0 = *reinterpret_cast<const double2_t *>(S(rb0, ca));
    p_ubot = *reinterpret_cast<const double2_t *>(S(rb1, ca));
    p_uw = *S(rb0, ca - 1); p_ue = *S(rb0, ca + 2);
    const double2_t u1st = *reinterpret_cast<const double2_t *>(S(rb0 + 1 < rb1 ? rb0 + 1 : rb0, ca));   // row rb0 + 1 (own band, if it has one)
    if (rb1 > rb0) {
      p_nypa = norm(p_u0.x, um2_0.x, p_um.x); p_nypb = norm(p_u0.y, um2_0.y, p_um.y);   // ny at row rb0 - 1
      if (r0 + rb0 == 0) {   // kappa_y(0, .) = 0 (:372): ny_prev := row 0's own ny, the very expression the row uses
        const double2_t up0 = (rb0 + 1 < rb1) ? u1st : p_ubot;
        p_nypa = norm(up0.x, p_um.x, p_u0.x); p_nypb = norm(up0.y, p_um.y, p_u0.y);
      }
    }
  };
  for (it = 0; it < nit; ++it) {
    const int phase = (a.chain_phase + it) & 3;
    // ---- the release behind iteration it - 1: leave bit and the region means of u(it); the halos were fetched while waiting
    if (it > 0) {
      const int go = have_go ? go_known : wg_wait_go(rs, bid, it, a, s_bc, c1, c2, own);   // (have_go: the master workgroup, short way)
      have_go = false;
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
    // (what an iteration needs that does NOT depend on the region means -- column -1's normalised x-gradient, the rows around the band in
    // registers, ny of the row above the band -- was taken BEFORE the release was waited for: pre_reads, at the end of the previous iteration)
    if (it == 0) pre_reads();
    double2_t um = p_um, u0 = p_u0;
    const double2_t ubot = p_ubot;
    double uw = p_uw, ue = p_ue;
    lds_barrier();                                             // (LDS only: the master's bookkeeping stores may still be in flight)
    stamp(it, kStampIt, 1);                                    // table in LDS, band borders in registers

    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0;
    if (rb1 > rb0) {
      double nypa = p_nypa, nypb = p_nypb;                      // ny at row rb0 - 1 (pre_reads)
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
        if (NRT >= 16 && NEARFORM) {
          // (the near copy of a 16-row band is a LOOP over its four groups: unrolled, the table forms of sixteen rows in flight took the
          // kernel to 256 VGPRs and 141 spilled registers, and values that live across the march were reloaded from scratch on the
          // far path as well)
#pragma unroll 1
          for (int g = 0; g < NRT / 4; ++g) {
            quarter_prio(4 * g, NRT);
#pragma unroll
            for (int k = 0; k < 4; ++k) row(4 * g + k, k, near_tag);
#pragma unroll
            for (int k = 0; k < 4; ++k) correct(k, near_tag);
          }
        } else if (NRT >= 4) {
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
    if (a.dbg_times && it == kStampIt && lane == 0 && (bid == 0 || bid == 100 || bid == 255))   // diagnostic: every wave of three tiles
      a.dbg_times[(size_t)CVH_RESIDENT_MAX_TILES * 12 + 16 + (bid == 0 ? 0 : bid == 100 ? 8 : 16) + wave] = __builtin_amdgcn_s_memrealtime();
    // ---- the tile's sums and its arrival.  One channel, H' sums: of the NS sums only [0] sum H', [2] sum I H' and [4] sum u_diff^2 are
    // carried.  Every wave reduces its three (DPP), lane 0 leaves them in LDS, ONE barrier (it also orders the tile writes before the border
    // reads below), and the three threads that store the arrival add the eight partials in a fixed order -- nobody else needs the totals.
    // (sred[0 .. 24): these partials; sred[24 .. 36): the master's, below)
    {
      const double v0 = wave_sum(acc[0]), v2 = wave_sum(acc[2]), v4 = wave_sum(acc[4]);
      if (lane == 0) { sred[wave * 3] = v4; sred[wave * 3 + 1] = v0; sred[wave * 3 + 2] = v2; }
    }
    __syncthreads();
    stamp(it, kStampIt, 3);                                    // all waves done, sums in LDS
    executed = it + 1;
    const unsigned gen = (unsigned)(it + 1);
    // three 16-byte lines {generation, payload}: sum u_diff^2, and the fixed-point sums the next iteration's means come from (distinct
    // addresses: 256 arrivals on one counter serialise for 6 us).  The arrival does not wait for the border stores below: the master needs
    // the sums only, the neighbours get their own signal.  (Two pieces -- the 64-bit sum of H' split over the spare words -- were tried: a third
    // fewer arrival stores and polls, and no faster: 13.03 vs 12.92 us at 2048^2, 5.73 vs 5.39 at 512^2, gpurun_out/r4s50.)
    if (tid < 3) {
      double t = sred[tid];
#pragma unroll
      for (int wv = 1; wv < RT_WAVES; ++wv) t += sred[wv * 3 + tid];   // fixed order
      const unsigned long long payload = tid == 0 ? (unsigned long long)__double_as_longlong(t)
                                       : (unsigned long long)__double2ll_rn(t * a.chain_scale[tid - 1]);
      st_line16_u64(rs->flag, ((unsigned)tid * CVH_RESIDENT_MAX_TILES + (unsigned)bid) * 16u, gen, 0u, payload);   // piece-major: the master's polls read neighbouring entries
    }
    // ---- what crosses to the neighbours: the tile's border (6 x 128 doubles, agent-scope stores) and, once those are acknowledged, the
    // border signal; then the neighbours' borders of u(it + 1) -- they exist as soon as the up-to-four neighbours have stored THEIR signal --
    // go into the halo ring while the barrier completes; at the image's border: BORDER_REPLICATE from the tile's own edge (src/main.cpp:351-354)
    double *const hb = halo_mine[it & 1];
    auto border_value = [&](int q) -> double {
      const int piece = q / RT_W, k = q % RT_W;
      if (piece < 2) return *S(TH - 2 + piece, k);                    // bottom two rows   (TH >= 2: host)
      if (piece == 2) return *S(0, k);                                // top row
      if (piece < 5) return *S(k < TH ? k : 0, TWv - 5 + piece);      // right two columns: TWv - 2, TWv - 1
      return *S(k < TH ? k : 0, 0);                                   // left column
    };
    // two neighbouring elements of a piece with ONE 16-byte store (RT_W is even: a pair never straddles two pieces): 384 stores per tile
    auto store_border = [&](int p) {
      const double v0 = border_value(2 * p), v1 = border_value(2 * p + 1);
      const unsigned long long b0 = (unsigned long long)__double_as_longlong(v0), b1 = (unsigned long long)__double_as_longlong(v1);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4r_t{(unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32)}, make_rsrc(hb, 0x7fffffffu),
                                             (unsigned)p * 16u, 0u, 16 /* sc1 */);
    };
    constexpr int kPairs = 3 * RT_W;
    const bool want_borders = it + 1 < nit;
    const int nb_lane = lane == 0 ? (ty > 0 ? bid - tc : -1) : lane == 1 ? (ty < tr - 1 ? bid + tc : -1) : lane == 2 ? (tx > 0 ? bid - 1 : -1)
                        : lane == 3 ? (tx < tc - 1 ? bid + 1 : -1) : -1;          // lanes 0-3 of a polling wave watch one neighbour each
    auto tagged = [&](const u32x4r_t &f) -> bool { return f.x >= gen && f.x != 0xffffffffu; };
    // (a wave-wide bounded wait for the lanes' neighbours)
    auto neighbours_arrived = [&]() -> int {
      bool sat = nb_lane < 0;
      for (int i = 0; i < a.res_poll_cap; ++i) {
        if (!sat) sat = tagged(ld_line16(rs->hflag, (unsigned)(nb_lane < 0 ? 0 : nb_lane) * 64u));
        if (__builtin_amdgcn_ballot_w64(!sat) == 0ull) return 1;
        if ((i & 15) == 15 && ld_agent((const unsigned *)&rs->error) != 0u) break;
        __builtin_amdgcn_s_sleep(2);
      }
      return 0;
    };
    const double *const hbn = a.res_halo + (size_t)(it & 1) * ntiles * RT_HALO;
    // element q of the 6 x 128 border: its value (a neighbour's store, or the tile's own edge) into the halo cell it belongs to
    // pair p of the 6 x 128 border (elements 2p, 2p + 1 of one piece): the values (a neighbour's 16-byte store, or the tile's own edge) into
    // the two halo cells they belong to
    auto fetch = [&](int p) {
      const int q = 2 * p, piece = q / RT_W, k = q % RT_W;
      double *d0, *d1;
      int nb;              // the neighbour the piece comes from (-1: the image ends here)
      double e0, e1;       // ... and then: the tile's own edge
      if (piece < 2) {            // top halo rows -2, -1 <- the tile above's bottom two rows
        d0 = S(piece - 2, k); d1 = S(piece - 2, k + 1);
        nb = ty > 0 ? bid - tc : -1; e0 = *S(0, k); e1 = *S(0, k + 1);
      } else if (piece == 2) {    // bottom halo row TH <- the tile below's top row
        d0 = S(TH, k); d1 = S(TH, k + 1);
        nb = ty < tr - 1 ? bid + tc : -1; e0 = *S(TH - 1, k); e1 = *S(TH - 1, k + 1);
      } else if (piece < 5) {     // left halo columns -2, -1 <- the left tile's right two columns
        d0 = k < TH ? S(k, piece - 5) : nullptr; d1 = k + 1 < TH ? S(k + 1, piece - 5) : nullptr;
        nb = tx > 0 ? bid - 1 : -1; e0 = *S(k < TH ? k : 0, 0); e1 = *S(k + 1 < TH ? k + 1 : 0, 0);
      } else {                    // right halo column TWv <- the right tile's left column
        d0 = k < TH ? S(k, TWv) : nullptr; d1 = k + 1 < TH ? S(k + 1, TWv) : nullptr;
        nb = tx < tc - 1 ? bid + 1 : -1; e0 = *S(k < TH ? k : 0, TWv - 1); e1 = *S(k + 1 < TH ? k + 1 : 0, TWv - 1);
      }
      if (nb >= 0) {
        const u32x4r_t v = ld_line16(hbn + (size_t)nb * RT_HALO, (unsigned)p * 16u);
        e0 = __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
        e1 = line16_f64(v);
      }
      if (d0) *d0 = e0;
      if (d1) *d1 = e1;
    };

    if (bid != 0) {
      // ---- an ordinary tile: border, border signal, the neighbours' borders; the release is polled at the top of the next iteration
      for (int p = tid; p < kPairs; p += RT_THREADS) store_border(p);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      stamp(it, kStampIt, 4);                                    // borders have reached memory
      if (tid == 0) st_line16_u64(rs->hflag, (unsigned)bid * 64u, gen, 0u, 0ull);
      if (want_borders) {
        if (tid < 64) {
          const int ok = neighbours_arrived();
          if (lane == 0) { if (!ok) st_agent(&rs->error, 1); s_flag[0] = ok; }
        }
        __syncthreads();
        const int okn = s_flag[0];
        __syncthreads();
        if (!okn) { gave_up = true; break; }
        for (int p = tid; p < kPairs; p += RT_THREADS) fetch(p);
        lds_barrier();       // (the halo ring is complete)
        pre_reads();
      }
    } else {
      // ---- workgroup 0 is the barrier's MASTER, and a tile like any other -- among the last to arrive as often as any other, so nothing of
      // its own may stand between its arrival and its first look at the arrival lines.  Its waves split the work (no workgroup barrier
      // until the release is out):
      //   waves 0-3 POLL: 64 arrival lines each, lane = tile, three 16-byte pieces per line; a wave whose share is complete leaves its
      //     partial sums and the generation in LDS; wave 0 collects the four, books the iteration and releases everybody.  These waves
      //     have no store in flight: memory operations of a wave return in order, and polls issued behind the border stores came back
      //     after 1.4 us instead of 0.8 (profiles/r04_C4/resident_timeline_2048_master_waves.txt);
      //   waves 4-7 WORK: the tile's border stores, the border signal (the last of the four whose stores are acknowledged), the wait for
      //     the tile's own neighbours and the fetch of their borders -- finished long before the release is.
      // (Single-wave code is latency-bound -- 8 cycles an instruction: everything wave 0 does is the critical path of 255 waiting
      // workgroups.)  Round-4 history of this block: DESIGN.md 4.1b.
      constexpr int kPollWaves = RT_WAVES / 2, kWorkThreads = RT_THREADS - 64 * kPollWaves;
      static_assert(64 * kPollWaves >= CVH_RESIDENT_MAX_TILES, "the master's polling waves watch 64 arrival lines each");
      int ok_w = 1;                  // (wave-uniform) this wave's errand went well
      if (wave >= kPollWaves) {
        for (int p = tid - 64 * kPollWaves; p < kPairs; p += kWorkThreads) store_border(p);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && __hip_atomic_fetch_add(&s_flag[9], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1 == (RT_WAVES - kPollWaves) * (int)gen) {
          if (a.dbg_times && it == kStampIt) a.dbg_times[(size_t)bid * 12 + 4] = __builtin_amdgcn_s_memrealtime();
          st_line16_u64(rs->hflag, (unsigned)bid * 64u, gen, 0u, 0ull);
        }
        if (want_borders) {
          ok_w = neighbours_arrived();
          if (ok_w) for (int p = tid - 64 * kPollWaves; p < kPairs; p += kWorkThreads) fetch(p);
        }
        if (lane == 0 && !ok_w) st_agent(&rs->error, 1);
      } else {
        const int b = wave * 64 + lane;
        const bool have = b < ntiles;
        bool done = false;
        u32x4r_t fa = {0u, 0u, 0u, 0u}, fb = {0u, 0u, 0u, 0u}, fc = {0u, 0u, 0u, 0u};
        // (one poll in flight: two in flight, half a round trip apart, sample a line twice as often and are SLOWER -- 2048^2 14.60 -> 15.04 us,
        // 1024^2 7.89 -> 8.19: reads of a line that is being written get in the way of the write)
        int rounds = 0;
        bool ok = !have;             // (per lane, sticky: a line that has arrived is not read again -- the last rounds poll the stragglers only)
        for (int round = 0; round < a.res_poll_cap; ++round) {
          if (!ok) {
            fa = ld_line16(rs->flag, (unsigned)b * 16u);
            fb = ld_line16(rs->flag, (CVH_RESIDENT_MAX_TILES + (unsigned)b) * 16u);
            fc = ld_line16(rs->flag, (2u * CVH_RESIDENT_MAX_TILES + (unsigned)b) * 16u);
          }
          ok = !have || (tagged(fa) && tagged(fb) && tagged(fc));
          rounds = round + 1;
          if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) { done = true; break; }
          if ((round & 15) == 15 && ld_agent((const unsigned *)&rs->error) != 0u) break;
          __builtin_amdgcn_s_sleep(kMasterSleep);
        }
        if (a.dbg_times && it == kStampIt && lane == 0) {   // diagnostic: when this wave's share was complete, after how many rounds
          a.dbg_times[(size_t)CVH_RESIDENT_MAX_TILES * 12 + wave] = __builtin_amdgcn_s_memrealtime();
          a.dbg_times[(size_t)CVH_RESIDENT_MAX_TILES * 12 + 8 + wave] = (unsigned long long)rounds;
        }
        if (done) {
          const double ws = wave_sum(have ? line16_f64(fa) : 0.0);                       // fixed order: lane = tile
          const long long r0s = row16_sum_i64(have ? line16_i64(fb) : 0ll), r1s = row16_sum_i64(have ? line16_i64(fc) : 0ll);
          const long long w0 = (read_lane_i64(r0s, 0) + read_lane_i64(r0s, 16)) + (read_lane_i64(r0s, 32) + read_lane_i64(r0s, 48));
          const long long w1 = (read_lane_i64(r1s, 0) + read_lane_i64(r1s, 16)) + (read_lane_i64(r1s, 32) + read_lane_i64(r1s, 48));
          if (lane == 0) {
            sred[24 + wave * 3] = ws;
            reinterpret_cast<long long *>(sred)[24 + wave * 3 + 1] = w0;
            reinterpret_cast<long long *>(sred)[24 + wave * 3 + 2] = w1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(&s_mflag[wave], (int)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        } else { ok_w = 0; if (lane == 0) st_agent(&rs->error, 1); }
        if (wave == 0 && done) {
          bool all = false;
          for (int round = 0; round < a.res_poll_cap; ++round) {
            const int f = lane < kPollWaves ? __hip_atomic_load(&s_mflag[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : (int)gen;
            if (__builtin_amdgcn_ballot_w64(f != (int)gen) == 0ull) { all = true; break; }
            if ((round & 63) == 63 && ld_agent((const unsigned *)&rs->error) != 0u) break;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (all) {
            if (a.dbg_times && it == kStampIt && tid == 0) a.dbg_times[5] = __builtin_amdgcn_s_memrealtime();   // master: everybody has arrived
            // iteration `it` is complete everywhere: norm (fixed order), stop rule (src/main.cpp:993-1000), region means of u(it + 1) from
            // the integer totals
            const double *vred = sred + 24;                          // (plain LDS reads: the acquire fence above orders them)
            const long long *vredq = reinterpret_cast<const long long *>(sred + 24);
            double s4 = vred[0];
            long long q0 = vredq[1], q1 = vredq[2];
#pragma unroll
            for (int wv = 1; wv < kPollWaves; ++wv) { s4 += vred[wv * 3]; q0 += vredq[wv * 3 + 1]; q1 += vredq[wv * 3 + 2]; }   // fixed order
            const double nrm = sqrt(s4);
            const int stop_now = nrm <= a.stop_cond;          // :1000, after the update
            // chain_means' formula (chain_device.h) on the totals
            const double sh = __builtin_fma((double)q0, a.chain_inv[0], 0.5 * a.npix);
            const double sih = __builtin_fma((double)q1, a.chain_inv[1], 0.5 * a.sum_img[0]);
            // (both quotients through one division sequence in lanes 0 / 1 was tried: norm + means 0.28 -> 0.50 us -- the two sequences overlap as they are)
            const double n1 = sih / sh, n2 = (a.sum_img[0] - sih) / (a.npix - sh);
            const unsigned leave = (stop_now || it + 1 >= nit) ? 1u : 0u;
            stamp(it, kStampIt, 6);                                  // master: norm and means known
            for (int i = lane, nl = go_lines(ntiles, a.res_go_shift); i < nl; i += 64) {
              st_line16(rs->go, (unsigned)i * 64u, gen, leave, n1);
              st_line16(rs->go, (unsigned)i * 64u + 16u, gen, leave, n2);
            }
            stamp(it, kStampIt, 7);                                  // master: release issued
            own = OwnRelease{(int)gen, (int)leave, n1, n2};
            // everything else the master books comes AFTER the release (off the critical path of the other workgroups).
            // The last iteration of the launch leaves the sums where the per-launch path expects them: set p0 + executed filled (one
            // shard per sum), the set behind it clear
            if (leave) {
              __hip_atomic_store(&a.chain->v[(phase + 1) & 3][lane], lane == 0 ? q0 : lane == 32 ? q1 : 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(&a.chain->v[(phase + 2) & 3][lane], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (inside a launch the books are written behind the workgroup's meeting below: on gfx9 a wave's stores count in vmcnt like its
            // loads, and every conservative `s_waitcnt vmcnt(0)` on wave 0's way into the next iteration would wait for them)
            if (leave) book(c1, c2, nrm, stop_now, true);
            else { book_pending = true; book_nrm = nrm; }
          } else if (lane == 0) st_agent(&rs->error, 1);
        }
        // (a master that gave up has raised the error word and released nobody: every wait below and in the other workgroups sees the
        // word and leaves)
      }
      // the eight waves meet ONCE, behind the release, and thread 0 hands over what it wrote: the master tile polls no release line and
      // has no round trip left on its way into the next iteration (it used to be the LAST tile into every iteration, 1.3 us behind the median)
      if (lane == 0) s_flag[wave] = ok_w;
      if (tid == 0) { s_bc[0] = (double)(own.gen == (int)gen ? own.leave : -1); s_bc[1] = own.c1; s_bc[2] = own.c2; }
      lds_barrier();
      int okn = 1;
#pragma unroll
      for (int wv = 0; wv < RT_WAVES; ++wv) okn &= s_flag[wv];
      go_known = (int)s_bc[0];
      const double k1 = s_bc[1], k2 = s_bc[2];
      lds_barrier();
      if (!okn || go_known < 0) { gave_up = true; break; }   // (a neighbour or the master's own collection gave up: the error word is up)
      if (book_pending) { book(c1, c2, book_nrm, 0, false); book_pending = false; }   // (c1 / c2: still the means this iteration ran with)
      if (want_borders) { c1 = k1; c2 = k2; pre_reads(); }     // (the workers' fetch lies in front of the meeting's first barrier)
      have_go = true;
    }
  }
  if (gave_up) return;
  // ---- leave: every workgroup waits for the release behind the last iteration it computed (the whole grid has then finished it),
  // then writes its tile back into the ping-pong buffer the per-launch path would hold the result in
  if (executed > 0) {
    double d1, d2;
    if (wg_wait_go(rs, bid, executed, a, s_bc, d1, d2, own) < 0) return;
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
