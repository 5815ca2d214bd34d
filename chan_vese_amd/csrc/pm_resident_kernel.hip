// pm_resident_kernel.hip — Perona-Malik (src/main.cpp:478-560) on a cache-resident plane (gfx950, wave64): the FP64 state of one
// channel lives in LDS for a whole chunk of time steps.
//
// A 2048^2 plane is 32 MiB of doubles; the 256 CUs of an MI355X hold 40 MiB of LDS.  One cooperative launch cuts the plane into
// tiles_y x tiles_x tiles of <= 128 rows x 128 columns, ONE workgroup (8 waves) per tile and per CU, loads every tile with a halo
// ring of width 2 into its CU's LDS once, runs `res_steps` time steps on it and writes the tiles back at the end.  A launch per
// step (or per two steps, pm_wave_k2_kernel.hip) moves the plane through the memory system every step; here a step moves the
// tiles' borders only, and -- unlike the CSV iteration (csv_resident_kernel.hip) -- a Perona-Malik step has no global sum, so
// there is NO grid barrier: a tile waits for its up-to-eight neighbours only.
//
// One step of a tile:
//   1. (steps > 0) every thread polls its <= 3 cells of the halo ring in the neighbours' border pieces until they carry the previous
//      step: an entry of the border buffer is 16 bytes {value, tag = (launch serial, step)}, written with ONE store, so there is no
//      separate signal to wait for (one memory round trip instead of signal-then-data); at the image's border the cell is the
//      clamped pixel (BORDER_REPLICATE), own tile or a neighbour's;
//   2. every wave computes its band of NR rows x 128 columns (lane <-> two adjacent columns) from the OLD tile: the row pass of the
//      Sobel pair (:503-504) is shared by the three rows of g that need it, g = 1 / (1 + |Sobel|^2 / K^2) (:503-520) of the own
//      columns marches down in registers, its x-neighbours come over DPP, the two edge columns' g (tile columns -1 and TW) from a
//      per-wave pre-pass; a row is rewritten in place as soon as it is computed, except the band's first and last two rows (the
//      neighbouring bands still read them), which wait in registers for
//   3. the workgroup barrier;
//   4. the tile's border -- top / bottom two rows, left / right two columns: a step reaches two pixels (g of a neighbour needs
//      the neighbour's 3 x 3) -- goes into a double-buffered global buffer with agent-scope 16-byte stores: from the registers while
//      the band is computed (full tiles), from LDS behind the barrier (ragged tiles).
// Every wait is a bounded poll: a thread that gives up raises CvhResident::error and its workgroup leaves, and so does everybody
// waiting for it -- the grid always drains; the host reports the error.
//
// The arithmetic of a pixel is pm_wave_k2_kernel.hip's (hence pm_wave_kernel's) operation by operation in both flavours: STRICT
// stays bit-exact against the oracle, FAST gives the very doubles the per-launch kernels give.
#include "csv_device.h"
#include "buffer_ops.h"
#include "wave_math.h"

using namespace cvh_dev;

namespace {

constexpr int PT_W = 128;                 // tile width: 64 lanes x 2 pixels
constexpr int PT_HMAX = 128;              // most rows a tile may have (LDS)
constexpr int PT_PITCH = PT_W + 4;        // doubles per LDS row: tile columns -2 .. 129
constexpr int PT_WAVES = 8, PT_THREADS = 64 * PT_WAVES;
constexpr int PT_HALO = 8 * PT_W;         // doubles a tile publishes per step: rows 0, 1, TH-2, TH-1, columns 0, 1, TW-2, TW-1
constexpr int PT_RING = 4 * PT_PITCH + 4 * PT_HMAX;          // cells of the halo ring (rows -2, -1, TH, TH+1; columns -2, -1, TW, TW+1)
constexpr int PT_GATHER = (PT_RING + PT_THREADS - 1) / PT_THREADS;   // ring cells per thread

constexpr int kRowsPerSched = 2;          // a scheduling barrier behind every second row of the march (round 3, pm_sched_variants.txt: 1, 4 or no barrier: no faster, more registers)

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4s_t __attribute__((ext_vector_type(4)));

struct PmResSmem {
  static constexpr int off_I = 0;                                          // (PT_HMAX + 4) rows x PT_PITCH: tile rows -2 .. TH+1
  static constexpr int off_edge = off_I + (PT_HMAX + 4) * PT_PITCH;       // per wave: g of tile column -1 [16], of tile column TW [16]
  static constexpr int off_flag = off_edge + PT_WAVES * 32;
  static constexpr int off_stage = off_flag + 2;                          // per wave: the four edge columns of up to four rows [4][4], on their way to the border buffer
  static constexpr int off_rstage = off_stage + PT_WAVES * 16;             // per wave: one row of the tile [PT_W] on its way to the border buffer (waves 0 and 7 of a full tile)
  static constexpr int doubles = off_rstage + PT_WAVES * PT_W;
  static constexpr size_t bytes = (size_t)doubles * sizeof(double);
};
static_assert(PmResSmem::bytes <= 160 * 1024, "the tile with its halo ring must fit one CU's LDS");

__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte agent-scope (sc1) accesses to the border buffer: one entry {value, tag}, one transaction
__device__ __forceinline__ u32x4s_t ld_line16(const void *base, unsigned byte_off)
{
  return __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(base, 0x7fffffffu), byte_off, 0u, 16 /* sc1 */);
}
__device__ __forceinline__ u32x4s_t tagged(double v, unsigned tag_lo, unsigned tag_hi)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return u32x4s_t{(unsigned)b, (unsigned)(b >> 32), tag_lo, tag_hi};
}

// lane l gets lane l + 1's v (wave_shl:1); lane 63 has no source lane and keeps `old` (wave_math.h has the other direction)
__device__ __forceinline__ double dpp_from_right_or(double old, double v)
{
  const long long vb = __double_as_longlong(v), ob = __double_as_longlong(old);
  const int lo = __builtin_amdgcn_update_dpp((int)ob, (int)vb, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(ob >> 32), (int)(vb >> 32), 0x130, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <bool FAST, int NR>
__global__ __launch_bounds__(PT_THREADS, 1) void pm_resident_kernel(const CvhPmArgs a)
{
  using L = PmResSmem;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *sI = smem + L::off_I;
  double *sEdge = smem + L::off_edge + 32 * (threadIdx.x >> 6);
  int *s_flag = (int *)(smem + L::off_flag);
  double *sStage = smem + L::off_stage + 16 * (threadIdx.x >> 6);
  double *sRow = smem + L::off_rstage + PT_W * (threadIdx.x >> 6);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = a.h, w = a.w;
  CvhResident *const rs = a.resident;
  if (tid == 0) s_flag[0] = 0;                           // raised by a thread whose wait gave up (read behind the next barrier)

  // ---- this workgroup's tile
  const int tr = a.tiles_y, tc = a.tiles_x, ntiles = tr * tc;
  const int bid = (int)blockIdx.x;
  const int ty = bid / tc, tx = bid % tc;
  constexpr int THN = PT_WAVES * NR;                   // rows of a tile: every wave a band of exactly NR rows (straight-line march)
  auto row0 = [&](int t) -> int { return t * THN < h ? t * THN : h; };
  const int r0 = row0(ty), r1 = row0(ty + 1);
  const int TH = r1 - r0;                              // THN; the last tile row of the image may be shorter (>= 2: host)
  const int c0 = tx * PT_W;
  const int TWv = (w - c0) < PT_W ? (w - c0) : PT_W;   // even, >= 2 (host: w even)
  // LDS index of tile element (row r in -2 .. TH+1, column c in -2 .. 129)
  auto SI = [&](int r, int c) -> int { return (r + 2) * PT_PITCH + (c + 2); };
  auto S = [&](int r, int c) -> double * { return sI + SI(r, c); };

  // ---- once per launch: the tile with its halo ring straight from the plane (clamped = BORDER_REPLICATE, the index clamp of :527-530)
  for (int q = tid; q < (TH + 4) * PT_PITCH; q += PT_THREADS) {
    const int r = q / PT_PITCH - 2, c = q % PT_PITCH - 2;
    const int gr = clampi(r0 + r, 0, h - 1), gc = clampi(c0 + c, 0, w - 1);
    sI[q] = a.in[(size_t)gr * w + gc];
  }
  // ---- once per launch: where this thread's cells of the halo ring come from in later steps -- a cell of the own tile (the image's
  // border: the clamped pixel) or an entry of a neighbour's border pieces
  int g_dst[PT_GATHER], g_src[PT_GATHER];     // LDS index (-1: none); source: >= 0 entry of the border buffer, < 0: ~(LDS index)
#pragma unroll
  for (int j = 0; j < PT_GATHER; ++j) {
    const int e = tid + j * PT_THREADS;
    int r = 0, c = 0;
    bool valid = false;
    if (e < 4 * PT_PITCH) {
      const int hr = e / PT_PITCH;
      c = e % PT_PITCH - 2;
      r = hr < 2 ? hr - 2 : TH + (hr - 2);
      valid = c <= TWv + 1;
    } else if (e < PT_RING) {
      const int e2 = e - 4 * PT_PITCH, hc = e2 / PT_HMAX;
      r = e2 % PT_HMAX;
      c = hc < 2 ? hc - 2 : TWv + (hc - 2);
      valid = r < TH;
    }
    g_dst[j] = -1; g_src[j] = 0;
    if (valid) {
      const int gr = clampi(r0 + r, 0, h - 1), gc = clampi(c0 + c, 0, w - 1);
      const int sy = gr < r0 ? ty - 1 : (gr >= r1 ? ty + 1 : ty), sx = gc / PT_W;
      g_dst[j] = SI(r, c);
      if (sy == ty && sx == tx) {
        g_src[j] = ~SI(gr - r0, gc - c0);
      } else {
        const int q0 = row0(sy), q1 = row0(sy + 1), th_n = q1 - q0;
        const int d0 = sx * PT_W, tw_n = (w - d0) < PT_W ? (w - d0) : PT_W;
        const int lr = gr - q0, lc = gc - d0;
        int piece, idx;
        if (sy != ty) { piece = sy < ty ? (lr == th_n - 2 ? 2 : 3) : (lr == 0 ? 0 : 1); idx = lc; }
        else { piece = sx < tx ? (lc == tw_n - 2 ? 6 : 7) : (lc == 0 ? 4 : 5); idx = lr; }
        g_src[j] = (sy * tc + sx) * PT_HALO + piece * PT_W + idx;
      }
    }
  }
  // ---- this wave's band, this lane's two columns
  const int rb0 = NR * wave;                                // (rows of the band beyond a short tile are computed from stale cells and dropped)
  const int ca = 2 * lane;                                  // tile column of pixel a (b = a + 1)
  const bool lane_valid = ca < TWv;
  // (the lane to the right of a ragged tile's last lane owns the two halo columns: its g of column TW is the edge value lane lv needs)
  const bool cb_a = (c0 + ca <= 0) || (c0 + ca >= w - 1), cb_b = (c0 + ca + 1 <= 0) || (c0 + ca + 1 >= w - 1);
  auto ring_row = [&](int r) -> bool { const int gi = r0 + r; return gi <= 0 || gi >= h - 1; };   // :518-519, clamped rows sit on the ring

  // g of one pixel from its 3 x 3 (pm_wave_k2_kernel.hip's g_of, same order of operations)
  auto g_of = [&](double a00, double a01, double a02, double a10, double a12, double a20, double a21, double a22, bool ring) -> double {
    const double rm = a02 - a00, rr = a12 - a10, rp = a22 - a20;
    const double gx = rm + rr * 2 + rp;
    const double sm = a00 + a01 * 2 + a02;
    const double sp = a20 + a21 * 2 + a22;
    const double gy = sp - sm;
    double g;
    if (FAST) g = rcp_refined(__builtin_fma(__builtin_fma(gx, gx, gy * gy), a.invK2, 1.0));
    else g = 1.0 / (1.0 + (gx * gx + gy * gy) / a.K2);
    return ring ? 1.0 : g;
  };
  struct Row { double2_t p; double w, e; };                 // own two columns, the column to their left, the column to their right
  auto load_row = [&](int r) -> Row {
    Row x;
    x.p = *reinterpret_cast<const double2_t *>(S(r, ca));
    x.w = *S(r, ca - 1);
    x.e = *S(r, ca + 2);
    return x;
  };

  // The Sobel pair of :503-504 is separable (row pass, then column pass -- that IS how the reference computes it): the row pass of a row,
  // d = I(j+1) - I(j-1) and s = I(j-1) + 2 I(j) + I(j+1), serves the three rows of g that need it.  (x * 2 is exact, so fma(x, 2, y) is
  // the reference's y + x * 2 bit for bit, in both flavours.)
  struct HRow { double da, db, sa, sb; };
  auto hrow = [&](const Row &x) -> HRow {
    HRow r;
    r.da = x.p.y - x.w;
    r.db = x.e - x.p.x;
    r.sa = __builtin_fma(x.p.x, 2.0, x.w) + x.p.y;
    r.sb = __builtin_fma(x.p.y, 2.0, x.p.x) + x.e;
    return r;
  };
  // g from the row passes of rows i-1, i, i+1.  k2: 1/K^2 (FAST) or K^2 (STRICT) -- or, on the image's border ring where g == 1
  // (:518-519), 0 / +inf: the same instructions then give exactly 1 (rcp(1) = 1, 1 / (1 + x / inf) = 1), no select per pixel
  auto g_from = [&](double dm, double d0, double dp, double sm, double sp, double k2) -> double {
    const double gx = __builtin_fma(d0, 2.0, dm) + dp;
    const double gy = sp - sm;
    if (FAST) return rcp_refined(__builtin_fma(__builtin_fma(gx, gx, gy * gy), k2, 1.0));
    return 1.0 / (1.0 + (gx * gx + gy * gy) / k2);
  };
  const double k2_ringf = FAST ? 0.0 : __builtin_inf();     // factor of a ring row: k2 * 0 = 0, k2 * inf = inf (k2 > 0; inf * inf = inf, 0 * 0 = 0)
  const double k2a = cb_a ? k2_ringf : (FAST ? a.invK2 : a.K2), k2b = cb_b ? k2_ringf : (FAST ? a.invK2 : a.K2);

  // diagnostic stamps (option "debug_times", tools/pm_resident_timeline.py): 12 words per workgroup, taken around step kStampStep
  constexpr int kStampStep = 5;
  auto stamp = [&](int st_now, int st_want, int slot) {
    if (a.dbg_times && st_now == st_want && tid == 0) a.dbg_times[(size_t)bid * 12 + slot] = __builtin_amdgcn_s_memrealtime();
  };
  const int nsteps = a.res_steps;
  bool gave_up = false;
  u32x4s_t pre[PT_GATHER];                 // the first poll of the next step's ring cells (issued behind the march)
  bool have_pre = false;
#pragma unroll
  for (int j = 0; j < PT_GATHER; ++j) pre[j] = u32x4s_t{0u, 0u, 0u, 0u};
  for (int st = 0; st < nsteps; ++st) {
    stamp(st, kStampStep, 0); stamp(st, kStampStep + 1, 8);
    // ---- 1. the neighbours' borders of the previous step into the halo ring.  No signal to wait for: every entry of the border buffer
    // carries its own tag {launch serial, step}, written with its value in ONE 16-byte store; a thread polls its <= 3 entries until they
    // carry the step it needs (one memory round trip instead of signal-then-data).  Entries of the own tile (the image's border) are
    // copied from LDS.
    if (st > 0) {
      const unsigned char *const hb = reinterpret_cast<const unsigned char *>(a.res_halo) + (size_t)((st - 1) & 1) * ntiles * PT_HALO * 16u;
      const unsigned want_lo = (unsigned)st, want_hi = a.res_serial;
      double v[PT_GATHER];
      unsigned need = 0;
#pragma unroll
      for (int j = 0; j < PT_GATHER; ++j) need |= (g_dst[j] >= 0 && g_src[j] >= 0) ? (1u << j) : 0u;
      for (int i = 0; i < a.res_poll_cap && need; ++i) {
#pragma unroll
        for (int j = 0; j < PT_GATHER; ++j) {
          if (need & (1u << j)) {
            const u32x4s_t c = (i == 0 && have_pre) ? pre[j] : ld_line16(hb, (unsigned)g_src[j] * 16u);    // (the first look was taken when the previous step's march had ended)
            if (c.z == want_lo && c.w == want_hi) { v[j] = __longlong_as_double((long long)(((unsigned long long)c.y << 32) | c.x)); need &= ~(1u << j); }
          }
        }
        if (need) {
          if ((i & 15) == 15 && ld_agent((const unsigned *)&rs->error) != 0u) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (need) { st_agent(&rs->error, 1); s_flag[0] = 1; }     // gave up (or somebody else did): the workgroup leaves behind the barrier
#pragma unroll
      for (int j = 0; j < PT_GATHER; ++j) {
        if (g_dst[j] >= 0 && !(need & (1u << j))) sI[g_dst[j]] = g_src[j] >= 0 ? v[j] : sI[~g_src[j]];
      }
    }
    lds_barrier();     // (LDS only, here and below: the border stores of the step before may still be in flight -- buffer_ops.h; measured: no difference)
    if (s_flag[0] != 0) { gave_up = true; break; }
    stamp(st, kStampStep, 2);                                 // halo ring in LDS

    // ---- 2. the band, from the old tile into registers
    // g of the tile's edge columns -1 (lanes 0..15: band row = lane) and TW (lanes 32..47: band row = lane - 32), per wave
    double gedge;
    {
      const int rr_ = rb0 + ((lane & 31) < NR ? (lane & 31) : NR - 1);
      const int r = rr_ < TH ? rr_ : TH - 1, cc = (lane & 32) ? TWv : -1;
      const int gcol = clampi(c0 + cc, 0, w - 1);
      gedge = g_of(*S(r - 1, cc - 1), *S(r - 1, cc), *S(r - 1, cc + 1), *S(r, cc - 1), *S(r, cc + 1), *S(r + 1, cc - 1), *S(r + 1, cc),
                   *S(r + 1, cc + 1), gcol == 0 || gcol == w - 1 || ring_row(r));
      if ((lane & 31) < 16) sEdge[(lane & 32 ? 16 : 0) + (lane & 15)] = gedge;   // (this wave's own array: no workgroup barrier)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // lane 0 reads the west edge value, lane 63 the east one (a full-width tile: a ragged tile's last lane gets it from the lane to its
    // right, which computes g of the halo column like any other); the other lanes read something and keep the DPP result
    const double *const pw = sEdge, *const pe = sEdge + 16;
    // Rows are rewritten IN PLACE as they are computed -- every reader of the old row has it in registers -- except the band's first two
    // and last two rows, which the bands above and below still read (their rows i +/- 1, i +/- 2): those wait in registers for the
    // workgroup barrier.
    constexpr int NK = NR < 4 ? NR : 4;
    double2_t keep[NK];
    const bool pub_regs = TH == THN && TWv == PT_W && st + 1 < nsteps;   // the border goes out from registers
    unsigned char *const hb_mine = reinterpret_cast<unsigned char *>(a.res_halo) + ((size_t)(st & 1) * ntiles + bid) * PT_HALO * 16u;
    const __amdgpu_buffer_rsrc_t rh = make_rsrc(hb_mine, PT_HALO * 16u);
    const unsigned tag_lo = (unsigned)(st + 1), tag_hi = a.res_serial;      // what the neighbours wait for before their step st + 1
    // left two columns (lane 0 -> pieces 4, 5) and right two (lane 63 -> pieces 6, 7)
    const bool col_lane = pub_regs && (lane == 0 || lane == 63);
    // (through the wave's staging row, so that a store instruction writes 64 NEIGHBOURING entries -- 1 KiB of the border buffer in one
    // piece -- instead of every other one: lane l holds columns 2l and 2l + 1, and stores columns l and 64 + l)
    auto st_row = [&](double2_t v, unsigned piece) {
      *reinterpret_cast<double2_t *>(sRow + ca) = v;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const double lo = sRow[lane], hi = sRow[64 + lane];
      __builtin_amdgcn_raw_buffer_store_b128(tagged(lo, tag_lo, tag_hi), rh, (piece * PT_W + (unsigned)lane) * 16u, 0u, 16 /* sc1 */);
      __builtin_amdgcn_raw_buffer_store_b128(tagged(hi, tag_lo, tag_hi), rh, (piece * PT_W + 64u + (unsigned)lane) * 16u, 0u, 16);
      __builtin_amdgcn_wave_barrier();        // (the next row's writes stay behind these reads)
    };
    {
      Row x0 = load_row(rb0), xp = load_row(rb0 + 1), xq = load_row(rb0 + 2);    // (xq: one row ahead of its use -- LDS latency)
      HRow h0, hp;
      double g0a, g0b, vda, vdb, gsa, gsb;
      {
        const Row xm = load_row(rb0 - 1);
        const HRow hm2 = hrow(load_row(rb0 - 2)), hm = hrow(xm);
        h0 = hrow(x0); hp = hrow(xp);
        const double fa = ring_row(rb0 - 1) ? k2_ringf : 1.0, fb = ring_row(rb0) ? k2_ringf : 1.0;
        const double gma = g_from(hm2.da, hm.da, h0.da, hm2.sa, h0.sa, k2a * fa);      // g of the own columns at rows rb0 - 1 and rb0
        const double gmb = g_from(hm2.db, hm.db, h0.db, hm2.sb, h0.sb, k2b * fa);
        g0a = g_from(hm.da, h0.da, hp.da, hm.sa, hp.sa, k2a * fb);
        g0b = g_from(hm.db, h0.db, hp.db, hm.sb, hp.sb, k2b * fb);
        // what row i shares with row i - 1: the vertical difference and the vertical sum of g between them (:544-547: (cn + c0)(In - I0)
        // of row i is -(cs + c0)(Is - I0) of row i - 1 before rounding; the negation is exact, the sums commute)
        vda = x0.p.x - xm.p.x; vdb = x0.p.y - xm.p.y;
        gsa = g0a + gma; gsb = g0b + gmb;
      }
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int i = rb0 + k;
        // (the two waves of a SIMD: whoever is ahead yields -- csv_resident_kernel.hip, quarter_prio: the older wave was through its band
        // after 4.6 us of a step's 6.6, the younger then ran alone)
        if (a.res_prio && NR >= 4 && k % (NR / 4) == 0) {
          const int q = k / (NR / 4);
          if (q == 0) __builtin_amdgcn_s_setprio(3);
          else if (q == 1) __builtin_amdgcn_s_setprio(2);
          else if (q == 2) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
        const Row xpp = xq;
        if (k + 1 < NR) xq = load_row(i + 3);
        const double gw_e = pw[k], ge_e = pe[k];
        const HRow hpp = hrow(xpp);
        // (wave-uniform select; opaque so that the 2 NR products below are not hoisted out of the step loop into 4 NR registers)
        double fr = ring_row(i + 1) ? k2_ringf : 1.0;
        asm volatile("" : "+v"(fr));
        const double gpa = g_from(h0.da, hp.da, hpp.da, h0.sa, hpp.sa, k2a * fr);
        const double gpb = g_from(h0.db, hp.db, hpp.db, h0.sb, hpp.sb, k2b * fr);
        // g of the columns next to the own two: the neighbour lanes' (DPP); lanes 0 / 63 have no such lane and keep the pre-pass's value
        const double gwa = dpp_from_left_or(gw_e, g0b);
        const double geb = dpp_from_right_or(ge_e, g0a);
        const double vna = xp.p.x - x0.p.x, vnb = xp.p.y - x0.p.y;      // Is - I0
        const double gna = gpa + g0a, gnb = gpb + g0b;                    // cs + c0
        const double hd = x0.p.y - x0.p.x;                                // Ie - I0 of a = -(Iw - I0) of b
        const double gab = g0b + g0a;                                     // ce + c0 of a = cw + c0 of b
        const double gwsa = gwa + g0a, gesb = geb + g0b;
        const double dwa = x0.w - x0.p.x, deb = x0.e - x0.p.y;
        double ox, oy;
        if (FAST) {
          double sa = gna * vna;
          sa = __builtin_fma(gab, hd, sa);
          sa = __builtin_fma(gsa, -vda, sa);
          sa = __builtin_fma(gwsa, dwa, sa);
          ox = __builtin_fma(a.L4, sa, x0.p.x);
          double sb = gnb * vnb;
          sb = __builtin_fma(gesb, deb, sb);
          sb = __builtin_fma(gsb, -vdb, sb);
          sb = __builtin_fma(gab, -hd, sb);
          oy = __builtin_fma(a.L4, sb, x0.p.y);
        } else {
          const double sa = gna * vna + gab * hd + gsa * (-vda) + gwsa * dwa;
          ox = x0.p.x + a.L * sa / 4;   // :544-547
          const double sb = gnb * vnb + gesb * deb + gsb * (-vdb) + gab * (-hd);
          oy = x0.p.y + a.L * sb / 4;
        }
        if (k < 2) keep[k] = double2_t{ox, oy};
        else if (k >= NR - 2) keep[k - (NR - NK)] = double2_t{ox, oy};
        else if (lane_valid && i < TH) *reinterpret_cast<double2_t *>(S(i, ca)) = double2_t{ox, oy};
        // The tile's edge columns leave four rows at a time (round 4): lanes 0 / 63 leave their two values of a row in the wave's staging
        // array, and behind every fourth row sixteen lanes store the quarter's 4 x 4 entries -- four rows of a piece are one 64-byte line of
        // the border buffer, written once instead of four times by two-lane stores (32 store instructions per band and step were 2-lane ones).
        constexpr int QS = NR >= 4 ? 4 : NR;          // rows per batch (eight: no different, 7.15 vs 7.19 us per step at 2048^2)
        if (col_lane) *reinterpret_cast<double2_t *>(sStage + (lane == 0 ? 0 : 8) + 2 * (k % QS)) = double2_t{ox, oy};     // [side][row of the batch][column]
        if (pub_regs && k % QS == QS - 1) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (lane < 4 * QS) {
            const int pc = lane / QS, rr = lane % QS;                 // piece 4 + pc (columns 0, 1, TW-2, TW-1), row of the batch
            const double v = sStage[(pc >> 1) * 8 + 2 * rr + (pc & 1)];
            __builtin_amdgcn_raw_buffer_store_b128(tagged(v, tag_lo, tag_hi), rh, ((unsigned)(4 + pc) * PT_W + (unsigned)(rb0 + k - (QS - 1) + rr)) * 16u, 0u, 16 /* sc1 */);
          }
          __builtin_amdgcn_wave_barrier();        // (the next batch's writes stay behind these reads)
        }
        // (the tile's top two rows leave as soon as they exist: two store instructions less in the burst at the end of the step, which is
        // when the entries everybody waits for -- the last rows -- are on their way)
        if (k == 1 && pub_regs && rb0 == 0) { st_row(keep[0], 0u); st_row(keep[1], 1u); }
        x0 = xp; xp = xpp;
        h0 = hp; hp = hpp;
        g0a = gpa; g0b = gpb;
        vda = vna; vdb = vnb; gsa = gna; gsb = gnb;
        if (k % kRowsPerSched == kRowsPerSched - 1) __builtin_amdgcn_sched_barrier(0);   // two rows at a time may be interleaved, not more (registers)
      }
    }
    // ---- the top / bottom two rows of a full tile straight from the registers: the stores travel while the workgroup meets
    if (pub_regs) {
      if (rb0 == 0 && NR < 2) { st_row(keep[0], 0u); st_row(keep[1], 1u); }      // (NR >= 2: they left behind row 1, above)
      if (rb0 + NR == TH) { st_row(keep[NK - 2], 2u); st_row(keep[NK - 1], 3u); }
    }
    stamp(st, kStampStep, 3);                                 // (thread 0's wave) band computed
    lds_barrier();
    stamp(st, kStampStep, 4);                                 // all waves
    // the first look at the NEXT step's ring cells: asked for here, taken when that step begins -- the round trip passes while the rows at the
    // band's ends are rewritten and the workgroup meets again
    have_pre = false;
    if (st + 1 < nsteps) {
      const unsigned char *const hbn = reinterpret_cast<const unsigned char *>(a.res_halo) + (size_t)(st & 1) * ntiles * PT_HALO * 16u;
#pragma unroll
      for (int j = 0; j < PT_GATHER; ++j) if (g_dst[j] >= 0 && g_src[j] >= 0) pre[j] = ld_line16(hbn, (unsigned)g_src[j] * 16u);
      have_pre = true;
    }
    // ---- 3. the band's first and last rows replace the old ones
    if (lane_valid) {
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        if ((k < 2 || k >= NR - 2) && rb0 + k < TH) *reinterpret_cast<double2_t *>(S(rb0 + k, ca)) = keep[k < 2 ? k : k - (NR - NK)];
      }
    }
    lds_barrier();
    stamp(st, kStampStep, 5);                                 // tile rewritten
    // ---- 4. the border of a tile that is not full: from LDS
    if (st + 1 < nsteps && !pub_regs) {
      for (int q = tid; q < PT_HALO; q += PT_THREADS) {
        const int piece = q / PT_W, k = q % PT_W;
        const int kr = k < TH ? k : TH - 1;
        double v;
        if (piece < 2) v = *S(piece, k);                             // top two rows
        else if (piece < 4) v = *S(TH - 4 + piece, k);               // bottom two rows: TH - 2, TH - 1
        else if (piece < 6) v = *S(kr, piece - 4);                   // left two columns
        else v = *S(kr, TWv - 8 + piece);                            // right two columns: TWv - 2, TWv - 1
        __builtin_amdgcn_raw_buffer_store_b128(tagged(v, tag_lo, tag_hi), rh, (unsigned)q * 16u, 0u, 16 /* sc1 */);
      }
    }
    stamp(st, kStampStep, 6);                                 // border on its way
  }
  if (gave_up) return;
  // ---- leave: the tile into the output plane (never the plane the launch read from: a neighbour may still be loading its ring)
  for (int q = tid; q < TH * (PT_W / 2); q += PT_THREADS) {
    const int r = q / (PT_W / 2), c = 2 * (q % (PT_W / 2));
    if (c < TWv) *reinterpret_cast<double2_t *>(a.out + (size_t)(r0 + r) * w + c0 + c) = *reinterpret_cast<const double2_t *>(S(r, c));
  }
}

}  // namespace

size_t cvh_pm_resident_lds_bytes() { return PmResSmem::bytes; }
int cvh_pm_resident_halo_doubles() { return 2 * PT_HALO; }   // 16-byte entries {value, tag}

namespace {
typedef void (*PmResKernel)(const CvhPmArgs);
PmResKernel pm_res_kernel(int fast, int nr)
{
  switch (nr) {
    case 2: return fast ? pm_resident_kernel<true, 2> : pm_resident_kernel<false, 2>;
    case 4: return fast ? pm_resident_kernel<true, 4> : pm_resident_kernel<false, 4>;
    case 8: return fast ? pm_resident_kernel<true, 8> : pm_resident_kernel<false, 8>;
    case 16: return fast ? pm_resident_kernel<true, 16> : pm_resident_kernel<false, 16>;
  }
  return nullptr;
}
}  // namespace

// Workgroups of the resident kernel one CU holds (0: not launchable): the least over the instantiations.
int cvh_pm_resident_blocks_per_cu()
{
  static int cached = -1;
  if (cached >= 0) return cached;
  int least = 1 << 30;
  for (int fast = 0; fast < 2; ++fast)
    for (int nr = 2; nr <= 16; nr *= 2) {
      const void *k = reinterpret_cast<const void *>(pm_res_kernel(fast, nr));
      int n = 0;
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PmResSmem::bytes) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, PT_THREADS, PmResSmem::bytes) != hipSuccess) {
        (void)hipGetLastError();
        return cached = 0;
      }
      if (n < least) least = n;
    }
  return cached = least;
}

hipError_t cvh_launch_pm_resident(const CvhPmArgs &a, hipStream_t s)
{
  const PmResKernel kern = pm_res_kernel(a.fast, a.res_band_rows);
  if (!kern) return hipErrorInvalidValue;
  if (a.note) {
    cvh_fill_note(a.note, (unsigned)(a.tiles_x * a.tiles_y), PT_THREADS, PmResSmem::bytes, "pm_resident_kernel<%s, %d>", a.fast ? "true" : "false", a.res_band_rows);
    return hipSuccess;
  }
  CvhPmArgs copy = a;
  void *params[] = {&copy};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void *>(kern), dim3(a.tiles_x * a.tiles_y), dim3(PT_THREADS), params, (unsigned)PmResSmem::bytes, s);
}
