// pm_wave_k2_kernel.hip — TWO Perona-Malik time steps per launch (gfx950, wave64), time-skewed inside each wave.
//
// perona_malik (src/main.cpp:478-560) is `trips` dependent sweeps over a plane; at 2048^2 one sweep moves 64 MiB and takes
// ~14 us of kernel time, much of it per-launch (prologue, pipeline fill, tail).  Here a wave marches down its strip ONCE and
// carries two pipeline stages: stage 1 turns rows of I(t) into rows of I(t+1) exactly as pm_wave_kernel does (same
// arithmetic, same order of operations: STRICT stays bit-exact), stage 2 follows 4 rows behind and turns those rows -- taken
// from registers, never written to memory -- into rows of I(t+2).  Half the launches, half the HBM/L2 traffic per step.
//
//   lane  <->  column 56 wc - 4 + lane : 4 halo columns on either side (2 per stage), 56 output columns per wave
//   stage 1 rows [s0 - 2, s1 + 3), stage 2 rows [s0, s1), 4 rows behind   (the 6 extra stage-1 rows are the price per strip)
//   x-neighbours of I and of g go through per-wave LDS row slots, one set per stage; g(i-1..i+1) of the own column and the
//   own column of I (8-row ring for stage 1: 4 rows live + 4 loads in flight; 5 live rows for stage 2) stay in registers.
// Border rules (src/main.cpp:518-519, :527-530): g == 1 on the image's border ring (from the GLOBAL row / column, the same
// in both stages); neighbour indices clamp.  For stage 1 the clamped loads provide that; stage 2 reads I(t+1), whose
// out-of-image rows / columns were never computed, so it clamps explicitly (own row at the top / bottom edge, own lane at
// the left / right edge).  Redundant halo rows / columns are computed from the same inputs in the same order by every wave
// that needs them: identical values, deterministic.
#include "buffer_ops.h"
#include "csv_device.h"

using namespace cvh_dev;

namespace {

constexpr int P2C = 56;   // output columns per wave

struct PmStage {
  double q[8];            // own column: ring index of row r is (r - base + 2) & 7
  double nw[4], ne[4];    // west / east neighbours of I, rows (r - base + 2) & 3
  double gr[4];           // g of the own column
  double gw, ge;          // g(i, col -/+ 1) of the row being produced
};

// POL 1: stores carry sc1 (agent-scope write-through) -- 2048^2 13.1 -> 12.7 us/step while the two state planes live in the Infinity
// Cache; plain stores (POL 0) beyond it, where write-through costs (CvhPmArgs::pol, chosen by footprint in api.hip; cf. wave2_device.h)
template <bool FAST, int POL>
__global__ __launch_bounds__(CVH_BLOCK, 4) void pm_wave_k2_kernel(const CvhPmArgs a)
{
  __shared__ double sx[4][16 * 64];   // per wave: stage 1 {4 rows of I, 4 rows of g}, stage 2 {4, 4}
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = a.h, w = a.w;
  const int nwc = a.tiles_x, nbc = (nwc + 3) >> 2;
  const int wc = (blockIdx.x % nbc) * 4 + wave, ws = blockIdx.x / nbc;
  const int s0 = ws * a.strip_rows;
  if (wc >= nwc || s0 >= h) return;
  const int s1 = (s0 + a.strip_rows) < h ? (s0 + a.strip_rows) : h;
  const int rows = s1 - s0;
  const int col = P2C * wc - 4 + lane;
  const int colc = clampi(col, 0, w - 1);
  const bool colborder = (colc == 0) || (colc == w - 1);
  const bool lane_out = lane >= 4 && lane < 60 && col < w;
  double *sI1 = sx[wave], *sG1 = sI1 + 4 * 64, *sI2 = sG1 + 4 * 64, *sG2 = sI2 + 4 * 64;
  // stage 1: the neighbour lanes hold the clamped columns themselves; stage 2: clamp at the image's edges explicitly
  const int lw1 = lane > 0 ? lane - 1 : 0, le1 = lane < 63 ? lane + 1 : 63;
  const int lw2 = (lane == 0 || col <= 0) ? lane : lane - 1, le2 = (lane == 63 || col >= w - 1) ? lane : lane + 1;
  const unsigned rowbytes = (unsigned)w * 8u;
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out, (unsigned)h * rowbytes);
  const unsigned voff_st = lane_out ? (unsigned)col * 8u : kOobOffset;

  auto LD = [&](int r) -> double { const double *rp = a.in + (size_t)clampi(r, 0, h - 1) * w; return rp[colc]; };
  auto fence = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto g_of = [&](double a00, double a01, double a02, double a10, double a12, double a20, double a21, double a22,
                  int gi) -> double {
    const double rm = a02 - a00, r0 = a12 - a10, rp = a22 - a20;
    const double gx = rm + r0 * 2 + rp;
    const double sm = a00 + a01 * 2 + a02;
    const double sp = a20 + a21 * 2 + a22;
    const double gy = sp - sm;
    double g;
    if (FAST) g = rcp_refined(__builtin_fma(__builtin_fma(gx, gx, gy * gy), a.invK2, 1.0));
    else g = 1.0 / (1.0 + (gx * gx + gy * gy) / a.K2);
    return (colborder || gi <= 0 || gi >= h - 1) ? 1.0 : g;   // :518-519 (clamped rows/cols sit on the ring)
  };

  // What both stages do before their first row: rows base-2 .. base+2 of the own column are in S.q[0..4]; neighbours of
  // rows base-2 .. base+2, g(base-1), g(base) and the neighbours of g(base) are derived through the stage's LDS slots.
  auto stage_prologue = [&](PmStage &S, double *sI, double *sG, int lw, int le, int base) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sI[j * 64 + lane] = S.q[j];
    fence();
#pragma unroll
    for (int j = 0; j < 4; ++j) { S.nw[j] = sI[j * 64 + lw]; S.ne[j] = sI[j * 64 + le]; }
    S.gr[1] = g_of(S.nw[0], S.q[0], S.ne[0], S.nw[1], S.ne[1], S.nw[2], S.q[2], S.ne[2], base - 1);
    S.gr[2] = g_of(S.nw[1], S.q[1], S.ne[1], S.nw[2], S.ne[2], S.nw[3], S.q[3], S.ne[3], base);
    S.gr[0] = 1.0; S.gr[3] = 1.0;
    fence();
    sI[0 * 64 + lane] = S.q[4];            // row base-2 has served its purpose: its slot takes row base+2
    sG[2 * 64 + lane] = S.gr[2];
    fence();
    S.nw[0] = sI[0 * 64 + lw]; S.ne[0] = sI[0 * 64 + le];
    S.gw = sG[2 * 64 + lw]; S.ge = sG[2 * 64 + le];
  };

  // One row of one stage, in two halves around ONE wave-level LDS fence.  `stage_publish` writes row i+3 of the own column
  // and g(i+1) into the stage's LDS slots (ring phase k = (i - base) & 7, compile-time after unrolling; S.q must hold rows
  // i-1 .. i+3); `stage_finish` reads their x-neighbours back and produces I_next(i, own column) -- pm_wave_kernel's loop body
  // (pm_kernels.hip), stage-agnostic.  The two stages of an iteration are independent of each other (stage 2 consumes what
  // stage 1 produced in the PREVIOUS iteration), so both publish, ONE fence follows, both finish: one LDS round trip per
  // iteration instead of two, and twice the independent arithmetic around it.
  auto stage_publish = [&](PmStage &S, double *sI, double *sG, int k, int i, double &gnew) {
    const double I0 = S.q[(k + 2) & 7], Ipp = S.q[(k + 4) & 7];
    sI[((k + 1) & 3) * 64 + lane] = S.q[(k + 5) & 7];   // publish row i+3, its neighbours are fetched after the fence
    gnew = g_of(S.nw[(k + 2) & 3], I0, S.ne[(k + 2) & 3], S.nw[(k + 3) & 3], S.ne[(k + 3) & 3],
                S.nw[(k + 0) & 3], Ipp, S.ne[(k + 0) & 3], i + 1);
    sG[((k + 3) & 3) * 64 + lane] = gnew;
  };
  auto stage_finish = [&](PmStage &S, double *sI, double *sG, int lw, int le, int k, int i, bool second, double gnew) -> double {
    double Im = S.q[(k + 1) & 7];
    const double I0 = S.q[(k + 2) & 7];
    double Ip = S.q[(k + 3) & 7];
    if (second) {                            // rows of I(t+1) outside the image: index clamp (:527-530), wave-uniform
      if (i <= 0) Im = I0;
      if (i >= h - 1) Ip = I0;
    }
    const double nw_n = sI[((k + 1) & 3) * 64 + lw], ne_n = sI[((k + 1) & 3) * 64 + le];
    const double gw_n = sG[((k + 3) & 3) * 64 + lw], ge_n = sG[((k + 3) & 3) * 64 + le];
    const double cn = S.gr[(k + 1) & 3], c0 = S.gr[(k + 2) & 3], cs = gnew;
    const double Iw = S.nw[(k + 2) & 3], Ie = S.ne[(k + 2) & 3];
    double outv;
    if (FAST) {
      double s = (cs + c0) * (Ip - I0);
      s = __builtin_fma(S.ge + c0, Ie - I0, s);
      s = __builtin_fma(cn + c0, Im - I0, s);
      s = __builtin_fma(S.gw + c0, Iw - I0, s);
      outv = __builtin_fma(a.L4, s, I0);
    } else {
      const double s = (cs + c0) * (Ip - I0) + (S.ge + c0) * (Ie - I0) + (cn + c0) * (Im - I0) + (S.gw + c0) * (Iw - I0);
      outv = I0 + a.L * s / 4;  // :544-547
    }
    S.nw[(k + 1) & 3] = nw_n; S.ne[(k + 1) & 3] = ne_n;   // row i+3
    S.gr[(k + 3) & 3] = gnew;                              // row i+1
    S.gw = gw_n; S.ge = ge_n;                              // g(i+1, col -/+ 1) for the next row
    return outv;
  };
  auto stage_row = [&](PmStage &S, double *sI, double *sG, int lw, int le, int k, int i, bool second) -> double {
    double gnew;
    stage_publish(S, sI, sG, k, i, gnew);
    fence();
    return stage_finish(S, sI, sG, lw, le, k, i, second, gnew);
  };

  PmStage A, B;
  const int base1 = s0 - 2;                  // first row stage 1 produces
  // ---- stage 1 prologue: rows base1-2 .. base1+5 of I(t) (ring slots 0..7), then slot 0 moves on to row base1+6
#pragma unroll
  for (int j = 0; j < 8; ++j) A.q[j] = LD(base1 - 2 + j);
  stage_prologue(A, sI1, sG1, lw1, le1, base1);
  A.q[0] = LD(base1 + 6);
  // ---- stage 1 alone for 6 rows: I(t+1) rows s0-2 .. s0+2 = stage 2's ring slots 0..4, row s0+3 waits in v1
  double v1 = 0.0;
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const int i1 = base1 + t;
    const double o = stage_row(A, sI1, sG1, lw1, le1, t & 7, i1, false);
    if (t < 5) B.q[t] = o; else v1 = o;
    A.q[(t + 1) & 7] = LD(i1 + 7);           // row i1-1 is dead: its slot takes row i1+7
  }
  B.q[5] = 0.0; B.q[6] = 0.0; B.q[7] = 0.0;
  stage_prologue(B, sI2, sG2, lw2, le2, s0);
  // ---- both stages: stage 2 produces row s0+j from what stage 1 produced up to the PREVIOUS iteration (v1 = row s0+j+3),
  // stage 1 produces row s0+j+4
  for (int jb = 0; jb < rows; jb += 8) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int i2 = s0 + jb + kk, i1 = i2 + 4;
      const int k1 = (6 + kk) & 7, k2 = kk;
      B.q[(k2 + 5) & 7] = v1;                // row i2+3 of I(t+1)
      double g1, g2;
      stage_publish(A, sI1, sG1, k1, i1, g1);
      stage_publish(B, sI2, sG2, k2, i2, g2);
      fence();
      v1 = stage_finish(A, sI1, sG1, lw1, le1, k1, i1, false, g1);
      A.q[(k1 + 1) & 7] = LD(i1 + 7);
      const double v2 = stage_finish(B, sI2, sG2, lw2, le2, k2, i2, true, g2);
      // lanes without an output column are dropped by the hardware (offset beyond the buffer), rows past the strip
      // end by an empty resource
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v2), i2 < s1 ? rout : make_rsrc(a.out, 0u), voff_st, (unsigned)i2 * rowbytes,
                                            POL ? 16 : 0);
    }
  }
}

}  // namespace

int cvh_pm_wave_k2_cols() { return P2C; }

hipError_t cvh_launch_pm_wave_k2(const CvhPmArgs &a, hipStream_t s)
{
  const int nbc = (a.tiles_x + 3) / 4, nstr = (a.h + a.strip_rows - 1) / a.strip_rows;
  if (a.fast && a.pol) CVH_LAUNCH((pm_wave_k2_kernel<true, 1>), nbc * nstr, 0, s, a, "pm_wave_k2_kernel<true, 1>");
  else if (a.fast) CVH_LAUNCH((pm_wave_k2_kernel<true, 0>), nbc * nstr, 0, s, a, "pm_wave_k2_kernel<true, 0>");
  else if (a.pol) CVH_LAUNCH((pm_wave_k2_kernel<false, 1>), nbc * nstr, 0, s, a, "pm_wave_k2_kernel<false, 1>");
  else CVH_LAUNCH((pm_wave_k2_kernel<false, 0>), nbc * nstr, 0, s, a, "pm_wave_k2_kernel<false, 0>");
  return hipGetLastError();
}
