// pm_kernels.hip — Perona-Malik explicit diffusion step for gfx950 (src/main.cpp:478-560).
//
// One launch per time step and channel plane: a 256-thread workgroup stages a
// (16+4) x (64+4) FP64 tile of I (halo 2: the flux needs g at the four neighbours, g needs
// the 3x3 Sobel) in LDS, computes the edge-stopping coefficient g on the (16+2) x (64+2)
// ring once per cell into LDS, then updates its 16 x 64 pixels.  FP64 state ping-pongs in
// HBM; the uint8 rounding (:551) is only observable after the last step and is done by
// pm_store_kernel (round-half-even, clamp).
#include "csv_device.h"

using cvh_dev::rcp_refined;

namespace {

constexpr int PTW = 64, PTH = 32;
constexpr int IP = PTW + 4;  // pitch of the I tile
constexpr int GP = PTW + 2;  // pitch of the g tile

using cvh_dev::clampi;

// FAST: reciprocal + cubic refinement (<= 1 ulp) and FMAs instead of the two IEEE divisions; the
// final uint8 image can then differ from the strict one only where a value sits within ~1e-13 of a
// rounding boundary.  STRICT rounds every operation as the reference's x86-64 build.
template <bool FAST>
__global__ __launch_bounds__(CVH_BLOCK) void pm_step_kernel(const CvhPmArgs a)
{
  __shared__ double sI[(PTH + 4) * IP];
  __shared__ double sg[(PTH + 2) * GP];
  const int tid = threadIdx.x;
  const int h = a.h, w = a.w;
  const int bx = blockIdx.x % a.tiles_x, by = blockIdx.x / a.tiles_x;
  const int i0 = by * PTH, j0 = bx * PTW;

  // I tile: rows i0-2 .. i0+PTH+1, cols j0-2 .. j0+PTW+1; clamped indices = the reference's
  // neighbour clamps (:527-530)
  {
    // all loads of the tile are issued before the first LDS write (no serialised round trips)
    constexpr int N = (PTH + 4) * IP, NRD = (N + CVH_BLOCK - 1) / CVH_BLOCK;
    double v[NRD];
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      const int idx = rd * CVH_BLOCK + tid;
      const int q = idx < N ? idx : N - 1;
      const int r = q / IP, c = q - r * IP;
      const int gi = clampi(i0 - 2 + r, 0, h - 1), gj = clampi(j0 - 2 + c, 0, w - 1);
      v[rd] = a.in[(size_t)gi * w + gj];
    }
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      const int idx = rd * CVH_BLOCK + tid;
      if (idx < N) sI[idx] = v[rd];
    }
  }
  __syncthreads();

  // g on rows i0-1 .. i0+PTH, cols j0-1 .. j0+PTW (:513-522).  Each lane marches down one
  // column keeping the 3x3 window of I in registers (3 LDS reads per cell instead of 8); the
  // two extra columns are done cell-per-lane by the wave that has the fewest rows.
  const int tx = tid & 63, ty = tid >> 6;
  auto g_of = [&](double a00, double a01, double a02, double a10, double a12, double a20, double a21, double a22,
                  int gi, int gj) -> double {
    // cv::Sobel ksize 3: row pass then column pass (see oracle/cv_oracle.c)
    const double rm = a02 - a00, r0 = a12 - a10, rp = a22 - a20;
    const double gx = rm + r0 * 2 + rp;
    const double sm = a00 + a01 * 2 + a02;
    const double sp = a20 + a21 * 2 + a22;
    const double gy = sp - sm;
    double g;
    if (FAST) g = rcp_refined(__builtin_fma(__builtin_fma(gx, gx, gy * gy), a.invK2, 1.0));
    else g = 1.0 / (1.0 + (gx * gx + gy * gy) / a.K2);
    // image border ring (and anything clamped onto it) keeps g = 1
    return (gi > 0 && gi < h - 1 && gj > 0 && gj < w - 1) ? g : 1.0;
  };
  {
    constexpr int GR = PTH + 2;                 // g rows
    constexpr int RPW = (GR + 3) / 4;           // rows per wave (9, 9, 9, 7 for PTH = 32)
    const int r_lo = ty * RPW, r_hi = (r_lo + RPW) < GR ? (r_lo + RPW) : GR;
    // g cell (r, c) sits at I-tile (r+1, c+1); window rows r, r+1, r+2, cols c, c+1, c+2
    const double *p = &sI[r_lo * IP + tx];
    double a00 = p[0], a01 = p[1], a02 = p[2];
    double a10 = p[IP], a11 = p[IP + 1], a12 = p[IP + 2];
    (void)a11;
    for (int r = r_lo; r < r_hi; ++r) {
      const double *pn = &sI[(r + 2) * IP + tx];
      const double a20 = pn[0], a21 = pn[1], a22 = pn[2];
      sg[r * GP + tx] = g_of(a00, a01, a02, a10, a12, a20, a21, a22, i0 - 1 + r, j0 - 1 + tx);
      a00 = a10; a01 = a11; a02 = a12; a10 = a20; a11 = a21; a12 = a22;
    }
    if (ty == 3) {                               // columns 64, 65: (PTH+2) x 2 cells, one per lane
      for (int cell = tx; cell < GR * 2; cell += 64) {
        const int r = cell >> 1, c = 64 + (cell & 1);
        const double *q = &sI[r * IP + c];
        sg[r * GP + c] = g_of(q[0], q[1], q[2], q[IP], q[IP + 2], q[2 * IP], q[2 * IP + 1], q[2 * IP + 2],
                              i0 - 1 + r, j0 - 1 + c);
      }
    }
  }
  __syncthreads();

  // update: lane tx owns column tx, wave ty rows ty*8 .. ty*8+7, marching down with the
  // vertical neighbours of I and g in registers (:524-548)
  const int gj = j0 + tx;
  {
    constexpr int RPW = PTH / 4;
    const int rb = ty * RPW;
    const double *pI = &sI[(rb + 2) * IP + (tx + 2)];
    const double *pg = &sg[(rb + 1) * GP + (tx + 1)];
    double In = pI[-IP], I0 = pI[0];             // rows rb-1, rb
    double cn = pg[-GP], c0 = pg[0];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int gi = i0 + rb + q;
      const double Is = pI[(q + 1) * IP], cs = pg[(q + 1) * GP];
      const double Ie = pI[q * IP + 1], Iw = pI[q * IP - 1];
      const double ce = pg[q * GP + 1], cw = pg[q * GP - 1];
      double outv;
      if (FAST) {
        double s = (cs + c0) * (Is - I0);
        s = __builtin_fma(ce + c0, Ie - I0, s);
        s = __builtin_fma(cn + c0, In - I0, s);
        s = __builtin_fma(cw + c0, Iw - I0, s);
        outv = __builtin_fma(a.L4, s, I0);
      } else {
        const double s = (cs + c0) * (Is - I0) + (ce + c0) * (Ie - I0) + (cn + c0) * (In - I0) + (cw + c0) * (Iw - I0);
        outv = I0 + a.L * s / 4;  // :544-547
      }
      if (gi < h && gj < w) a.out[(size_t)gi * w + gj] = outv;
      In = I0; I0 = Is; cn = c0; c0 = cs;
    }
  }
}

// ---- wave-streaming variant (same arithmetic, no workgroup barriers) -----------------------
// Each WAVE owns 60 output columns (lanes 0,1 and 62,63 are halo columns: the flux needs g one
// column out, g needs I one column further) and marches down `strip_rows` rows.  The lane's own
// column of I lives in a register ring of 8 rows (4 live + 4 in flight), x-neighbours of I and of
// g go through per-wave LDS row slots written one step ahead of their use, g(i-1..i+1) of the own
// column stay in registers.  Halo lanes compute like the others; only their stores are masked.
constexpr int PWC = 60;

// An "s" asm operand must really live in SGPRs: pin a wave-uniform pointer there.
__device__ __forceinline__ const double *uniform_ptr(const double *p)
{
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const double *)(((unsigned long long)hi << 32) | lo);
}

template <bool FAST>
__global__ __launch_bounds__(CVH_BLOCK) void pm_wave_kernel(const CvhPmArgs a)
{
  __shared__ double sx[4][8 * 64];   // per wave: 4 row slots of I + 4 row slots of g
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = a.h, w = a.w;
  const int nwc = a.tiles_x, nbc = (nwc + 3) >> 2;
  const int wc = (blockIdx.x % nbc) * 4 + wave, ws = blockIdx.x / nbc;
  const int s0 = ws * a.strip_rows;
  if (wc >= nwc || s0 >= h) return;
  const int s1 = (s0 + a.strip_rows) < h ? (s0 + a.strip_rows) : h;
  const int col = PWC * wc - 2 + lane;
  const int colc = clampi(col, 0, w - 1);
  const bool colborder = (colc == 0) || (colc == w - 1);
  const bool lane_out = lane >= 2 && lane < 62 && col < w;
  const unsigned long long store_mask = __ballot(lane_out);
  const unsigned ooff32 = (unsigned)colc * 8u;
  double *sI = sx[wave], *sG = sx[wave] + 4 * 64;
  // neighbour addresses (halo lanes 0 / 63 read their own entry: their results are never used)
  const int lw = lane > 0 ? lane - 1 : 0, le = lane < 63 ? lane + 1 : 63;

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

  // ---- prologue.  Ring q[j & 7] holds row s0-2+j of the own column for j = 0..7.
  double q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = LD(s0 - 2 + j);
  // neighbour rings: nw/ne[j & 3] = I(row s0-2+j, col -/+ 1); filled for j = 0..3
  double nw[4], ne[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) sI[j * 64 + lane] = q[j];
  fence();
#pragma unroll
  for (int j = 0; j < 4; ++j) { nw[j] = sI[j * 64 + lw]; ne[j] = sI[j * 64 + le]; }
  // g ring: gr[j & 3] = g(row s0-2+j); rows s0-1 (j=1) and s0 (j=2) are needed before step 0
  double gr[4];
  gr[1] = g_of(nw[0], q[0], ne[0], nw[1], ne[1], nw[2], q[2], ne[2], s0 - 1);
  gr[2] = g_of(nw[1], q[1], ne[1], nw[2], ne[2], nw[3], q[3], ne[3], s0);
  gr[0] = 1.0; gr[3] = 1.0;
  fence();
  // row s0-2 has served its purpose: its slot (index 0) now takes row s0+2, needed by step 0
  sI[0 * 64 + lane] = q[4];
  q[0] = LD(s0 + 6);                       // ... and its register slot takes row s0+6 (ring = rows i-1 .. i+6)
  sG[2 * 64 + lane] = gr[2];
  fence();
  nw[0] = sI[0 * 64 + lw]; ne[0] = sI[0 * 64 + le];
  double gw = sG[2 * 64 + lw], ge = sG[2 * 64 + le];   // g(s0, col -/+ 1)

  for (int ib = s0; ib < s1; ib += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = ib + k;                 // output row; ring index of row r is (r - s0 + 2) & 7
      // slots: row i-1 -> (k+1)&7, i -> (k+2)&7, i+1 -> (k+3)&7, i+2 -> (k+4)&7, i+3 -> (k+5)&7
      const double Im = q[(k + 1) & 7], I0 = q[(k + 2) & 7], Ip = q[(k + 3) & 7], Ipp = q[(k + 4) & 7];
      // publish row i+3 of this column, fetch its neighbours (used from the next step on)
      sI[((k + 1) & 3) * 64 + lane] = q[(k + 5) & 7];
      // g(i+1): window rows i, i+1, i+2
      const double gnew = g_of(nw[(k + 2) & 3], I0, ne[(k + 2) & 3], nw[(k + 3) & 3], ne[(k + 3) & 3],
                               nw[(k + 0) & 3], Ipp, ne[(k + 0) & 3], i + 1);
      sG[((k + 3) & 3) * 64 + lane] = gnew;
      fence();
      const double nw_n = sI[((k + 1) & 3) * 64 + lw], ne_n = sI[((k + 1) & 3) * 64 + le];
      const double gw_n = sG[((k + 3) & 3) * 64 + lw], ge_n = sG[((k + 3) & 3) * 64 + le];
      const double cn = gr[(k + 1) & 3], c0 = gr[(k + 2) & 3], cs = gnew;
      const double Iw = nw[(k + 2) & 3], Ie = ne[(k + 2) & 3];
      double outv;
      if (FAST) {
        double s = (cs + c0) * (Ip - I0);
        s = __builtin_fma(ge + c0, Ie - I0, s);
        s = __builtin_fma(cn + c0, Im - I0, s);
        s = __builtin_fma(gw + c0, Iw - I0, s);
        outv = __builtin_fma(a.L4, s, I0);
      } else {
        const double s = (cs + c0) * (Ip - I0) + (ge + c0) * (Ie - I0) + (cn + c0) * (Im - I0) + (gw + c0) * (Iw - I0);
        outv = I0 + a.L * s / 4;  // :544-547
      }
      {
        // masked store in assembly (see csv_wave_kernel.hip for why); rows past the strip end are dropped
        const double *ob = uniform_ptr(a.out + (size_t)i * w);
        unsigned long long exec_keep;
        if (i < s1)
          asm volatile("s_mov_b64 %0, exec\n\t"
                       "s_mov_b64 exec, %4\n\t"
                       "s_nop 4\n\t"
                       "global_store_dwordx2 %1, %2, %3\n\t"
                       "s_mov_b64 exec, %0"
                       : "=&s"(exec_keep) : "v"(ooff32), "v"(outv), "s"(ob), "s"(store_mask) : "memory");
      }
      // rotate: row i-1 is dead -> its slot takes row i+7; neighbour / g rings advance
      q[(k + 1) & 7] = LD(i + 7);
      nw[(k + 1) & 3] = nw_n; ne[(k + 1) & 3] = ne_n;   // row i+3
      gr[(k + 3) & 3] = gnew;                            // row i+1
      gw = gw_n; ge = ge_n;                              // g(i+1, col -/+ 1) for the next step
    }
  }
}

__global__ void pm_load_kernel(const uint8_t *plane, double *state, size_t n)
{
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x)
    state[q] = (double)plane[q];  // :495-496
}

__global__ void pm_store_kernel(const double *state, uint8_t *plane, size_t n)
{
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    const double r = rint(state[q]);  // cvRound: round half to even (:551)
    plane[q] = (uint8_t)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
  }
}

inline int flat_grid(size_t n)
{
  size_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

void cvh_pm_grid(int h, int w, int *tiles_x, int *tiles_y)
{
  *tiles_x = (w + PTW - 1) / PTW;
  *tiles_y = (h + PTH - 1) / PTH;
}

hipError_t cvh_launch_pm_load(const uint8_t *plane, double *state, size_t n, hipStream_t s)
{
  hipLaunchKernelGGL(pm_load_kernel, dim3(flat_grid(n)), dim3(256), 0, s, plane, state, n);
  return hipGetLastError();
}

hipError_t cvh_launch_pm_wave(const CvhPmArgs &a, hipStream_t s)
{
  const int nbc = (a.tiles_x + 3) / 4, nstr = (a.h + a.strip_rows - 1) / a.strip_rows;
  if (a.fast) CVH_LAUNCH(pm_wave_kernel<true>, nbc * nstr, 0, s, a, "pm_wave_kernel<true>");
  else CVH_LAUNCH(pm_wave_kernel<false>, nbc * nstr, 0, s, a, "pm_wave_kernel<false>");
  return hipGetLastError();
}

int cvh_pm_wave_cols() { return PWC; }

hipError_t cvh_launch_pm_step(const CvhPmArgs &a, hipStream_t s)
{
  if (a.fast) CVH_LAUNCH(pm_step_kernel<true>, a.tiles_x * a.tiles_y, 0, s, a, "pm_step_kernel<true>");
  else CVH_LAUNCH(pm_step_kernel<false>, a.tiles_x * a.tiles_y, 0, s, a, "pm_step_kernel<false>");
  return hipGetLastError();
}

hipError_t cvh_launch_pm_store(const double *state, uint8_t *plane, size_t n, hipStream_t s)
{
  hipLaunchKernelGGL(pm_store_kernel, dim3(flat_grid(n)), dim3(256), 0, s, state, plane, n);
  return hipGetLastError();
}
