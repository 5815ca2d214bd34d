// csv_kernels.hip — the fused Chan-Sandberg-Vese level-set step for gfx950 (CDNA4, wave64).
//
// One launch per iteration of the reference's timestep loop (src/main.cpp:963-1001):
//   * a 256-thread workgroup stages a (R+3) x (256+4) FP64 tile of u (halo 2 left/top,
//     1 right/bottom — the 7-point cross of curvature(), :342-375) in LDS with coalesced
//     row-major reads; out-of-image halo cells are filled by index clamping, which IS
//     BORDER_REPLICATE on u (:351-354);
//   * each thread owns one column and marches down R rows keeping u(i-1), u(i), u(i+1) and
//     the previous row's normalised y-gradient in registers; the normalised x-gradient of
//     the left neighbour comes from the adjacent lane by DPP (wave_shr:1), the wave's own
//     left-edge column is normalised once per tile with lanes mapped to rows;
//   * region terms (:299-312,:979), the addWeighted combine (:985), delta_eps (:204-210,
//     the ParallelPixelFunction map of :988-992), the update (:994) and ||u_diff||^2 (:993)
//     are fused behind the stencil; H_eps(u_new) (:188-194) is evaluated once per pixel and
//     its I-weighted sums are reduced wave -> LDS -> one partial row per workgroup;
//   * the last-arriving workgroup (agent-scope ticket) adds the partial rows in a fixed
//     order and publishes next iteration's c1/c2 (:973-974), the norm and the sticky stop
//     flag (:1000).  Summation order is fixed => bitwise reproducible run to run.
//
// Compile with -ffp-contract=off: the STRICT flavour then rounds every per-pixel operation
// exactly as the reference's x86-64 -O3 build (no FMA); the FAST flavour asks for FMAs
// explicitly.
#include "csv_device.h"

using namespace cvh_dev;

namespace {

// ---- tile staging ---------------------------------------------------------------------
// Asynchronous global -> LDS copy of the (ROWS x PITCH) tile in 16-byte pieces
// (global_load_lds_dwordx4: no VGPR round trip, every piece of the tile in flight at once).
// One wave-instruction writes 64 consecutive pieces = 1 KiB of the linear LDS image; the
// SOURCE address is per lane (row/column of the piece, clamped into the image).  Needs an
// even width (16-byte aligned rows).  Out-of-image columns are patched at read time.
template <int ROWS>
__device__ __forceinline__ void stage_tile_dma(const double *u_in, double *su, int i0, int j0, int h,
                                               int w, int tid)
{
  constexpr int CPR = PITCH / 2;  // 16-byte pieces per tile row
  constexpr int NPIECE = ROWS * CPR;
  const int wave_base = tid & ~63;
#pragma unroll
  for (int rd = 0; rd < (NPIECE + CVH_BLOCK - 1) / CVH_BLOCK; ++rd) {
    const int k = rd * CVH_BLOCK + tid;
    if (k < NPIECE) {
      const int r = k / CPR, cc = k - r * CPR;
      const int gi = clampi(i0 - 2 + r, 0, h - 1), gj = clampi(j0 - 2 + 2 * cc, 0, w - 2);
      const double *g = u_in + ((size_t)gi * w + gj);
      double *l = su + 2 * (rd * CVH_BLOCK + wave_base);  // wave-uniform base; HW adds lane*16
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)g,
          (__attribute__((address_space(3))) void *)(unsigned int)(uintptr_t)l, 16, 0, 0);
    }
  }
}

// Generic path (any width): all 8-byte loads issued before the first LDS write.
template <int ROWS>
__device__ __forceinline__ void stage_tile_regs(const double *u_in, double *su, int i0, int j0, int h,
                                                int w, int tid)
{
  constexpr int N = ROWS * PITCH;
  constexpr int NRD = (N + CVH_BLOCK - 1) / CVH_BLOCK;
  double v[NRD];
#pragma unroll
  for (int rd = 0; rd < NRD; ++rd) {
    const int idx = rd * CVH_BLOCK + tid;
    const int q = idx < N ? idx : N - 1;
    const int r = q / PITCH, c = q - r * PITCH;
    const int gi = clampi(i0 - 2 + r, 0, h - 1), gj = clampi(j0 - 2 + c, 0, w - 1);
    v[rd] = u_in[(size_t)gi * w + gj];
  }
#pragma unroll
  for (int rd = 0; rd < NRD; ++rd) {
    const int idx = rd * CVH_BLOCK + tid;
    if (idx < N) su[idx] = v[rd];
  }
}

// LDS layout of the step kernel (dynamic, one array, every carve offset a multiple of 16).
template <int C, int R, bool FAST, bool LUT>
struct StepSmem {
  static constexpr int NS = cvh_nsums(C);
  static constexpr int off_u = 0;                                        // (R+3) x PITCH doubles
  static constexpr int off_red = off_u + (R + 3) * PITCH;                // 4*NS
  static constexpr int off_fin = off_red + 4 * NS + (4 * NS) % 2;        // NS
  static constexpr int off_atan = off_fin + NS + NS % 2;                 // FAST: 2*CVH_ATAN_N
  static constexpr int off_lut = off_atan + (FAST ? 2 * CVH_ATAN_N : 0); // LUT: C*256
  static constexpr int off_flag = off_lut + (LUT ? C * 256 : 0);         // 2 doubles (s_last)
  static constexpr int doubles = off_flag + 2;
  static constexpr size_t bytes = (size_t)doubles * sizeof(double);
};

template <int C, int R, bool FAST, bool LUT, bool DMA>
__global__ __launch_bounds__(CVH_BLOCK) void csv_step_kernel(const CvhStepArgs a)
{
  using L = StepSmem<C, R, FAST, LUT>;
  constexpr int NS = cvh_nsums(C);
  constexpr int ROWS = R + 3;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *su = smem + L::off_u;
  double *sred = smem + L::off_red;
  double *sfin = smem + L::off_fin;
  double *satan = smem + L::off_atan;
  double *slut = smem + L::off_lut;
  int *s_last = (int *)(smem + L::off_flag);

  if (a.st->stopped) return;  // sticky stop: src/main.cpp:1000

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = a.h, w = a.w;
  const int bx = blockIdx.x % a.tiles_x, by = blockIdx.x / a.tiles_x;
  const int i0 = by * R, j0 = bx * TW;
  if (tid == 0) *s_last = 0;

  // ---- stage the tile: rows i0-2 .. i0+R, columns j0-2 .. j0+TW+1, replicate by clamping
  if (DMA) stage_tile_dma<ROWS>(a.u_in, su, i0, j0, h, w, tid);
  else stage_tile_regs<ROWS>(a.u_in, su, i0, j0, h, w, tid);

  // ---- this thread's image samples for all R rows (addresses clamped, values of
  // out-of-image pixels are never used)
  const int gj = j0 + tid;
  const int gjc = gj < w ? gj : w - 1;
  unsigned char ib[C][R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int gic = (i0 + r) < h ? (i0 + r) : h - 1;
#pragma unroll
    for (int k = 0; k < C; ++k) ib[k][r] = a.img[k][(size_t)gic * w + gjc];
  }

  double c1[C], c2[C], l1[C], l2[C];
#pragma unroll
  for (int k = 0; k < C; ++k) { c1[k] = a.st->c1[k]; c2[k] = a.st->c2[k]; l1[k] = a.lambda1[k]; l2[k] = a.lambda2[k]; }
  const double eps = a.eps;
  const double eps2 = eps * eps;

  if (FAST) {
    static_assert(2 * CVH_ATAN_N <= 2 * CVH_BLOCK, "atan table copy assumes two loads per thread");
    const double t0 = a.atan_tab[tid];
    const double t1 = a.atan_tab[tid + CVH_BLOCK < 2 * CVH_ATAN_N ? tid + CVH_BLOCK : 0];
    satan[tid] = t0;
    if (tid + CVH_BLOCK < 2 * CVH_ATAN_N) satan[tid + CVH_BLOCK] = t1;
  }
  if (LUT) {
    // region term per 8-bit value: dt/C * (lambda2 (v-c2)^2 - lambda1 (v-c1)^2) [+ (-nu dt)]
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const double v = (double)tid;
      const double d1 = v - c1[k], d2 = v - c2[k];
      const double reg = (d2 * d2) * l2[k] - (d1 * d1) * l1[k];
      slut[k * 256 + tid] = (k == 0) ? __builtin_fma(reg, a.beta, a.gamma) : reg * a.beta;
    }
  }
  if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- normalised x-gradient of this wave's left-edge column, lanes <-> rows
  double nx_edge;
  {
    const int er = lane < R ? lane : R - 1;
    const double *p = &su[(er + 2) * PITCH + wave * 64 + 1];
    nx_edge = normalised<FAST>(p[1] - p[0], central(p[-1], p[1]));
  }

  const int c = tid + 2;
  double um = su[1 * PITCH + c], u0 = su[2 * PITCH + c];
  double ny_prev = normalised<FAST>(u0 - um, central(su[c], u0));  // ny at row i0-1

  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0;

#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int gi = i0 + r;
    const double up = su[(r + 3) * PITCH + c];
    // BORDER_REPLICATE in x at the image edge (the DMA path cannot clamp single columns)
    const double uw = (gj == 0) ? u0 : su[(r + 2) * PITCH + c - 1];
    const double ue = (gj >= w - 1) ? u0 : su[(r + 2) * PITCH + c + 1];
    const double nx = normalised<FAST>(ue - u0, central(uw, ue));  // :365-366
    const double ny = normalised<FAST>(up - u0, central(um, up));  // :367-368
    const double nxl = from_left_lane(nx, read_lane(nx_edge, r));
    // backward differences with BORDER_REPLICATE on nx / ny (:371-372): zero on column 0 / row 0
    const double kx = (gj == 0) ? 0.0 : nx - nxl;
    const double ky = (gi == 0) ? 0.0 : ny - ny_prev;
    const double kappa = kx + ky;  // :373

    const bool valid = (gi < h) && (gj < w);
    int Iv[C];
    double Ik[C];
#pragma unroll
    for (int k = 0; k < C; ++k) { Iv[k] = ib[k][r]; Ik[k] = (double)Iv[k]; }

    double ud, hv;
    if (FAST) {
      double reg;
      if (LUT) {
        reg = slut[Iv[0]];
#pragma unroll
        for (int k = 1; k < C; ++k) reg += slut[k * 256 + Iv[k]];
      } else {
        reg = 0.0;
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const double d1 = Ik[k] - c1[k], d2 = Ik[k] - c2[k];
          reg += (d2 * d2) * l2[k] - (d1 * d1) * l1[k];
        }
        reg = __builtin_fma(reg, a.beta, a.gamma);
      }
      ud = __builtin_fma(kappa, a.alpha, reg);                                   // :985
      const double q = __builtin_fma(u0 * u0, a.dk1, a.dk2);                     // 1/delta_eps(u)
      const double r0 = __builtin_amdgcn_rcp(q);
      const double e = __builtin_fma(-q, r0, 1.0);
      ud = ud * __builtin_fma(__builtin_fma(e, e, e), r0, r0);                   // :992
    } else {
      ud = 0.0;  // :965
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const double d1 = Ik[k] - c1[k], d2 = Ik[k] - c2[k];
        const double vin = (d1 * d1) * l1[k];   // variance_penalty, :307-310
        const double vout = (d2 * d2) * l2[k];
        ud += vout - vin;                        // :979
      }
      ud = kappa * a.alpha + ud * a.beta + a.gamma;        // :985
      ud = ud * (eps / (kPi * (eps2 + u0 * u0)));           // :209, :992
    }
    const double un = u0 + ud;                              // :994
    if (FAST)
      hv = __builtin_fma(atan_table(un * a.inv_eps, satan), 1.0 / kPi, 0.5);
    else
      hv = heaviside_strict(un, eps);
    if (valid) a.u_out[(size_t)gi * w + gj] = un;
    // out-of-image lanes add exact zeros
    const double hz = valid ? hv : 0.0;
    const double udz = valid ? ud : 0.0;
    acc[0] += hz;
    if (!FAST) acc[1] += valid ? 1 - hv : 0.0;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      if (FAST) {
        acc[2 + k] = __builtin_fma(Ik[k], hz, acc[2 + k]);
      } else {
        acc[2 + k] += Ik[k] * hz;          // :276
        acc[2 + C + k] += valid ? Ik[k] * (1 - hv) : 0.0;
      }
    }
    if (FAST) acc[2 + 2 * C] = __builtin_fma(udz, udz, acc[2 + 2 * C]);
    else acc[2 + 2 * C] += udz * udz;      // :993
    um = u0; u0 = up; ny_prev = ny;
  }

  const double total = block_reduce<NS>(acc, sred);
  publish_partials_and_maybe_finalize<C>(a, total, sred, sfin, s_last, gridDim.x);
}

// Sums of H(u), (1-H(u)), I H, I (1-H) for the initial level set (seeds c1/c2 of step 1).
template <int C>
__global__ __launch_bounds__(CVH_BLOCK) void csv_init_sums_kernel(const CvhStepArgs a)
{
  constexpr int NS = cvh_nsums(C);
  __shared__ double sred[4 * NS];
  const size_t n = (size_t)a.h * a.w;
  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0;
  for (size_t q = (size_t)blockIdx.x * CVH_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * CVH_BLOCK) {
    const double hv = heaviside_strict(a.u_in[q], a.eps);
    const double omh = 1 - hv;
    acc[0] += hv;
    acc[1] += omh;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const double I = (double)a.img[k][q];
      acc[2 + k] += I * hv;
      acc[2 + C + k] += I * omh;
    }
  }
  const double total = block_reduce<NS>(acc, sred);
  if (threadIdx.x < NS) a.partials[(size_t)blockIdx.x * NS + threadIdx.x] = total;
}

template <int C>
__global__ __launch_bounds__(CVH_BLOCK) void csv_finalize_kernel(const CvhStepArgs a, int is_init)
{
  constexpr int NS = cvh_nsums(C);
  __shared__ double sred[4 * NS];
  __shared__ double sfin[NS];
  if (!is_init && a.st->stopped) return;
  finalize<C>(a, is_init, sred, sfin);
}

template <int C, int R, bool FAST, bool LUT>
hipError_t launch_step_v(const CvhStepArgs &a, hipStream_t s)
{
  using L = StepSmem<C, R, FAST, LUT>;
  static_assert(L::bytes <= 64 * 1024, "dynamic LDS above 64 KiB would need hipFuncSetAttribute");
  const int grid = a.tiles_x * a.tiles_y;
  // the LDS-DMA loader needs 16-byte aligned rows: even width (hipMalloc bases are aligned)
  if (a.use_dma && (a.w % 2 == 0) && a.w >= 2)
    CVH_LAUNCH((csv_step_kernel<C, R, FAST, LUT, true>), grid, L::bytes, s, a, "csv_step_kernel<%d, %d, %s, %s, true>", C, R, CVH_TF(FAST), CVH_TF(LUT));
  else
    CVH_LAUNCH((csv_step_kernel<C, R, FAST, LUT, false>), grid, L::bytes, s, a, "csv_step_kernel<%d, %d, %s, %s, false>", C, R, CVH_TF(FAST), CVH_TF(LUT));
  return hipGetLastError();
}

template <int C>
hipError_t launch_step_c(const CvhStepArgs &a, int fast, hipStream_t s)
{
  if (!fast) return a.tile_rows == 14 ? launch_step_v<C, 14, false, false>(a, s) : launch_step_v<C, 16, false, false>(a, s);
  if (a.tile_rows == 14)
    return a.use_lut ? launch_step_v<C, 14, true, true>(a, s) : launch_step_v<C, 14, true, false>(a, s);
  return a.use_lut ? launch_step_v<C, 16, true, true>(a, s) : launch_step_v<C, 16, true, false>(a, s);
}

}  // namespace

void cvh_step_grid(int h, int w, int tile_rows, int *tiles_x, int *tiles_y)
{
  *tiles_x = (w + TW - 1) / TW;
  *tiles_y = (h + tile_rows - 1) / tile_rows;
}

int cvh_step_max_blocks(int h, int w)
{
  int tx, ty;
  cvh_step_grid(h, w, 12, &tx, &ty);  // smallest rows-per-workgroup of any step kernel
  return tx * ty;
}

hipError_t cvh_launch_step(const CvhStepArgs &a, int channels, int fast, hipStream_t s)
{
  return channels == 1 ? launch_step_c<1>(a, fast, s) : launch_step_c<3>(a, fast, s);
}

int cvh_init_sum_blocks(int h, int w)
{
  const size_t n = (size_t)h * w;
  size_t b = (n + CVH_BLOCK * 8 - 1) / (CVH_BLOCK * 8);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

hipError_t cvh_launch_init_sums(const CvhStepArgs &a, int channels, int, int *nparts_out,
                                hipStream_t s)
{
  const int nb = cvh_init_sum_blocks(a.h, a.w);
  *nparts_out = nb;
  if (channels == 1)
    hipLaunchKernelGGL(csv_init_sums_kernel<1>, dim3(nb), dim3(CVH_BLOCK), 0, s, a);
  else
    hipLaunchKernelGGL(csv_init_sums_kernel<3>, dim3(nb), dim3(CVH_BLOCK), 0, s, a);
  return hipGetLastError();
}

hipError_t cvh_launch_finalize(const CvhStepArgs &a, int channels, int is_init, hipStream_t s)
{
  if (channels == 1)
    hipLaunchKernelGGL(csv_finalize_kernel<1>, dim3(1), dim3(CVH_BLOCK), 0, s, a, is_init);
  else
    hipLaunchKernelGGL(csv_finalize_kernel<3>, dim3(1), dim3(CVH_BLOCK), 0, s, a, is_init);
  return hipGetLastError();
}
