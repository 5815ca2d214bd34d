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

  // g on rows i0-1 .. i0+PTH, cols j0-1 .. j0+PTW (:513-522)
#pragma unroll 3
  for (int idx = tid; idx < (PTH + 2) * GP; idx += CVH_BLOCK) {
    const int r = idx / GP, c = idx - r * GP;
    const int gi = i0 - 1 + r, gj = j0 - 1 + c;
    double g = 1.0;  // image border ring (and anything clamped onto it)
    if (gi > 0 && gi < h - 1 && gj > 0 && gj < w - 1) {
      const double *p = &sI[(r + 1) * IP + (c + 1)];
      // cv::Sobel ksize 3: row pass then column pass (see oracle/cv_oracle.c)
      const double rm = p[-IP + 1] - p[-IP - 1], r0 = p[1] - p[-1], rp = p[IP + 1] - p[IP - 1];
      const double gx = rm + r0 * 2 + rp;
      const double sm = p[-IP - 1] + p[-IP] * 2 + p[-IP + 1];
      const double sp = p[IP - 1] + p[IP] * 2 + p[IP + 1];
      const double gy = sp - sm;
      if (FAST) g = rcp_refined(__builtin_fma(__builtin_fma(gx, gx, gy * gy), a.invK2, 1.0));
      else g = 1.0 / (1.0 + (gx * gx + gy * gy) / a.K2);
    }
    sg[idx] = g;
  }
  __syncthreads();

  const int tx = tid & 63, ty = tid >> 6;
  const int gj = j0 + tx;
#pragma unroll
  for (int q = 0; q < PTH / 4; ++q) {
    const int r = ty * (PTH / 4) + q;
    const int gi = i0 + r;
    if (gi < h && gj < w) {
      const double *p = &sI[(r + 2) * IP + (tx + 2)];
      const double *g = &sg[(r + 1) * GP + (tx + 1)];
      const double I0 = p[0], c0 = g[0];
      if (FAST) {
        double s = (g[GP] + c0) * (p[IP] - I0);
        s = __builtin_fma(g[1] + c0, p[1] - I0, s);
        s = __builtin_fma(g[-GP] + c0, p[-IP] - I0, s);
        s = __builtin_fma(g[-1] + c0, p[-1] - I0, s);
        a.out[(size_t)gi * w + gj] = __builtin_fma(a.L4, s, I0);
      } else {
        const double s = (g[GP] + c0) * (p[IP] - I0) + (g[1] + c0) * (p[1] - I0) +
                         (g[-GP] + c0) * (p[-IP] - I0) + (g[-1] + c0) * (p[-1] - I0);
        a.out[(size_t)gi * w + gj] = I0 + a.L * s / 4;  // :544-547
      }
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

hipError_t cvh_launch_pm_step(const CvhPmArgs &a, hipStream_t s)
{
  if (a.fast) hipLaunchKernelGGL(pm_step_kernel<true>, dim3(a.tiles_x * a.tiles_y), dim3(CVH_BLOCK), 0, s, a);
  else hipLaunchKernelGGL(pm_step_kernel<false>, dim3(a.tiles_x * a.tiles_y), dim3(CVH_BLOCK), 0, s, a);
  return hipGetLastError();
}

hipError_t cvh_launch_pm_store(const double *state, uint8_t *plane, size_t n, hipStream_t s)
{
  hipLaunchKernelGGL(pm_store_kernel, dim3(flat_grid(n)), dim3(256), 0, s, state, plane, n);
  return hipGetLastError();
}
