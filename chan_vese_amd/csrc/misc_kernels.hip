// misc_kernels.hip — mask, selection compositing and the ParallelPixelFunction map (gfx950).
#include "cvh_internal.h"

namespace {

constexpr double kPi = 3.14159265358979323846;

inline int flat_grid(size_t n)
{
  size_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// mask = ((float)u > 0), src/main.cpp:397-400
__global__ void mask_kernel(const double *u, uint8_t *mask, size_t n, int invert)
{
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    const uint8_t m = ((float)u[q] > 0.0f) ? 1 : 0;
    mask[q] = invert ? (uint8_t)(1 - m) : m;
  }
}

// separate(), src/main.cpp:402-403
__global__ void separate_kernel(const uint8_t *img3, const double *u, uint8_t *sel3, size_t n, int invert)
{
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    bool m = (float)u[q] > 0.0f;
    if (invert) m = !m;
    for (int c = 0; c < 3; ++c) sel3[q * 3 + c] = m ? img3[q * 3 + c] : (uint8_t)255;
  }
}

// ParallelPixelFunction::operator(), src/ParallelPixelFunction.cpp:15-16, with the
// function chosen by tag (regularized_delta / regularized_heaviside / 1 - heaviside).
__global__ void ppf_kernel(double *data, size_t n, int op, double eps)
{
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    const double x = data[q];
    double y;
    if (op == CVH_OP_DELTA) {
      y = eps / (kPi * (eps * eps + x * x));            // src/main.cpp:209
    } else {
      y = (1 + 2 / kPi * atan(x / eps)) / 2;            // src/main.cpp:193
      if (op == CVH_OP_ONE_MINUS_HEAVISIDE) y = 1 - y;  // src/main.cpp:267
    }
    data[q] = y;
  }
}

// levelset_checkerboard, src/main.cpp:226-231: sign(sin(pi i/5) * sin(pi j/5)).  The two sine vectors come from the host's
// libm (sv = [h row factors | w column factors]); the product is ONE IEEE multiplication, so the device reproduces
// cvh_levelset_checkerboard_host bit for bit without 8 bytes per pixel crossing PCIe.
__global__ void checkerboard_kernel(const double *sv, double *u, int h, int w)
{
  for (int i = (int)blockIdx.x; i < h; i += (int)gridDim.x) {
    const double si = sv[i];
    for (int j = (int)threadIdx.x; j < w; j += (int)blockDim.x) {
      const double z = si * sv[h + j];
      u[(size_t)i * w + j] = (z == 0) ? 0.0 : (z < 0 ? -1.0 : 1.0);
    }
  }
}

// Per-plane sum(p) and sum(p^2) as exact 64-bit integers (out[2k], out[2k+1]; zeroed by the caller): the region means'
// sum(I) and, for one channel, the tol-free stop norm of src/main.cpp:950-959 (every partial sum of the reference's loop
// is an integer below 2^53 there, so its result does not depend on the order).
__global__ void image_sums_kernel(const uint8_t *p0, const uint8_t *p1, const uint8_t *p2, int C, size_t n,
                                  unsigned long long *out)
{
  __shared__ unsigned long long sh[4][6];
  const uint8_t *pl[3] = {p0, p1, p2};
  unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
  const size_t pieces = n / 16, stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < C; ++k) {
    unsigned s1 = 0, s2 = 0;   // flushed every 256 pieces: 256 * 16 * 65025 < 2^32
    int pending = 0;
    for (size_t q = t0; q < pieces; q += stride) {
      const uint4 v = reinterpret_cast<const uint4 *>(pl[k])[q];
      const unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int b = 0; b < 4; ++b) { const unsigned x = (wds[i] >> (8 * b)) & 0xffu; s1 += x; s2 += x * x; }
      }
      if (++pending == 256) { acc[2 * k] += s1; acc[2 * k + 1] += s2; s1 = s2 = 0; pending = 0; }
    }
    for (size_t q = pieces * 16 + t0; q < n; q += stride) { const unsigned x = pl[k][q]; s1 += x; s2 += x * x; }
    acc[2 * k] += s1; acc[2 * k + 1] += s2;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    unsigned long long v = acc[s];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) sh[wave][s] = v;
  }
  __syncthreads();
  if (threadIdx.x < 2 * C) atomicAdd(&out[threadIdx.x], sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

}  // namespace

hipError_t cvh_launch_checkerboard(const double *sv, double *u, int h, int w, hipStream_t s)
{
  hipLaunchKernelGGL(checkerboard_kernel, dim3(h < 2048 ? h : 2048), dim3(256), 0, s, sv, u, h, w);
  return hipGetLastError();
}

hipError_t cvh_launch_image_sums(const uint8_t *const *planes, int channels, size_t n, unsigned long long *out, hipStream_t s)
{
  hipLaunchKernelGGL(image_sums_kernel, dim3(flat_grid(n / 16 + 1) > 1024 ? 1024 : flat_grid(n / 16 + 1)), dim3(256), 0, s,
                     planes[0], channels > 1 ? planes[1] : nullptr, channels > 2 ? planes[2] : nullptr, channels, n, out);
  return hipGetLastError();
}

// Contour map of one video frame, src/VideoWriterManager.cpp:60-74: mask = (uint8(round(u)) > 0)
// (convertTo(CV_8U) rounds half to even and saturates, so mask = rint(u) >= 1), contours by
// cv::findContours(RETR_TREE, CHAIN_APPROX_SIMPLE) drawn 1 pixel wide with 8-connected lines.
// Restated (OpenCV is absent; its 2.4 behaviour as recalled: the outermost pixel ring of the mask is
// cleared before tracing; a border point of an 8-connected component is a 1-pixel with a 0-pixel
// in its 4-neighbourhood; hole borders are pixels of the surrounding component; the polygonal
// approximation re-drawn with 8-connected lines covers the traced pixels again): contour(i,j) = 1 where
// the cleared mask is 1 and one of its 4 neighbours is 0.
__device__ __forceinline__ int video_mask_at(const double *u, int h, int w, int i, int j)
{
  if (i <= 0 || j <= 0 || i >= h - 1 || j >= w - 1) return 0;
  return __builtin_rint(u[(size_t)i * w + j]) >= 1.0;
}

__global__ void contour_kernel(const double *u, uint8_t *out, int h, int w)
{
  const size_t n = (size_t)h * w;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(q / w), j = (int)(q % w);
    uint8_t c = 0;
    if (video_mask_at(u, h, w, i, j))
      c = !(video_mask_at(u, h, w, i - 1, j) && video_mask_at(u, h, w, i + 1, j) &&
            video_mask_at(u, h, w, i, j - 1) && video_mask_at(u, h, w, i, j + 1));
    out[q] = c;
  }
}

hipError_t cvh_launch_contour(const double *u, uint8_t *out, int h, int w, hipStream_t s)
{
  hipLaunchKernelGGL(contour_kernel, dim3(flat_grid((size_t)h * w)), dim3(256), 0, s, u, out, h, w);
  return hipGetLastError();
}

// FP32-state mode (option "state" = 32): the float buffers take over a level set (rounded to nearest even; the double copy is rewritten
// with the rounded values), and the double mirror is refreshed from them for get / mask / contour / selection / the initial sums.
__global__ void state_narrow_kernel(double *u, float *uf, size_t n)
{
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    const float f = (float)u[q];
    uf[q] = f;
    u[q] = (double)f;
  }
}
__global__ void state_widen_kernel(const float *uf, double *u, size_t n)
{
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) u[q] = (double)uf[q];
}
hipError_t cvh_launch_state_narrow(double *u, float *uf, size_t n, hipStream_t s)
{
  hipLaunchKernelGGL(state_narrow_kernel, dim3((unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384)), dim3(256), 0, s, u, uf, n);
  return hipGetLastError();
}
hipError_t cvh_launch_state_widen(const float *uf, double *u, size_t n, hipStream_t s)
{
  hipLaunchKernelGGL(state_widen_kernel, dim3((unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384)), dim3(256), 0, s, uf, u, n);
  return hipGetLastError();
}

hipError_t cvh_launch_mask(const double *u, uint8_t *mask, size_t n, int invert, hipStream_t s)
{
  hipLaunchKernelGGL(mask_kernel, dim3(flat_grid(n)), dim3(256), 0, s, u, mask, n, invert);
  return hipGetLastError();
}

hipError_t cvh_launch_separate(const uint8_t *img3, const double *u, uint8_t *sel3, size_t n,
                               int invert, hipStream_t s)
{
  hipLaunchKernelGGL(separate_kernel, dim3(flat_grid(n)), dim3(256), 0, s, img3, u, sel3, n, invert);
  return hipGetLastError();
}

hipError_t cvh_launch_ppf(double *data, size_t n, int op, double eps, hipStream_t s)
{
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(ppf_kernel, dim3(flat_grid(n)), dim3(256), 0, s, data, n, op, eps);
  return hipGetLastError();
}
