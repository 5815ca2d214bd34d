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

}  // namespace

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
