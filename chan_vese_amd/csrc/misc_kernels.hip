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
