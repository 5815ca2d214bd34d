// diagnostic: does a 128 MiB FP64 buffer updated IN PLACE stay in the 256 MiB Infinity Cache between passes,
// while a ping-pong pair (256 MiB footprint) does not?  Same bytes per pass in both forms.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void pass_k(const double2 *__restrict__ in, double2 *out, const uint4 *img, size_t n2, double add)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n2; i += stride) {
    double2 v = in[i];
    if ((i & 7) == 0) { uint4 b = img[i >> 3]; v.x += (double)(b.x & 1) * 1e-30; }
    v.x += add; v.y += add;
    out[i] = v;
  }
}
int main()
{
  const size_t n = (size_t)4096 * 4096, n2 = n / 2;
  double2 *a, *b; uint4 *img;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&img, n);
  hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8); hipMemset(img, 0, n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 8, blk = 256, iters = 200;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < iters; ++it) {
        if (mode == 0) { if (it & 1) hipLaunchKernelGGL(pass_k, dim3(grid), dim3(blk), 0, 0, b, a, img, n2, 1.0); else hipLaunchKernelGGL(pass_k, dim3(grid), dim3(blk), 0, 0, a, b, img, n2, 1.0); }
        else if (mode == 1) hipLaunchKernelGGL(pass_k, dim3(grid), dim3(blk), 0, 0, a, a, img, n2, 1.0);
        else hipLaunchKernelGGL(pass_k, dim3(grid), dim3(blk), 0, 0, a, a, img, n2 / 2, 1.0);   // 64 MiB in place
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / iters, bytes = (mode == 2 ? 0.5 : 1.0) * (16.0 * n + n);
      printf("%s: %.2f us per pass, %.2f TB/s (read+write+image bytes)\n",
             mode == 0 ? "ping-pong 2 x 128 MiB" : mode == 1 ? "in place 128 MiB" : "in place 64 MiB", us, bytes / us / 1e6);
    }
  }
  return 0;
}
