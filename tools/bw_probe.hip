// bw_probe.hip — achievable HBM bandwidth on MI355X for 8- vs 16-byte-per-lane streaming
// (diagnostic; informs the CSV kernel's load/store width).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double double2_t __attribute__((ext_vector_type(2)));

template <typename T, int MODE>  // MODE 0 copy, 1 read-only, 2 write-only
__global__ __launch_bounds__(256) void k(const T *in, T *out, size_t n, double *sink)
{
  T acc = T(0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (MODE == 0) out[i] = in[i];
    if (MODE == 1) acc += in[i];
    if (MODE == 2) out[i] = T(1.5);
  }
  if (MODE == 1) { double s = sizeof(T) == 8 ? ((double *)&acc)[0] : ((double *)&acc)[0] + ((double *)&acc)[1]; if (s == 12345.678) sink[0] = s; }
}

// row-strided: each wave streams its own 512-byte-wide column strip down many rows (the CSV wave kernel's pattern)
__global__ __launch_bounds__(256) void strip8(const double *in, double *out, int h, int w, int rows_per_wave)
{
  const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwc = w / 64, wc = gw % nwc, ws = gw / nwc;
  const int r0 = ws * rows_per_wave;
  for (int r = r0; r < r0 + rows_per_wave && r < h; ++r) out[(size_t)r * w + wc * 64 + lane] = in[(size_t)r * w + wc * 64 + lane];
}

// the CSV wave kernel's exact access pattern, no arithmetic: 63 output columns per wave (lane 0 =
// halo), 8-byte loads/stores, one image byte per lane, rows_per_wave rows, 3 halo rows
template <int WC, int USEIMG, int HALO>
__global__ __launch_bounds__(256) void strip63(const double *in, double *out, const unsigned char *img, int h, int w, int rows_per_wave, int nwc)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbc = (nwc + 3) >> 2;
  const int wc = (blockIdx.x % nbc) * 4 + wave, ws = blockIdx.x / nbc;
  if (wc >= nwc) return;
  int col = WC * wc - (WC == 63 ? 1 : 0) + lane; const bool valid = (WC == 64 || lane >= 1) && col < w; col = col < 0 ? 0 : (col >= w ? w - 1 : col);
  const int r0 = ws * rows_per_wave;
  double acc = 0;
  if (HALO) for (int r = r0 - 2; r < r0; ++r) acc += in[(size_t)(r < 0 ? 0 : r) * w + col];
  for (int r = r0; r < r0 + rows_per_wave && r < h; ++r) {
    const double v = in[(size_t)(HALO ? (r + 1 < h ? r + 1 : h - 1) : r) * w + col];
    const double b = USEIMG ? (double)img[(size_t)r * w + col] : 1.0;
    if (valid) out[(size_t)r * w + col] = v + b + acc;
  }
}

template <typename F> float timeit(F f)
{
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  f(); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a)); for (int i = 0; i < 10; ++i) f(); CHK(hipEventRecord(b)); CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, a, b)); return ms / 10;
}

int main()
{
  const size_t bytes = (size_t)512 << 20;  // 512 MiB per buffer: beyond the 256 MiB Infinity Cache
  double *a, *b, *sink; CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMalloc(&sink, 64));
  CHK(hipMemset(a, 0, bytes)); CHK(hipMemset(b, 0, bytes));
  for (int grid : {2048, 8192}) {
    float t;
    t = timeit([&] { hipLaunchKernelGGL((k<double, 0>), dim3(grid), dim3(256), 0, 0, a, b, bytes / 8, sink); });
    printf("grid %5d copy  8B/lane: %.1f us  %.2f TB/s (r+w)\n", grid, t * 1e3, 2 * bytes / (t * 1e-3) / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k<double2_t, 0>), dim3(grid), dim3(256), 0, 0, (double2_t *)a, (double2_t *)b, bytes / 16, sink); });
    printf("grid %5d copy 16B/lane: %.1f us  %.2f TB/s (r+w)\n", grid, t * 1e3, 2 * bytes / (t * 1e-3) / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k<double, 1>), dim3(grid), dim3(256), 0, 0, a, b, bytes / 8, sink); });
    printf("grid %5d read  8B/lane: %.1f us  %.2f TB/s\n", grid, t * 1e3, bytes / (t * 1e-3) / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k<double2_t, 1>), dim3(grid), dim3(256), 0, 0, (double2_t *)a, (double2_t *)b, bytes / 16, sink); });
    printf("grid %5d read 16B/lane: %.1f us  %.2f TB/s\n", grid, t * 1e3, bytes / (t * 1e-3) / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k<double, 2>), dim3(grid), dim3(256), 0, 0, a, b, bytes / 8, sink); });
    printf("grid %5d write 8B/lane: %.1f us  %.2f TB/s\n", grid, t * 1e3, bytes / (t * 1e-3) / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k<double2_t, 2>), dim3(grid), dim3(256), 0, 0, (double2_t *)a, (double2_t *)b, bytes / 16, sink); });
    printf("grid %5d write16B/lane: %.1f us  %.2f TB/s\n", grid, t * 1e3, bytes / (t * 1e-3) / 1e12);
  }
  // 4096 x 4096 doubles (128 MiB in, 128 MiB out): the CSV footprint, strip pattern
  const int h = 4096, w = 4096;
  for (int rpw : {32, 45, 64, 128}) {
    const int waves = (w / 64) * ((h + rpw - 1) / rpw);
    float t = timeit([&] { hipLaunchKernelGGL(strip8, dim3((waves + 3) / 4), dim3(256), 0, 0, a, b, h, w, rpw); });
    printf("strip copy 4096^2 rows/wave %3d (%d waves): %.1f us  %.2f TB/s (r+w)\n", rpw, waves, t * 1e3, 2.0 * h * w * 8 / (t * 1e-3) / 1e12);
  }
  {
    unsigned char *img; CHK(hipMalloc(&img, (size_t)h * w)); CHK(hipMemset(img, 1, (size_t)h * w));
    const int rpw = 45;
    auto runv = [&](const char *name, auto kern, int wcw) {
      const int nwc = (w + wcw - 1) / wcw;
      const int nstr = (h + rpw - 1) / rpw, blocks = ((nwc + 3) / 4) * nstr;
      int flip = 0;
      float t = timeit([&] { if (flip ^= 1) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, a, b, img, h, w, rpw, nwc); else hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, b, a, img, h, w, rpw, nwc); });
      printf("%-46s: %.1f us  => %.2f TB/s algorithmic (17 B/px)\n", name, t * 1e3, 17.0 * h * w / (t * 1e-3) / 1e12);
    };
    runv("63-col strips, image bytes, halo rows (wave kernel)", strip63<63, 1, 1>, 63);
    runv("63-col strips, no image, halo rows", strip63<63, 0, 1>, 63);
    runv("63-col strips, image bytes, no halo rows", strip63<63, 1, 0>, 63);
    runv("64-col aligned strips, image bytes, halo rows", strip63<64, 1, 1>, 64);
    runv("64-col aligned strips, no image, halo rows", strip63<64, 0, 1>, 64);
    runv("64-col aligned strips, no image, no halo", strip63<64, 0, 0>, 64);
  }
  float t = timeit([&] { hipLaunchKernelGGL((k<double2_t, 0>), dim3(8192), dim3(256), 0, 0, (double2_t *)a, (double2_t *)b, (size_t)h * w / 2, sink); });
  printf("flat copy 4096^2 16B/lane: %.1f us  %.2f TB/s (r+w)\n", t * 1e3, 2.0 * h * w * 8 / (t * 1e-3) / 1e12);
  return 0;
}
