// math_probe.hip — cost of the CSV kernel's per-pixel FP64 building blocks on MI355X, isolated from
// memory: ns per call per SIMD at a given occupancy, with 1 or 4 independent streams per thread.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../chan_vese_amd/csrc/csv_device.h"
using namespace cvh_dev;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double heaviside_fast(double x, const double *tab)
{
  const double a = fmin(fabs(x), 1e300);
  const double n = a - 1.0, d = a + 1.0;
  const double y0 = n * __builtin_amdgcn_rcp(d);
  const double fi = __builtin_rint(y0 * 128.0);
  const double c = fi * (1.0 / 128.0);
  const int j = (int)fi;
  const double num = __builtin_fma(-c, d, n);
  const double den = __builtin_fma(c, n, d);
  const double r0 = __builtin_amdgcn_rcp(den);
  const double r = __builtin_fma(__builtin_fma(-den, r0, 1.0), r0, r0);
  const double z = num * r;
  const double z2 = z * z;
  const double p = __builtin_fma(z2, 0.2, -1.0 / 3.0);
  const double az = __builtin_fma(z * z2, p, z);
  const double atpi = __builtin_fma(az, 1.0 / kPi, tab[j + 128]);
  return 0.5 + __builtin_copysign(atpi, x);
}

// MODE 0: two normalised gradients; 1: heaviside_fast; 2: delta reciprocal; 3: all three (a "row")
template <int MODE, int NS>
__global__ __launch_bounds__(256) void probe(double *out, const double *tabg, int iters, double k1, double k2)
{
  __shared__ double tab[258];
  for (int q = threadIdx.x; q < 257; q += 256) tab[q] = tabg[q];
  __syncthreads();
  double x[NS], acc = 0;
  for (int s = 0; s < NS; ++s) x[s] = 0.3 + threadIdx.x * 0.01 + s;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double v = x[s];
      if (MODE == 0 || MODE == 3) {
        const double nx = normalised<true>(v - 0.25, 0.5 * (v + 0.5));
        const double ny = normalised<true>(v + 0.125, 0.5 * (v - 0.75));
        v = v + (nx - ny) * 1e-3;
      }
      if (MODE == 2 || MODE == 3) {
        const double qd = __builtin_fma(v, v, k2) * k1;
        const double r0 = __builtin_amdgcn_rcp(qd);
        const double e = __builtin_fma(-qd, r0, 1.0);
        v = v + 1e-3 * __builtin_fma(__builtin_fma(e, e, e), r0, r0);
      }
      if (MODE == 1 || MODE == 3) {
        const double hv = heaviside_fast(v, tab);
        acc += hv;
        v = v + 1e-4 * hv;
      }
      x[s] = v;
    }
  }
  for (int s = 0; s < NS; ++s) acc += x[s];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int NS>
void run(const char *name, int bpc, int cus, double *d_out, const double *d_tab)
{
  const int iters = 4000, grid = bpc * cus;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<MODE, NS>), dim3(grid), dim3(256), 0, 0, d_out, d_tab, iters / 10, 3.14, 1.0);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<MODE, NS>), dim3(grid), dim3(256), 0, 0, d_out, d_tab, iters, 3.14, 1.0);
  CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const double calls_per_simd = (double)iters * NS * bpc;
  printf("%-28s streams %d waves/SIMD %d: %7.1f ns per wave-call per SIMD\n", name, NS, bpc, ms * 1e6 / calls_per_simd);
}

int main()
{
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  double *d_out, *d_tab; CHK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8)); CHK(hipMalloc(&d_tab, 258 * 8));
  double tab[258];
  for (int j = 0; j < 257; ++j) tab[j] = (0.78539816339744830962 + atan((j - 128) / 128.0)) / 3.14159265358979323846;
  CHK(hipMemcpy(d_tab, tab, 257 * 8, hipMemcpyHostToDevice));
  for (int bpc : {1, 4, 6}) {
    run<0, 1>("2 x normalised (24 DP + 2 rsq)", bpc, cus, d_out, d_tab);
    run<0, 4>("2 x normalised (24 DP + 2 rsq)", bpc, cus, d_out, d_tab);
    run<1, 1>("heaviside_fast (21 DP + 2 rcp)", bpc, cus, d_out, d_tab);
    run<1, 4>("heaviside_fast (21 DP + 2 rcp)", bpc, cus, d_out, d_tab);
    run<2, 1>("delta rcp (6 DP + 1 rcp)", bpc, cus, d_out, d_tab);
    run<3, 1>("row: all (~55 DP + 5 trans)", bpc, cus, d_out, d_tab);
    run<3, 4>("row: all (~55 DP + 5 trans)", bpc, cus, d_out, d_tab);
  }
  return 0;
}
