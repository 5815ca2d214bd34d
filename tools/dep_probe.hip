// dep_probe.hip — latency of DEPENDENT FP64 chains and how many waves per SIMD it takes to
// saturate the FP64 pipe with them (diagnostic for the CSV kernel's issue model).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NCH, int MIX>
__global__ __launch_bounds__(256) void probe(double *out, int iters, double a, double b)
{
  double x[NCH];
  for (int k = 0; k < NCH; ++k) x[k] = 1.0 + threadIdx.x * 1e-3 + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8 / NCH; ++r)
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (MIX == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
        if (MIX == 1) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b)); }
      }
    if (MIX == 1) {  // one transcendental per 8 fma, dependent
#pragma unroll
      for (int k = 0; k < NCH; ++k) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[k]));
    }
  }
  double s = 0;
  for (int k = 0; k < NCH; ++k) s += x[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH, int MIX>
void run(int bpc, int cus, int iters, double *d_out)
{
  const int grid = bpc * cus;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<NCH, MIX>), dim3(grid), dim3(256), 0, 0, d_out, iters / 10, 1.0000001, 1e-9);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<NCH, MIX>), dim3(grid), dim3(256), 0, 0, d_out, iters, 1.0000001, 1e-9);
  CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_simd = (double)iters * (8 + (MIX ? NCH : 0)) * bpc;
  printf("chains %d mix %d waves/SIMD %d: %.3f ms  => %.2f ns per wave-instr per SIMD (x2.1GHz = %.1f cycles)\n", NCH, MIX, bpc, ms,
         ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.1);
}

int main()
{
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  double *d_out; CHK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8));
  const int iters = 20000;
  for (int bpc : {1, 2, 4, 6, 8}) {
    run<1, 0>(bpc, cus, iters, d_out);
    run<2, 0>(bpc, cus, iters, d_out);
    run<8, 0>(bpc, cus, iters, d_out);
    run<1, 1>(bpc, cus, iters, d_out);
  }
  return 0;
}
