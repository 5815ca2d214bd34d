// dp_rate_probe.hip — measures FP64 VALU issue cost on the GPU (cycles per wave-instruction per
// SIMD) and the clock held under that load.  Diagnostic only; not part of the product path.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

enum { OP_FMA64, OP_MUL64, OP_ADD64, OP_RCP64, OP_RSQ64, OP_FMA32, OP_MOV32, OP_CNDMASK, OP_CVT, OP_LDEXP, OP_RNDNE, OP_MIN64, OP_CND_SGPR, OP_CMP_CND, OP_CMP64, OP_READLANE, OP_DPP, OP_BFI, OP_XOR, OP_RCP32, OP_RSQ32, OP_CVT32_64, OP_CVT64_32, OP_MOV64, NOPS };
static const char *names[NOPS] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "v_fma_f32", "v_mov_b32", "v_cndmask", "v_cvt_f64_u32", "v_ldexp_f64", "v_rndne_f64", "v_min_f64", "v_cndmask(sgpr mask)", "v_cmp+v_cndmask", "v_cmp_gt_f64", "v_readlane", "v_mov_dpp", "v_bfi_b32", "v_xor_b32", "v_rcp_f32", "v_rsq_f32", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_mov_b64"};

template <int OP>
__global__ __launch_bounds__(256) void probe(double *out, unsigned long long *ticks, int iters, double a, double b)
{
  double x[8];
  float f[8];
  int n[8];
  for (int k = 0; k < 8; ++k) { x[k] = 1.0 + threadIdx.x * 1e-3 + k; f[k] = (float)x[k]; n[k] = threadIdx.x + k; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (OP == OP_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
      if (OP == OP_MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      if (OP == OP_ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(b));
      if (OP == OP_RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[k]));
      if (OP == OP_RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(x[k]));
      if (OP == OP_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"((float)a), "v"((float)b));
      if (OP == OP_MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(n[k]) : "v"(n[(k + 1) & 7]));
      if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
      if (OP == OP_CVT) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x[k]) : "v"(n[k]));
      if (OP == OP_LDEXP) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[k]) : "v"(n[k] & 1));
      if (OP == OP_RNDNE) asm volatile("v_rndne_f64 %0, %0" : "+v"(x[k]));
      if (OP == OP_MIN64) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      if (OP == OP_CND_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
      if (OP == OP_CMP_CND) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[k]) : "v"(n[(k + 1) & 7]) : "vcc");
      if (OP == OP_CMP64) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(x[k]), "v"(a) : "vcc");
      if (OP == OP_READLANE) { int t; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t) : "v"(n[k])); n[(k + 1) & 7] += 0; asm volatile("" :: "s"(t)); }
      if (OP == OP_DPP) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
      if (OP == OP_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(n[k]) : "v"(n[(k + 1) & 7]), "v"(n[(k + 2) & 7]));
      if (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
      if (OP == OP_RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
      if (OP == OP_RSQ32) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[k]));
      if (OP == OP_CVT32_64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(x[k]));
      if (OP == OP_CVT64_32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[k]) : "v"(f[k]));
      if (OP == OP_MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(x[k]) : "v"(x[(k + 1) & 7]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int k = 0; k < 8; ++k) s += x[k] + f[k] + n[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    ticks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
    ticks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
  }
}

template <int OP>
void run(int blocks_per_cu, int cus, int iters, double *d_out, unsigned long long *d_ticks)
{
  const int grid = blocks_per_cu * cus;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), 0, 0, d_out, d_ticks, iters / 10, 1.0000001, 1e-9);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), 0, 0, d_out, d_ticks, iters, 1.0000001, 1e-9);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> t((size_t)grid * 8);
  CHK(hipMemcpy(t.data(), d_ticks, t.size() * 8, hipMemcpyDeviceToHost));
  double st = 0, sr = 0;
  for (int i = 0; i < grid * 4; ++i) { st += (double)t[2 * i]; sr += (double)t[2 * i + 1]; }
  const double clk_mhz = st / sr * 100.0;                 // memtime ticks per 100 MHz realtime tick
  const double wave_cycles = st / (grid * 4);              // cycles one wave spent in the loop
  const double instr = (double)iters * 8;                  // per wave
  // waves per SIMD = blocks_per_cu (4 waves per block over 4 SIMDs)
  printf("%-14s waves/SIMD=%d  %.3f ms  clock %.0f MHz  cycles/instr/wave %.2f  => per-SIMD issue %.2f cycles/instr\n",
         names[OP], blocks_per_cu, ms, clk_mhz, wave_cycles / instr, wave_cycles / instr / blocks_per_cu);
}

int main()
{
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s CUs %d clockRate %d kHz\n", p.name, cus, p.clockRate);
  double *d_out; unsigned long long *d_ticks;
  CHK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8));  // up to 8 blocks per CU
  CHK(hipMalloc(&d_ticks, (size_t)cus * 8 * 8 * 8));
  const int iters = 20000;
  for (int bpc : {4, 8}) {
    run<OP_CND_SGPR>(bpc, cus, iters, d_out, d_ticks);
    run<OP_CMP_CND>(bpc, cus, iters, d_out, d_ticks);
    run<OP_CMP64>(bpc, cus, iters, d_out, d_ticks);
    run<OP_READLANE>(bpc, cus, iters, d_out, d_ticks);
    run<OP_DPP>(bpc, cus, iters, d_out, d_ticks);
    run<OP_BFI>(bpc, cus, iters, d_out, d_ticks);
    run<OP_XOR>(bpc, cus, iters, d_out, d_ticks);
  }
  for (int bpc : {3, 8}) {
    run<OP_FMA64>(bpc, cus, iters, d_out, d_ticks);
    run<OP_MUL64>(bpc, cus, iters, d_out, d_ticks);
    run<OP_ADD64>(bpc, cus, iters, d_out, d_ticks);
    run<OP_RCP64>(bpc, cus, iters, d_out, d_ticks);
    run<OP_RSQ64>(bpc, cus, iters, d_out, d_ticks);
    run<OP_FMA32>(bpc, cus, iters, d_out, d_ticks);
    run<OP_MOV32>(bpc, cus, iters, d_out, d_ticks);
    run<OP_CNDMASK>(bpc, cus, iters, d_out, d_ticks);
    run<OP_CVT>(bpc, cus, iters, d_out, d_ticks);
    run<OP_LDEXP>(bpc, cus, iters, d_out, d_ticks);
    run<OP_RNDNE>(bpc, cus, iters, d_out, d_ticks);
    run<OP_MIN64>(bpc, cus, iters, d_out, d_ticks);
    run<OP_RCP32>(bpc, cus, iters, d_out, d_ticks);
    run<OP_RSQ32>(bpc, cus, iters, d_out, d_ticks);
    run<OP_CVT32_64>(bpc, cus, iters, d_out, d_ticks);
    run<OP_CVT64_32>(bpc, cus, iters, d_out, d_ticks);
    run<OP_MOV64>(bpc, cus, iters, d_out, d_ticks);
  }
  return 0;
}
