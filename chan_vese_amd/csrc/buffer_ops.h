// buffer_ops.h — buffer-instruction helpers shared by the wave-streaming kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace cvh_dev {

// Level-set rows and image pieces move through BUFFER instructions: the row base is a scalar
// offset (soffset), the lane's column a constant VGPR offset -- no per-row vector address
// arithmetic -- and a lane whose offset lies outside the buffer (>= num_records) is dropped
// by the hardware.  That gives a maskless, straight-line, compiler-visible store: hipcc counts
// loads AND stores in its vmcnt waits, so the 4-row load pipeline and the stores stay in flight.
// (An `if (lane_valid)` store is a control-flow diamond; a store hidden in inline assembly is not
// counted, and every counted wait then also drains the stores and the younger loads: measured,
// waves spent 50 % of their cycles in s_waitcnt -- profiles/README.md.)
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOffset = 0x80000000u;   // beyond any buffer this kernel accepts (< 2 GiB)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000 /* raw, 32-bit data format (gfx950) */);
}
// (The ablation builds of round 2 -- every global load of the march replaced by two integer instructions and the stores dropped, or the
// arithmetic of a row dropped with every load, LDS exchange and store kept: 42-46 / 53.5-54.6 us against 59-60.7 us for the full kernel
// at 4096^2 -- are a patch under tools/experiments/pruned_flavours/ now, not macros in the shipped sources.)
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ u32x4_t buf_load_b128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store_f64(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, voff, soff, 0);
}

// A workgroup barrier that orders LDS only.  __syncthreads() also waits for every global store the wave has in flight -- on gfx9 stores count
// in vmcnt like loads, and a wave's memory operations return in order -- so a wave that has just sent its border to the neighbours (resident
// kernels) would sit at the barrier until the last store is acknowledged (0.7-1.5 us) before it may issue the loads it is really waiting for.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace cvh_dev
