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
#ifndef CVH_LOAD_AUX
#define CVH_LOAD_AUX 0
#endif
#ifndef CVH_LOADI_AUX
#define CVH_LOADI_AUX 0
#endif
#ifndef CVH_STORE_AUX
#define CVH_STORE_AUX 0
#endif
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOffset = 0x80000000u;   // beyond any buffer this kernel accepts (< 2 GiB)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000 /* raw, 32-bit data format (gfx950) */);
}
// Diagnostic builds (tools/abl_bench.sh; results are wrong by design): -DCVH_ABLATE_MEMORY replaces every
// global load of the march by two integer instructions and drops the stores; -DCVH_ABLATE_COMPUTE keeps
// the memory and LDS traffic and drops the arithmetic of a row.
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
#ifdef CVH_ABLATE_MEMORY
  return __builtin_bit_cast(double, 0x4000000000000000ull | (unsigned long long)(voff + soff));
#else
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, CVH_LOAD_AUX));
#endif
}
__device__ __forceinline__ u32x4_t buf_load_b128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
#ifdef CVH_ABLATE_MEMORY
  const unsigned v = (voff + soff) * 0x9e3779b1u;
  return u32x4_t{v, v ^ 0x55aa55aau, v + 0x01020304u, v};
#else
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, CVH_LOADI_AUX);
#endif
}
__device__ __forceinline__ void buf_store_f64(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
#ifndef CVH_ABLATE_MEMORY
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, voff, soff, CVH_STORE_AUX);
#endif
}


}  // namespace cvh_dev
