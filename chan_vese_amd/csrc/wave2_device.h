// wave2_device.h — layout constants and 16-byte row accessors shared by the 2-pixel wave kernels
// (csv_wave2_kernel.hip: one iteration per launch; csv_persist_kernel.hip: a chunk of iterations per launch).
#pragma once
#include "csv_device.h"
#include "buffer_ops.h"

namespace cvh_dev {

constexpr int W2 = 126;      // output columns per wave
constexpr int XP2 = 130;     // ring slot pitch in doubles (129 used; 1040 bytes keeps 16-byte alignment)
constexpr int R2 = 4;        // rows per group = ring slots
constexpr int IMGP2 = 144;   // bytes per row of the per-wave image tile (9 x 16-byte pieces)
constexpr int NS2 = cvh_nsums(1);

// C channels: one image tile per channel; the region term is a 256-entry table per channel
template <bool FAST, int C = 1>
struct Wave2Smem {
  static constexpr int NS = cvh_nsums(C);
  static constexpr int wave_doubles = R2 * XP2 + 64 + C * R2 * IMGP2 / 8; // ring + scratch + image tiles
  static constexpr int off_x = 0;
  static constexpr int off_red = off_x + 4 * wave_doubles;                // 4*NS
  static constexpr int off_fin = off_red + 4 * NS + (4 * NS) % 2;         // NS (+1 pad)
  static constexpr int off_atan = off_fin + NS + NS % 2;                  // FAST: CVH_ATAN2_N
  static constexpr int off_lut = (off_atan + (FAST ? CVH_ATAN2_N + 1 : 0) + 1) & ~1;  // FAST: C x 256 x {term, I}
  static constexpr int off_flag = off_lut + (FAST ? C * 512 : 0);
  static constexpr int doubles = off_flag + 2;
  static constexpr size_t bytes = (size_t)doubles * sizeof(double);
};

typedef double double2_t __attribute__((ext_vector_type(2)));
// Cache policy of the level-set rows (gfx950 aux bits: 1 = sc0, 2 = nt, 16 = sc1).  POL 1: stores sc1 (agent-scope write-through: no
// dirty lines pile up in the XCD's L2) and loads sc0 -- faster while the ping-pong pair (mostly) fits the 256 MiB Infinity Cache
// (2048^2: 20.9 vs 22.1 us; 4096^2: 0..-1.3 us); POL 0: plain -- faster beyond it (4608^2: 80.6 vs 85.2 us; 5120^2: 100.5 vs 110.2 us).
// The host picks by footprint (api.hip, fill_args); nt stores +1.6 us, sc1 loads +0.5 us at 4096^2.
template <int POL>
__device__ __forceinline__ double2_t buf_load_f64x2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, POL == 1 ? 1 : (POL == 2 ? 2 : 0)));   // POL 2 (diagnostic): nt loads
}
template <int POL>
__device__ __forceinline__ void buf_store_f64x2(double2_t v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, voff, soff, POL == 1 ? 16 : 0);
}

// The level set as it lives in HBM (option "state"): FP64 (the reference's CV_64FC1, src/main.cpp:225 -- the default and the parity
// mode) or, DECLARED, FP32: 9 instead of 17 bytes per pixel-iteration, every new value rounded to float before it is stored and before
// H_eps of it is summed; arithmetic, tables and sums stay FP64 / 64-bit fixed point.  StateIO<ST32, POL> moves a lane's pair of
// adjacent values (and the one east-extra value) in that format; the LDS ring and every register of the march hold doubles.
typedef float float2_t __attribute__((ext_vector_type(2)));
template <bool ST32, int POL>
struct StateIO {
  using raw2_t = double2_t;          // a lane's pair as loaded
  using raw1_t = double;
  static constexpr unsigned kBytes = 8u;
  static __device__ __forceinline__ raw2_t load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return buf_load_f64x2<POL>(r, voff, soff); }
  static __device__ __forceinline__ raw1_t load1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return buf_load_f64(r, voff, soff); }
  static __device__ __forceinline__ double2_t widen(raw2_t v) { return v; }
  static __device__ __forceinline__ double widen(raw1_t v) { return v; }
  static __device__ __forceinline__ double stored(double v) { return v; }                      // the value a later load returns
  static __device__ __forceinline__ void store2(double2_t v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { buf_store_f64x2<POL>(v, r, voff, soff); }
};
template <int POL>
struct StateIO<true, POL> {
  using raw2_t = float2_t;
  using raw1_t = float;
  static constexpr unsigned kBytes = 4u;
  static __device__ __forceinline__ raw2_t load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
  {
    return __builtin_bit_cast(float2_t, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, POL == 1 ? 1 : (POL == 2 ? 2 : 0)));
  }
  static __device__ __forceinline__ raw1_t load1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
  {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  }
  static __device__ __forceinline__ double2_t widen(raw2_t v) { return double2_t{(double)v.x, (double)v.y}; }
  static __device__ __forceinline__ double widen(raw1_t v) { return (double)v; }
  static __device__ __forceinline__ double stored(double v) { return (double)(float)v; }       // round to nearest even, as the store does
  static __device__ __forceinline__ void store2(double2_t v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
  {   // (an 8-byte store: outside the wide-store data hazard of DESIGN.md 4.1)
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, float2_t{(float)v.x, (float)v.y}), r, voff, soff, POL == 1 ? 16 : 0);
  }
};

}  // namespace cvh_dev
