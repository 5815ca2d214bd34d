// wave_math.h — FAST-flavour arithmetic helpers shared by the wave-streaming CSV kernels (gfx950).
#pragma once
#include "csv_device.h"

namespace cvh_dev {

// (pi/4 + atan(c))/pi from the table plus the series of the small remainder — see
// cvh_fill_atan2_table.  H_eps(x) = 1/2 + copysign(atan|x| / pi, x).  No selects:
// atan(a) = pi/4 + atan((a-1)/(a+1)); with y ~ (a-1)/(a+1) rounded to c = j/128,
// atan(y) = atan(c) + atan(z), z = (n - c d)/(d + c n), n = a-1, d = a+1: ONE accurate
// reciprocal (of d + c n) and one raw one (to pick c).  |z| <= 1/256.
// d = a*b + c as a 3-address v_fma_f64: with a constant addend hipcc otherwise copies the
// constant into the destination first (v_mov_b64 + v_fmac_f64), one extra VALU slot per use.
__device__ __forceinline__ double fma3(double a, double b, double c)
{
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// Coefficients of the far-field series with eps and 1/pi folded in (wave-uniform).
struct FarCoef { double k0, k1, k2, k3, k4, thr; };

// H_eps(u) - 1/2 = copysign(atan(|u|/eps)/pi, u).  The sums carry this CENTRED value (the
// finalisation adds N/2 and sum(I)/2, exact integers or half-integers): one addition less per pixel.
//
// Far field (|u| >= 32 eps; with the reference's default time step that is every pixel away from
// the contour after a handful of iterations): atan(a) = pi/2 - atan(1/a), a = |u|/eps, and the
// series of atan(t), t = eps/|u| <= 1/32, truncated after t^9/9 (next term < 2.5e-18) needs one
// reciprocal and no table.  With r = 1/|u|:
//   atan(eps r)/pi = r (k0 + r^2 (k1 + r^2 (k2 + r^2 (k3 + r^2 k4)))), k_i = (-1)^i eps^(2i+1)/((2i+1) pi).
// (The diagnostic option far_terms = 4 zeroes k4 and moves the threshold to 64 eps: round 1's 4-term series.)
// The argument is CLAMPED to the far field so that the formula stays finite on every lane: the
// march evaluates it unconditionally (no branch inside a row: a group of 4 rows is one basic block
// and hipcc overlaps the rows' dependent chains), and the rare lanes with |u| below the threshold are
// corrected per group by near_field_correction().
__device__ __forceinline__ double heaviside_centred_far(double u, const FarCoef &fc)
{
  const double au = fmax(fabs(u), fc.thr);
  const double r0 = __builtin_amdgcn_rcp(au);
  const double r = __builtin_fma(__builtin_fma(-au, r0, 1.0), r0, r0);
  const double r2 = r * r;
  double p = fma3(r2, fc.k4, fc.k3);
  p = fma3(p, r2, fc.k2);
  p = fma3(p, r2, fc.k1);
  p = fma3(p, r2, fc.k0);
  return __builtin_copysign(__builtin_fma(-r, p, 0.5), u);
}

// Table form for |u| below the far threshold (valid for any u): atan(a) = pi/4 + atan((a-1)/(a+1)); with
// y ~ (a-1)/(a+1) rounded to c = j/128, atan(y) = atan(c) + atan(z), z = (n - c d)/(d + c n),
// n = a-1, d = a+1: ONE accurate reciprocal (of d + c n) and one raw one (to pick c).  |z| <= 1/256.
__device__ __forceinline__ double heaviside_centred_near(double u, double inv_eps, const double *tab /*LDS, CVH_ATAN2_N*/)
{
  const double x = u * inv_eps;
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
  const double atpi = __builtin_fma(az, 1.0 / kPi, tab[j + 128]);  // atan(a)/pi in [0, 1/2]
  return __builtin_copysign(atpi, x);
}

// FAST form of d+ / sqrt(d+^2 + d0^2 + eta^2) (src/main.cpp:365-368) from the three samples
// along one axis: evaluated as 2d+ / sqrt((2d+)^2 + (2d0)^2 + 4 eta^2) -- the same value bit for
// bit (every intermediate is an exact power-of-two multiple), one instruction shorter because
// 2 d0 = fwd - bwd needs no halving and 2 u(0) is shared by both axes.
__device__ __forceinline__ double normalised4(double fwd, double bwd, double centre2)
{
  const double a = __builtin_fma(fwd, 2.0, -centre2), d = fwd - bwd;
  return a * rsqrt_refined(__builtin_fma(a, a, __builtin_fma(d, d, 4.0 * kEta2)));
}

// Byte BYTE (0 or 1) of `word` times 16 -- the byte offset of a 16-byte {term, sample} table entry -- in ONE instruction: an SDWA
// operand select on the shift (hipcc emits v_and / v_bfe + v_lshl_add: two).  With lds_read_d2 below the hot path of a group of four rows
// is 475 instead of 490 instructions (three channels: 558 instead of 581) -- measured in one process against builds without either
// (round 4, gpurun_out/r4s5, r4s6): no difference beyond the +-0.3 us between two contexts at 4096^2 x 1, -0.7 us at 4096^2 x 3.
template <int BYTE>
__device__ __forceinline__ unsigned byte_x16(unsigned word)
{
  static_assert(BYTE == 0 || BYTE == 1, "the samples of a lane's two pixels are bytes 0 and 1 of the word");
  unsigned r;
  if (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(word));
  else asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(word));
  return r;
}

// 16-byte LDS read at an INTEGER byte address.  The kernels' only LDS is the dynamic `extern __shared__` block, which starts at LDS
// address 0 (lds_base_is_zero() guards that): a table entry's address is then (sample x 16) + a compile-time constant, and the constant
// rides in the instruction's offset field -- through a generic pointer hipcc adds the block's (relocatable, zero) base with a
// v_add_u32 per lookup.
typedef double lds_double2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lds_double2_t lds_read_d2(unsigned byte_addr)
{
  return *(const __attribute__((address_space(3))) lds_double2_t *)(unsigned long)byte_addr;
}
__device__ __forceinline__ bool lds_base_is_zero(const void *dynamic_smem)
{
  return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char *)dynamic_smem == 0u;
}

__device__ __forceinline__ double dpp_from_left(double v)
{
  const long long vb = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_mov_dpp((int)vb, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(vb >> 32), 0x138, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// lane l gets lane l - 1's v (wave_shr:1); lane 0 has no source lane and keeps `old` (bound_ctrl:0): no select for the wave's left edge
__device__ __forceinline__ double dpp_from_left_or(double old, double v)
{
  const long long vb = __double_as_longlong(v), ob = __double_as_longlong(old);
  const int lo = __builtin_amdgcn_update_dpp((int)ob, (int)vb, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(ob >> 32), (int)(vb >> 32), 0x138, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

}  // namespace cvh_dev
