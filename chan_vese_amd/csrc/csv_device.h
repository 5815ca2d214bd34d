// csv_device.h — device helpers shared by the CSV step kernels (gfx950, wave64).
// Citations are file:line in the reference repository.
#pragma once
#include "cvh_internal.h"

namespace cvh_dev {


constexpr double kPi = 3.14159265358979323846;  // boost::math::constants::pi<double>()
constexpr double kEta2 = 1E-8 * 1E-8;           // std::pow(eta, 2), src/main.cpp:347-348
constexpr int TW = 256;                          // tile width = one column per thread
constexpr int PITCH = TW + 4;                    // LDS row pitch (halo 2 + 2)

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// 1/sqrt(s) to <= 1 ulp from the hardware estimate with one cubic (Halley) step.
__device__ __forceinline__ double rsqrt_refined(double s)
{
  const double r = __builtin_amdgcn_rsq(s);
  const double e = __builtin_fma(-(s * r), r, 1.0);           // 1 - s r^2
  const double p = __builtin_fma(e, 0.375, 0.5);              // 1/2 + 3e/8
  return __builtin_fma(r * e, p, r);                          // r (1 + e/2 + 3e^2/8)
}

// 1/q the same way (cubic step on the hardware reciprocal).
__device__ __forceinline__ double rcp_refined(double q)
{
  const double r = __builtin_amdgcn_rcp(q);
  const double e = __builtin_fma(-q, r, 1.0);
  return __builtin_fma(__builtin_fma(e, e, e), r, r);
}

// d+ / sqrt(d+^2 + d0^2 + eta^2): src/main.cpp:365-368 (same-axis pairing).
template <bool FAST>
__device__ __forceinline__ double normalised(double up, double uc)
{
  if (FAST) {
    const double s = __builtin_fma(up, up, __builtin_fma(uc, uc, kEta2));
    return up * rsqrt_refined(s);
  }
  return up / sqrt(up * up + uc * uc + kEta2);
}

// central difference as filter2D evaluates it: (-0.5)*a + 0.5*b (exactly 0.5*(b-a)).
__device__ __forceinline__ double central(double a, double b) { return -0.5 * a + 0.5 * b; }

// value of `v` in lane-1; lane 0 of the wave receives `edge`.
__device__ __forceinline__ double from_left_lane(double v, double edge)
{
  const long long vb = __double_as_longlong(v), eb = __double_as_longlong(edge);
  const int lo = __builtin_amdgcn_update_dpp((int)eb, (int)vb, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(eb >> 32), (int)(vb >> 32), 0x138, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double read_lane(double v, int l)
{
  const long long vb = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)vb, l);
  const int hi = __builtin_amdgcn_readlane((int)(vb >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// H_eps for the sums (src/main.cpp:193), reference rounding: libm-grade atan, IEEE divide.
__device__ __forceinline__ double heaviside_strict(double x, double eps)
{
  return (1 + 2 / kPi * atan(x / eps)) / 2;
}

// atan(x) to ~1 ulp with ONE reciprocal: table of atan(c), c = i/128, in LDS plus
// atan(x) = atan(c) + atan(z), z = (x - c)/(1 + x c); for |x| > 1 the same with t = 1/|x|:
// atan|x| = pi/2 - atan(c) - atan(z), z = (1 - c|x|)/(|x| + c), c = round(128/|x|)/128.
// |z| <= 1/256, so z - z^3/3 + z^5/5 truncates below 2e-18.
__device__ __forceinline__ double atan_table(double x, const double *tab /*LDS [2][CVH_ATAN_N]*/)
{
  const double ax = fmin(fabs(x), 1e300);
  const bool big = ax > 1.0;
  const double sel = big ? __builtin_amdgcn_rcp(ax) : ax;
  const double fi = __builtin_rint(sel * (double)(CVH_ATAN_N - 1));
  const double c = fi * (1.0 / (CVH_ATAN_N - 1));
  const int idx = (int)fi + (big ? CVH_ATAN_N : 0);
  const double A = big ? 1.0 : ax, B = big ? ax : 1.0;
  const double num = __builtin_fma(-c, B, A);
  const double den = __builtin_fma(c, A, B);
  const double r0 = __builtin_amdgcn_rcp(den);
  const double r = __builtin_fma(__builtin_fma(-den, r0, 1.0), r0, r0);  // one Newton step
  const double z = num * r;
  const double z2 = z * z;
  const double p = __builtin_fma(z2, 0.2, -1.0 / 3.0);
  double az = __builtin_fma(z * z2, p, z);
  az = big ? -az : az;
  const double res = tab[idx] + az;
  return __builtin_copysign(res, x);
}

// Adds acc[] over the workgroup in a fixed order; on return threads tid < NS hold the
// workgroup total of sum tid in `total` (others undefined).
template <int NS>
__device__ __forceinline__ double block_reduce(double (&acc)[NS], double *sred /*[4*NS]*/)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    double v = acc[s];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) sred[wave * NS + s] = v;
  }
  __syncthreads();
  double total = 0;
  if (tid < NS) total = ((sred[tid] + sred[NS + tid]) + sred[2 * NS + tid]) + sred[3 * NS + tid];
  __syncthreads();
  return total;
}

// Adds the partial rows (fixed order), then publishes c1/c2, norm, trace row, stop flag.
// Called by all 256 threads of ONE workgroup.
template <int C>
__device__ void finalize(const CvhStepArgs &a, int is_init, double *sred, double *sfin)
{
  constexpr int NS = cvh_nsums(C);
  const int tid = threadIdx.x;
  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0;
  // Eight partial rows per thread are requested before the first is added (one L2 round trip per
  // eight rows instead of one per row); the order of the additions is fixed.
  constexpr int UNR = 8;
  for (int b0 = tid; b0 < a.nparts; b0 += CVH_BLOCK * UNR) {
    double v[UNR][NS];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int b = b0 + u * CVH_BLOCK;
      const int bc = b < a.nparts ? b : b0;
#pragma unroll
      for (int s = 0; s < NS; ++s)
        v[u][s] = __hip_atomic_load(&a.partials[(size_t)bc * NS + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (b0 + u * CVH_BLOCK < a.nparts) {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] += v[u][s];
      }
    }
  }
  const double total = block_reduce<NS>(acc, sred);
  if (tid < NS) sfin[tid] = total;
  __syncthreads();
  if (tid == 0) {
    CvhState *st = a.st;
    if (a.derive_complement && !is_init) {
      // FAST flavour carries only sum H and sum I H; complements from exact totals.
      // derive_complement == 2 (wave kernel): the sums are of H - 1/2.
      if (a.derive_complement == 2) {
        sfin[0] += 0.5 * a.npix;
        for (int k = 0; k < C; ++k) sfin[2 + k] += 0.5 * a.sum_img[k];
      }
      sfin[1] = a.npix - sfin[0];
      for (int k = 0; k < C; ++k) sfin[2 + C + k] = a.sum_img[k] - sfin[2 + k];
    }
    if (!is_init) {
      const double nrm = sqrt(sfin[2 + 2 * C]);
      const int t = st->steps_done;  // index of the step just executed
      if (a.trace && t < a.trace_cap) {
        double *row = a.trace + (size_t)t * (2 * C + 1);
        for (int k = 0; k < C; ++k) { row[k] = st->c1[k]; row[C + k] = st->c2[k]; }
        row[2 * C] = nrm;
      }
      st->norm = nrm;
      st->steps_done = t + 1;
      if (nrm <= st->stop_cond) st->stopped = 1;  // src/main.cpp:1000, after the update
      if (a.host_status) {  // the host polls these two words in pinned memory instead of copying the state back
        __hip_atomic_store(&a.host_status[1], st->stopped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.host_status[0], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    for (int k = 0; k < C; ++k) {
      st->c1[k] = sfin[2 + k] / sfin[0];          // nom / denom, src/main.cpp:280
      st->c2[k] = sfin[2 + C + k] / sfin[1];
    }
    st->ticket = 0;
  }
}

template <int C>
__device__ __forceinline__ void publish_partials_and_maybe_finalize(const CvhStepArgs &a,
                                                                    double total, double *sred,
                                                                    double *sfin, int *s_last,
                                                                    int nblocks)
{
  constexpr int NS = cvh_nsums(C);
  const int tid = threadIdx.x;
  // write-through (sc1) stores of this workgroup's row, drained before the ticket
  if (tid < NS)
    __hip_atomic_store(&a.partials[(size_t)blockIdx.x * NS + tid], total, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  if (!a.fused_finalize) return;
  if (tid < 64) {  // the storing wave is the signalling wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(&a.st->ticket, 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
      *s_last = (t == (unsigned)nblocks - 1u);
      if (*s_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  __syncthreads();
  if (*s_last) finalize<C>(a, 0, sred, sfin);
}


}  // namespace cvh_dev
