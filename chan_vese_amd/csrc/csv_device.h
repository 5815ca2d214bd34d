// csv_device.h — device helpers shared by the CSV step kernels (gfx950, wave64).
// Citations are file:line in the reference repository.
#pragma once
#include "cvh_internal.h"
#include <type_traits>

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
// Sum of v over the 64 lanes of the wave, the same value in every lane, by a FIXED tree: xor-1, xor-2
// inside each quad, rotate-by-4 and rotate-by-8 inside each row of 16 (DPP moves: no LDS round trips),
// then ((row0 + row1) + row2) + row3 through scalar registers.  This reduction sits on the critical
// path between two iterations (end of the last wave -> partial row -> ticket -> finalisation).
__device__ __forceinline__ double wave_sum(double v)
{
  auto dpp = [](double x, auto ctrl_tag) {
    constexpr int ctrl = decltype(ctrl_tag)::value;
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), ctrl, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x124>{});  // row_ror:4
  v += dpp(v, std::integral_constant<int, 0x128>{});  // row_ror:8
  return ((read_lane(v, 0) + read_lane(v, 16)) + read_lane(v, 32)) + read_lane(v, 48);
}

// Adds acc[] over the workgroup in a fixed order; on return threads tid < NS hold the
// workgroup total of sum tid in `total` (others undefined).
template <int NS>
__device__ __forceinline__ double block_reduce(double (&acc)[NS], double *sred /*[4*NS]*/)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double v[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) v[s] = wave_sum(acc[s]);
  if (lane == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) sred[wave * NS + s] = v[s];
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
  // The finalising workgroup sits on the critical path of every iteration: the state words
  // thread 0 needs are requested before the partial rows and the stop condition is a launch argument.
  int t_pref = 0;
  double c_old[2 * C];
  if (tid == 0 && !is_init) {
    t_pref = a.st->steps_done;
    if (a.trace) {
#pragma unroll
      for (int k = 0; k < C; ++k) { c_old[k] = a.st->c1[k]; c_old[C + k] = a.st->c2[k]; }
    }
  }
  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0;
  // Six partial rows per thread are requested before the first is added (one memory round trip per
  // six rows instead of one per row); the order of the additions is fixed.
  constexpr int UNR = 6;   // 6 x NS doubles in flight per thread: fits the step kernels' register caps
  const bool derived = a.derive_complement && !is_init;  // FAST flavour: sums [1], [2+C..2+2C) are derived, their slots hold zeros
  for (int b0 = tid; b0 < a.nparts; b0 += CVH_BLOCK * UNR) {
    double v[UNR][NS];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int b = b0 + u * CVH_BLOCK;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const bool unused = derived && (s == 1 || (s >= 2 + C && s < 2 + 2 * C));
        v[u][s] = 0.0;
        if (b < a.nparts && !unused)
          v[u][s] = __hip_atomic_load(&a.partials[(size_t)b * NS + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (b0 + u * CVH_BLOCK < a.nparts) {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] += v[u][s];
      }
    }
  }
  if (a.dbg_times && tid == 0 && !is_init) a.dbg_times[(size_t)a.nparts * 18 + 2] = __builtin_amdgcn_s_memrealtime();
  const double total = block_reduce<NS>(acc, sred);
  if (a.dbg_times && tid == 0 && !is_init) a.dbg_times[(size_t)a.nparts * 18 + 3] = __builtin_amdgcn_s_memrealtime();
  if (tid < NS) sfin[tid] = total;
  __syncthreads();
  if (is_init && a.chain) {
    // chain mode (chain_device.h): seed the fixed-point sum set this step reads with the sums of the initial level set
    // (centred: minus N/2 and sum(I_k)/2), all in shard 0, and clear the set the first launch adds into
    constexpr int SH = 64 / (1 + C);
    for (int i = tid; i < 128; i += CVH_BLOCK) {
      const int second = i >= 64, j = i & 63, sum = j / SH, shard = j % SH;
      long long v = 0;
      if (!second && shard == 0) {
        const double centred = sum == 0 ? sfin[0] - 0.5 * a.npix : sfin[1 + sum] - 0.5 * a.sum_img[sum - 1];
        v = __double2ll_rn(centred * a.chain_scale[sum]);
      }
      a.chain->v[(a.chain_phase + second) & 3][j] = v;
    }
  }
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
      const int t = t_pref;  // index of the step just executed
      if (a.trace && t < a.trace_cap) {
        double *row = a.trace + (size_t)t * (2 * C + 1);
        for (int k = 0; k < C; ++k) { row[k] = c_old[k]; row[C + k] = c_old[C + k]; }
        row[2 * C] = nrm;
      }
      st->norm = nrm;
      st->steps_done = t + 1;
      const int stop_now = nrm <= a.stop_cond;  // src/main.cpp:1000, after the update
      if (stop_now) st->stopped = 1;
      if (a.host_status) {  // the host polls these two words in pinned memory instead of copying the state back
        __hip_atomic_store(&a.host_status[1], stop_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.host_status[0], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    for (int k = 0; k < C; ++k) {
      st->c1[k] = sfin[2 + k] / sfin[0];          // nom / denom, src/main.cpp:280
      st->c2[k] = sfin[2 + C + k] / sfin[1];
    }
    st->ticket = 0;
    if (is_init) st->pending = 0;
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
  // Hand-off of the partial rows to the last-arriving workgroup.  This is the hardware-level form that
  // /opt/skills/guides/MI355X_MICROARCH.md (Workgroup dispatch ... inter-workgroup visibility, "Valid forms",
  // first table row) documents as measured-valid on gfx950 in place of a language-level release/acquire pair:
  //   producer: EVERY handed-off byte stored write-through (`sc1`: relaxed agent-scope atomic store), the storing
  //             wave drains them (`s_waitcnt vmcnt(0)`), and only then ONE lane of the same wave adds to the counter;
  //   consumer: the workgroup whose add returned nblocks-1 loads every handed-off byte with `sc1` loads
  //             (relaxed agent-scope atomic loads in finalize()), after a workgroup barrier that the adding wave joins.
  // A language-level __ATOMIC_RELEASE on the ticket would lower to `buffer_wbl2 sc1` (a write-back of the XCD's
  // whole L2, ~1.7-6.5 us per workgroup on this critical path) and buy nothing here: the rows never sit dirty in L2.
  // The acquire fence below is kept for the plain loads of the state words (c1/c2/steps_done of the previous launch).
  // gfx950-only by design (this library targets nothing else); not a portable C++ memory-model hand-off.
  if (tid < NS)
    __hip_atomic_store(&a.partials[(size_t)blockIdx.x * NS + tid], total, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  if (!a.fused_finalize) return;
  if (tid < 64) {  // the storing wave is the signalling wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.dbg_times && tid == 0) a.dbg_times[(size_t)nblocks * 19 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
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
  if (a.dbg_times && tid == 0) {  // diagnostic stamps: after the partial row is out, after the ticket
    a.dbg_times[(size_t)nblocks * 17 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  }
  __syncthreads();
  if (*s_last) {
    if (a.dbg_times && tid == 0) a.dbg_times[(size_t)nblocks * 18] = __builtin_amdgcn_s_memrealtime();
    finalize<C>(a, 0, sred, sfin);
    if (a.dbg_times && tid == 0) a.dbg_times[(size_t)nblocks * 18 + 1] = __builtin_amdgcn_s_memrealtime();
  }
}


}  // namespace cvh_dev
