// chain_device.h — chain mode of the wave-streaming CSV kernels (gfx950): the sums the NEXT iteration needs,
// sum (H - 1/2) and sum I_k (H - 1/2), travel between launches as 64-bit FIXED-POINT integers accumulated with agent-scope
// atomic adds (integer addition is associative: bitwise reproducible whatever the arrival order), so no launch waits for
// a last workgroup; the norm / stop rule / trace of an iteration are booked by an extra workgroup of the FOLLOWING launch
// (or by the flush kernel at a host synchronisation point).  See CvhChainAcc in cvh_internal.h for the rotation.
#pragma once
#include "csv_device.h"

namespace cvh_dev {

typedef const int __attribute__((address_space(4))) *const_int_p;   // read through the scalar (constant) cache: s_load

// Sum of a 64-bit integer over each ROW of 16 lanes (DPP, exact).
__device__ __forceinline__ long long row16_sum_i64(long long v)
{
  auto dpp = [](long long x, auto ctrl_tag) {
    constexpr int ctrl = decltype(ctrl_tag)::value;
    const int lo = __builtin_amdgcn_mov_dpp((int)x, ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(x >> 32), ctrl, 0xf, 0xf, true);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x124>{});  // row_ror:4
  v += dpp(v, std::integral_constant<int, 0x128>{});  // row_ror:8
  return v;
}
__device__ __forceinline__ long long read_lane_i64(long long v, int l)
{
  const int lo = __builtin_amdgcn_readlane((int)v, l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// A set is 64 integers: (1 + C) sums x 64 / (1 + C) shards (C = 1: 2 x 32, C = 3: 4 x 16); lane l of a wave owns entry l.
template <int C> __host__ __device__ constexpr int chain_shards() { return 64 / (1 + C); }

// c1 / c2 of the level set a launch reads, from the set's 64 entries (one per lane).  Same formula as finalize():
// nom / denom (src/main.cpp:280) with the centred sums shifted back by N/2 and sum(I_k)/2, complements from exact totals.
template <int C>
__device__ __forceinline__ void chain_means(const CvhStepArgs &a, long long entry, double (&c1)[C], double (&c2)[C])
{
  const long long r = row16_sum_i64(entry);
  long long q[1 + C];
  if (C == 1) { q[0] = read_lane_i64(r, 0) + read_lane_i64(r, 16); q[1] = read_lane_i64(r, 32) + read_lane_i64(r, 48); }
  else {
#pragma unroll
    for (int s = 0; s < 1 + C; ++s) q[s] = read_lane_i64(r, 16 * s);
  }
  const double sh = __builtin_fma((double)q[0], a.chain_inv[0], 0.5 * a.npix);             // sum H
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const double sih = __builtin_fma((double)q[1 + k], a.chain_inv[1 + k], 0.5 * a.sum_img[k]);   // sum I_k H
    c1[k] = sih / sh;
    c2[k] = (a.sum_img[k] - sih) / (a.npix - sh);
  }
}

// A workgroup's contribution: its sums of H - 1/2 and I_k (H - 1/2) as fixed-point integers into the set the next launch
// reads (shard by workgroup index: no queue on one address), its sum u_diff^2 as a row for the bookkeeper.  `total` is
// block_reduce's result: thread t < NS holds sum t ([0] sum H', [2 + k] sum I_k H', [2 + 2C] sum u_diff^2).
template <int C>
__device__ __forceinline__ void chain_publish(const CvhStepArgs &a, double total)
{
  const int tid = threadIdx.x;
  long long *const set = &a.chain->v[(a.chain_phase + 1) & 3][0];
  const int shard = (int)blockIdx.x % chain_shards<C>();
  if (tid == 0)
    __hip_atomic_fetch_add(&set[shard], __double2ll_rn(total * a.chain_scale[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid >= 2 && tid < 2 + C)
    __hip_atomic_fetch_add(&set[(tid - 1) * chain_shards<C>() + shard], __double2ll_rn(total * a.chain_scale[tid - 1]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 2 + 2 * C) a.chain_s4[(size_t)(a.chain_phase & 1) * a.nparts + blockIdx.x] = total;
}

// The extra workgroup of every chain-mode launch (and the whole of the flush kernel): books the iteration of the
// PREVIOUS launch -- adds its per-workgroup sum u_diff^2 rows in a fixed order, norm, trace row, stop rule
// (src/main.cpp:993-1000) -- off the critical path of the launch that runs beside it.  A stop found here means the
// launch running now computes an iteration the reference never executes: its output goes to the other ping-pong
// buffer and is never read (steps_done stays at the stopping iteration, later launches are no-ops).
// Returns the number of iterations booked so far (= index of the launch running now).
template <int C>
__device__ int chain_bookkeeping(const CvhStepArgs &a, bool flush, const double *c1_now, const double *c2_now, double *sred)
{
  const int tid = threadIdx.x;
  CvhState *st = a.st;
  const int pending = st->pending, t = st->steps_done;   // t = iterations booked so far = index of the pending launch
  constexpr int TR = 2 * C + 1;
  int stop_now = 0;
  if (pending) {
    const double *rows = a.chain_s4 + (size_t)((a.chain_pb + t) & 1) * a.nparts;
    double acc[1] = {0.0};
    for (int b = tid; b < a.nparts; b += CVH_BLOCK) acc[0] += rows[b];
    const double total = block_reduce<1>(acc, sred);
    if (tid == 0) {
      const double nrm = sqrt(total);
      if (a.trace && t < a.trace_cap) a.trace[(size_t)t * TR + 2 * C] = nrm;
      st->norm = nrm;
      st->steps_done = t + 1;
      stop_now = nrm <= a.stop_cond;                    // src/main.cpp:1000, after the update
      if (stop_now) st->stopped = 1;
      if (a.host_status) {
        __hip_atomic_store(&a.host_status[1], stop_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.host_status[0], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  const int tn = pending ? t + 1 : t;                   // iterations booked now = index of the launch running now
  if (tid == 0) {
    if (flush) {
      st->pending = 0;
    } else {
      st->pending = stop_now ? 0 : 1;
      if (!stop_now && a.trace && tn < a.trace_cap) {
        for (int k = 0; k < C; ++k) { a.trace[(size_t)tn * TR + k] = c1_now[k]; a.trace[(size_t)tn * TR + C + k] = c2_now[k]; }
      }
    }
  }
  return tn;
}

// Body of the bookkeeping workgroup inside a step kernel.
template <int C>
__device__ __forceinline__ void chain_bookkeeper_block(const CvhStepArgs &a, long long entry, double *sred)
{
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) a.chain->v[(a.chain_phase + 2) & 3][lane] = 0;   // the set the NEXT launch adds into
  double c1[C], c2[C];
  chain_means<C>(a, entry, c1, c2);
  chain_bookkeeping<C>(a, false, c1, c2, sred);
}

}  // namespace cvh_dev
