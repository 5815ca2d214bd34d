// chain_kernels.hip — the flush kernel of chain mode (chain_device.h): at a host synchronisation point the last launch's
// iteration has no successor to book it.  One workgroup books it (norm, stop rule, trace row) and writes the region
// means of the current level set into the state block (cvh_get_means, kernels that read c1/c2 from there).
#include "chain_device.h"

using namespace cvh_dev;

namespace {

template <int C>
__global__ __launch_bounds__(CVH_BLOCK) void csv_chain_flush_kernel(const CvhStepArgs a)
{
  __shared__ double sred[4];
  const int lane = threadIdx.x & 63;
  // sums of the level set after the last BOOKED iteration live in set (pb + steps_done); a stop found now leaves it there
  const int done = chain_bookkeeping<C>(a, true, nullptr, nullptr, sred);
  const long long entry = a.chain->v[(a.chain_pb + done) & 3][lane];
  double c1[C], c2[C];
  chain_means<C>(a, entry, c1, c2);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < C; ++k) { a.st->c1[k] = c1[k]; a.st->c2[k] = c2[k]; }
    // what cvh_sync reports, straight into the pinned host block {steps_done, stopped, norm}: no device-to-host copy behind the flush
    if (a.host_status) {
      const long long nb = __double_as_longlong(a.st->norm);
      __hip_atomic_store(&a.host_status[2], (int)nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&a.host_status[3], (int)(nb >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&a.host_status[1], a.st->stopped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&a.host_status[0], a.st->steps_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace

hipError_t cvh_launch_chain_flush(const CvhStepArgs &a, int channels, hipStream_t s)
{
  if (channels == 1) hipLaunchKernelGGL(csv_chain_flush_kernel<1>, dim3(1), dim3(CVH_BLOCK), 0, s, a);
  else hipLaunchKernelGGL(csv_chain_flush_kernel<3>, dim3(1), dim3(CVH_BLOCK), 0, s, a);
  return hipGetLastError();
}
