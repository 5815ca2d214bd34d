// csv_kernels.hip — the fused Chan-Sandberg-Vese level-set step for gfx950 (CDNA4, wave64).
//
// One launch per iteration of the reference's timestep loop (src/main.cpp:963-1001):
//   * a 256-thread workgroup stages a (R+3) x (256+4) FP64 tile of u (halo 2 left/top,
//     1 right/bottom — the 7-point cross of curvature(), :342-375) in LDS with coalesced
//     row-major reads; out-of-image halo cells are filled by index clamping, which IS
//     BORDER_REPLICATE on u (:351-354);
//   * each thread owns one column and marches down R rows keeping u(i-1), u(i), u(i+1) and
//     the previous row's normalised y-gradient in registers; the normalised x-gradient of
//     the left neighbour comes from the adjacent lane by DPP (wave_shr:1), the wave's own
//     left-edge column is normalised once per tile with lanes mapped to rows;
//   * region terms (:299-312,:979), the addWeighted combine (:985), delta_eps (:204-210,
//     the ParallelPixelFunction map of :988-992), the update (:994) and ||u_diff||^2 (:993)
//     are fused behind the stencil; H_eps(u_new) (:188-194) is evaluated once per pixel and
//     its I-weighted sums are reduced wave -> LDS -> one partial row per workgroup;
//   * the last-arriving workgroup (agent-scope ticket) adds the partial rows in a fixed
//     order and publishes next iteration's c1/c2 (:973-974), the norm and the sticky stop
//     flag (:1000).  Summation order is fixed => bitwise reproducible run to run.
//
// Compile with -ffp-contract=off: the STRICT flavour then rounds every per-pixel operation
// exactly as the reference's x86-64 -O3 build (no FMA); the FAST flavour asks for FMAs
// explicitly.
#include "cvh_internal.h"

namespace {

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

// H_eps and its complement for the sums (src/main.cpp:193, :267).
template <bool FAST>
__device__ __forceinline__ double heaviside(double x, double eps)
{
  (void)FAST;
  return (1 + 2 / kPi * atan(x / eps)) / 2;
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
  for (int b = tid; b < a.nparts; b += CVH_BLOCK) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      acc[s] += __hip_atomic_load(&a.partials[(size_t)b * NS + s], __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
  }
  const double total = block_reduce<NS>(acc, sred);
  if (tid < NS) sfin[tid] = total;
  __syncthreads();
  if (tid == 0) {
    CvhState *st = a.st;
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

template <int C, int R, bool FAST>
__global__ __launch_bounds__(CVH_BLOCK) void csv_step_kernel(const CvhStepArgs a)
{
  constexpr int NS = cvh_nsums(C);
  constexpr int ROWS = R + 3;
  __shared__ double su[ROWS * PITCH];
  __shared__ double sred[4 * NS];
  __shared__ double sfin[NS];
  __shared__ int s_last;

  if (a.st->stopped) return;  // sticky stop: src/main.cpp:1000

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = a.h, w = a.w;
  const int bx = blockIdx.x % a.tiles_x, by = blockIdx.x / a.tiles_x;
  const int i0 = by * R, j0 = bx * TW;
  if (tid == 0) s_last = 0;

  // ---- stage the tile: rows i0-2 .. i0+R, columns j0-2 .. j0+TW+1, replicate by clamping
  for (int idx = tid; idx < ROWS * PITCH; idx += CVH_BLOCK) {
    const int r = idx / PITCH, c = idx - r * PITCH;
    const int gi = clampi(i0 - 2 + r, 0, h - 1), gj = clampi(j0 - 2 + c, 0, w - 1);
    su[idx] = a.u_in[(size_t)gi * w + gj];
  }

  double c1[C], c2[C], l1[C], l2[C];
#pragma unroll
  for (int k = 0; k < C; ++k) { c1[k] = a.st->c1[k]; c2[k] = a.st->c2[k]; l1[k] = a.lambda1[k]; l2[k] = a.lambda2[k]; }
  const double eps = a.eps;
  const double eps2 = eps * eps;
  __syncthreads();

  // ---- normalised x-gradient of this wave's left-edge column, lanes <-> rows
  double nx_edge;
  {
    const int er = lane < R ? lane : R - 1;
    const double *p = &su[(er + 2) * PITCH + wave * 64 + 1];
    nx_edge = normalised<FAST>(p[1] - p[0], central(p[-1], p[1]));
  }

  const int c = tid + 2;
  const int gj = j0 + tid;
  double um = su[1 * PITCH + c], u0 = su[2 * PITCH + c];
  double ny_prev = normalised<FAST>(u0 - um, central(su[c], u0));  // ny at row i0-1

  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0;

#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int gi = i0 + r;
    const double up = su[(r + 3) * PITCH + c];
    const double uw = su[(r + 2) * PITCH + c - 1];
    const double ue = su[(r + 2) * PITCH + c + 1];
    const double nx = normalised<FAST>(ue - u0, central(uw, ue));  // :365-366
    const double ny = normalised<FAST>(up - u0, central(um, up));  // :367-368
    const double nxl = from_left_lane(nx, read_lane(nx_edge, r));
    // backward differences with BORDER_REPLICATE on nx / ny (:371-372): zero on column 0 / row 0
    const double kx = (gj == 0) ? 0.0 : nx - nxl;
    const double ky = (gi == 0) ? 0.0 : ny - ny_prev;
    const double kappa = kx + ky;  // :373

    const bool valid = (gi < h) && (gj < w);
    const size_t g = (size_t)(valid ? gi : 0) * w + (valid ? gj : 0);
    double Ik[C];
    double ud = 0.0;  // :965
#pragma unroll
    for (int k = 0; k < C; ++k) {
      Ik[k] = (double)a.img[k][g];
      const double d1 = Ik[k] - c1[k], d2 = Ik[k] - c2[k];
      const double vin = (d1 * d1) * l1[k];   // variance_penalty, :307-310
      const double vout = (d2 * d2) * l2[k];
      ud += vout - vin;                        // :979
    }
    double delta;
    if (FAST) {
      ud = __builtin_fma(kappa, a.alpha, __builtin_fma(ud, a.beta, a.gamma));
      delta = eps * rcp_refined(kPi * __builtin_fma(u0, u0, eps2));
    } else {
      ud = kappa * a.alpha + ud * a.beta + a.gamma;   // :985
      delta = eps / (kPi * (eps2 + u0 * u0));          // :209
    }
    ud = ud * delta;                                   // :992
    const double un = u0 + ud;                         // :994
    const double hv = heaviside<FAST>(un, eps);
    const double omh = 1 - hv;
    if (valid) {
      a.u_out[g] = un;
      acc[0] += hv;
      acc[1] += omh;
#pragma unroll
      for (int k = 0; k < C; ++k) {
        acc[2 + k] += Ik[k] * hv;        // :276
        acc[2 + C + k] += Ik[k] * omh;
      }
      acc[2 + 2 * C] += ud * ud;         // :993
    }
    um = u0; u0 = up; ny_prev = ny;
  }

  const double total = block_reduce<NS>(acc, sred);
  publish_partials_and_maybe_finalize<C>(a, total, sred, sfin, &s_last, gridDim.x);
}

// Sums of H(u), (1-H(u)), I H, I (1-H) for the initial level set (seeds c1/c2 of step 1).
template <int C>
__global__ __launch_bounds__(CVH_BLOCK) void csv_init_sums_kernel(const CvhStepArgs a)
{
  constexpr int NS = cvh_nsums(C);
  __shared__ double sred[4 * NS];
  const size_t n = (size_t)a.h * a.w;
  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0;
  for (size_t q = (size_t)blockIdx.x * CVH_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * CVH_BLOCK) {
    const double hv = heaviside<false>(a.u_in[q], a.eps);
    const double omh = 1 - hv;
    acc[0] += hv;
    acc[1] += omh;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const double I = (double)a.img[k][q];
      acc[2 + k] += I * hv;
      acc[2 + C + k] += I * omh;
    }
  }
  const double total = block_reduce<NS>(acc, sred);
  if (threadIdx.x < NS) a.partials[(size_t)blockIdx.x * NS + threadIdx.x] = total;
}

template <int C>
__global__ __launch_bounds__(CVH_BLOCK) void csv_finalize_kernel(const CvhStepArgs a, int is_init)
{
  constexpr int NS = cvh_nsums(C);
  __shared__ double sred[4 * NS];
  __shared__ double sfin[NS];
  if (!is_init && a.st->stopped) return;
  finalize<C>(a, is_init, sred, sfin);
}

constexpr int kRows = 16;

template <int C>
hipError_t launch_step_c(const CvhStepArgs &a, int fast, hipStream_t s)
{
  const dim3 grid(a.tiles_x * a.tiles_y), block(CVH_BLOCK);
  if (fast)
    hipLaunchKernelGGL((csv_step_kernel<C, kRows, true>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((csv_step_kernel<C, kRows, false>), grid, block, 0, s, a);
  return hipGetLastError();
}

}  // namespace

int cvh_step_tile_rows(int, int) { return kRows; }

void cvh_step_grid(int h, int w, int *tiles_x, int *tiles_y)
{
  *tiles_x = (w + TW - 1) / TW;
  *tiles_y = (h + kRows - 1) / kRows;
}

hipError_t cvh_launch_step(const CvhStepArgs &a, int channels, int fast, hipStream_t s)
{
  return channels == 1 ? launch_step_c<1>(a, fast, s) : launch_step_c<3>(a, fast, s);
}

int cvh_init_sum_blocks(int h, int w)
{
  const size_t n = (size_t)h * w;
  size_t b = (n + CVH_BLOCK * 8 - 1) / (CVH_BLOCK * 8);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

hipError_t cvh_launch_init_sums(const CvhStepArgs &a, int channels, int, int *nparts_out,
                                hipStream_t s)
{
  const int nb = cvh_init_sum_blocks(a.h, a.w);
  *nparts_out = nb;
  if (channels == 1)
    hipLaunchKernelGGL(csv_init_sums_kernel<1>, dim3(nb), dim3(CVH_BLOCK), 0, s, a);
  else
    hipLaunchKernelGGL(csv_init_sums_kernel<3>, dim3(nb), dim3(CVH_BLOCK), 0, s, a);
  return hipGetLastError();
}

hipError_t cvh_launch_finalize(const CvhStepArgs &a, int channels, int is_init, hipStream_t s)
{
  if (channels == 1)
    hipLaunchKernelGGL(csv_finalize_kernel<1>, dim3(1), dim3(CVH_BLOCK), 0, s, a, is_init);
  else
    hipLaunchKernelGGL(csv_finalize_kernel<3>, dim3(1), dim3(CVH_BLOCK), 0, s, a, is_init);
  return hipGetLastError();
}
