// csv_persist_kernel.hip — the 2-pixel wave kernel (csv_wave2_kernel.hip) as ONE launch for a whole chunk of iterations
// (gfx950, wave64, 1 channel, FAST arithmetic, chain-mode sums).
//
// A kernel boundary per iteration costs the launch gap, a dispatch ramp and -- the largest part -- a prologue in which all
// 3060 waves request their first seven rows at once (tools/wave_timeline.py: first group starts 5.4 us after the launch).
// Here every workgroup keeps its strips for `persist_steps` iterations and the boundary becomes two waits:
//   A  the 8 neighbouring workgroups have finished iteration e   -> rows of u(e+1) this workgroup's prologue reads exist;
//      the prologue loads of iteration e+1 are issued now, while slower workgroups still finish iteration e;
//   B  ALL workgroups have finished iteration e (counter `arrive`) -> the region means of u(e+1) are complete (fixed-point sums of
//      chain mode) and nobody reads the buffer iteration e+1 overwrites any more; compute starts.
// Memory model: level-set rows are stored with sc1 (agent-scope write-through) and a workgroup's rows are complete
// (s_waitcnt vmcnt(0) + barrier) before it raises its flags with agent-scope atomics; the rows are READ with sc1 too
// (agent-scope loads: every XCD has its own L2, and a line it cached two iterations ago must not be served again).
// Every wait is a bounded poll (`persist_poll_cap`): a workgroup that gives up raises CvhPersist::error and leaves, and so
// does every workgroup waiting for it -- the grid always drains.  The launch is cooperative (all workgroups co-resident).
// The bookkeeping workgroup of chain mode (norm, stop rule, trace row of the PREVIOUS iteration) runs the same loop one wait
// behind; a stop it finds is published as the index of the stopping iteration, so that all workgroups leave at the same
// iteration (the one after the garbage iteration, exactly as with one launch per iteration: chain_device.h).
#include "csv_device.h"
#include "buffer_ops.h"
#include "wave_math.h"
#include "chain_device.h"
#include "wave2_device.h"
#include <type_traits>

using namespace cvh_dev;

namespace {

__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Level-set rows are read with sc1 (agent scope): a line another XCD rewrote is never served from this XCD's L2 or the CU's L1.
// (The alternative -- plain loads after an agent-scope acquire fence per workgroup and iteration -- invalidates the whole L2 of
// the XCD ~100 times per iteration, under the workgroups that are still streaming: measured 76 vs XX us at 4096^2.)
constexpr int kCohLoad = 16;
__device__ __forceinline__ double2_t coh_load_f64x2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, kCohLoad));
}
__device__ __forceinline__ double coh_load_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, kCohLoad));
}

constexpr int kWordStride = 16;   // one synchronisation word per 64 bytes: pollers of different words never share a line
constexpr unsigned kAbort = 0xffffffffu;

// Thread 0 polls *p >= target (bounded), the workgroup meets at a barrier; false = gave up (error raised).  Only for words
// with ONE poller (same-address requests serialise at the memory side: hundreds of pollers would starve the writers).
__device__ __forceinline__ bool wg_wait_ge(const unsigned *p, unsigned target, const CvhStepArgs &a, int *s_flag)
{
  if (threadIdx.x == 0) {
    int ok = 0;
    for (int i = 0; i < a.persist_poll_cap; ++i) {
      if (ld_agent(p) >= target) { ok = 1; break; }
      if (i >= 64) __builtin_amdgcn_s_sleep(24); else if (i >= 4) __builtin_amdgcn_s_sleep(4);
    }
    if (!ok) st_agent(&a.persist->error, 1);
    *s_flag = ok;
  }
  __syncthreads();
  const int ok = *s_flag;
  __syncthreads();
  return ok != 0;
}

// Wait for this workgroup's own "go" word to reach iteration `it`: returns the word (bit 0: leave, kAbort: a wait gave up
// somewhere), or kAbort after raising the error itself.
__device__ __forceinline__ unsigned wg_wait_go(const unsigned *p, int it, const CvhStepArgs &a, int *s_flag)
{
  if (threadIdx.x == 0) {
    unsigned v = kAbort;
    int ok = 0;
    for (int i = 0; i < a.persist_poll_cap; ++i) {
      v = ld_agent(p);
      if ((v >> 1) >= (unsigned)it) { ok = 1; break; }
      if (i >= 64) __builtin_amdgcn_s_sleep(24); else if (i >= 2) __builtin_amdgcn_s_sleep(6);
    }
    if (!ok) { st_agent(&a.persist->error, 1); v = kAbort; }
    *s_flag = (int)v;
  }
  __syncthreads();
  const unsigned v = (unsigned)*s_flag;
  __syncthreads();
  return v;
}

// Lanes 0..7 of wave 0 poll the `done` words of the 8 neighbours of workgroup (br, bc) in the nbr x nbc arrangement.
__device__ __forceinline__ bool wg_wait_neighbours(int br, int bc, int nbr, int nbc, unsigned gen, const CvhStepArgs &a, int *s_flag)
{
  if (threadIdx.x < 64) {
    const int l = (int)threadIdx.x;
    const int k = l < 4 ? l : l + 1;                 // 0..8 without the centre
    const int r = br + k / 3 - 1, c = bc + k % 3 - 1;
    const bool have = l < 8 && r >= 0 && r < nbr && c >= 0 && c < nbc;
    const unsigned *p = &a.persist->done[(have ? r * nbc + c : 0) * kWordStride];
    bool sat = !have;
    int ok = 0;
    for (int i = 0; i < a.persist_poll_cap; ++i) {
      if (!sat) sat = ld_agent(p) >= gen;
      if (__builtin_amdgcn_ballot_w64(!sat) == 0ull) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(96);   // ~2.5 us: off the critical path (the last workgroup's neighbours are done long before it is)
    }
    if (l == 0) {
      if (!ok) st_agent(&a.persist->error, 1);
      *s_flag = ok;
    }
  }
  __syncthreads();
  const int ok = *s_flag;
  __syncthreads();
  return ok != 0;
}

// The bookkeeping workgroup: loop j books iteration j - 1 of this launch (j = 0: the last iteration of the launch before),
// as chain_bookkeeping() does for one launch per iteration, and clears the sum set iteration j + 1 adds into.
__device__ void persist_bookkeeper(const CvhStepArgs &a, double *sred, int *s_flag)
{
  constexpr int C = 1, TR = 2 * C + 1;
  CvhPersist *ps = a.persist;
  CvhState *st = a.st;
  const int tid = threadIdx.x, lane = tid & 63;
  int pending = st->pending, t = st->steps_done;      // plain loads: written before this launch began
  for (int j = 0; j < a.persist_steps; ++j) {
    if (j > 0) {
      if (wg_wait_go(&ps->go[(size_t)a.nparts * kWordStride], j, a, s_flag) == kAbort) return;   // everything it reads below is an agent-scope load
    }
    const int phase = (a.chain_phase + j) & 3;
    const long long entry = __hip_atomic_load(&a.chain->v[phase][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) __hip_atomic_store(&a.chain->v[(phase + 2) & 3][lane], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double c1[C], c2[C];
    chain_means<C>(a, entry, c1, c2);
    int stop_now = 0;
    if (pending) {
      const double *rows = a.chain_s4 + (size_t)((a.chain_pb + t) & 1) * a.nparts;
      double acc[1] = {0.0};
      for (int b = tid; b < a.nparts; b += CVH_BLOCK) acc[0] += __hip_atomic_load(&rows[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double total = block_reduce<1>(acc, sred);
      if (tid == 0) {
        const double nrm = sqrt(total);
        if (a.trace && t < a.trace_cap) a.trace[(size_t)t * TR + 2 * C] = nrm;
        st->norm = nrm;
        st->steps_done = t + 1;
        stop_now = nrm <= a.stop_cond;                  // src/main.cpp:1000, after the update
        if (a.host_status) __hip_atomic_store(&a.host_status[0], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        *s_flag = stop_now;
      }
      __syncthreads();
      stop_now = *s_flag;
      __syncthreads();
      t += 1;
    }
    if (tid == 0) {
      st->pending = stop_now ? 0 : 1;
      if (!stop_now && a.trace && t < a.trace_cap) { a.trace[(size_t)t * TR] = c1[0]; a.trace[(size_t)t * TR + C] = c2[0]; }
      if (stop_now) st_agent(&ps->stop_mark, j + 1);   // 2 + (j - 1)
    }
    pending = 1;
    if (stop_now) {
      // the sticky flag is what every workgroup read when the kernel started: it may change only once all of them have
      // (loop 0: not before they finished iteration 0; later loops waited for that already)
      if (j == 0 && a.persist_steps > 1 && !wg_wait_ge(&ps->arrive, (unsigned)a.nparts, a, s_flag)) return;   // its only poller
      if (tid == 0) {
        st->stopped = 1;
        if (a.host_status) __hip_atomic_store(&a.host_status[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) st_agent(&ps->booked, (unsigned)(j + 1));
    if (stop_now) return;
  }
}

__global__ __launch_bounds__(CVH_BLOCK, 3) void csv_persist_kernel(const CvhStepArgs a)
{
  using L = Wave2Smem<true>;
  constexpr int NS = NS2, R = R2;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *sred = smem + L::off_red;
  double *satan = smem + L::off_atan;
  double *slut = smem + L::off_lut;
  int *s_flag = (int *)(smem + L::off_flag);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = a.h, w = a.w;
  CvhPersist *const ps = a.persist;

  // ---- this workgroup's strips: as csv_wave2_kernel (class-major numbering, 2 wave-columns x 2 strips)
  const int nwc = a.tiles_x, nstrips = a.tiles_y;
  const int nbc = (nwc + 1) >> 1, nbr = (nstrips + 1) >> 1;
  int bid = (int)blockIdx.x;
  const bool bookkeeper = bid >= a.nparts;
  if (a.wave_xcd && !bookkeeper) {
    const int nb = a.nparts, x = bid & 7, j = bid >> 3, q = nb >> 3, r = nb & 7;
    if (a.wave_cls > 0) {
      const int S = a.wave_cls, cl = j / S;
      int rank = 0;
      for (int xx = 0; xx < 8; ++xx) {
        const int nx = q + (xx < r ? 1 : 0);
        const int before = nx < cl * S ? nx : cl * S;
        int mine = nx - cl * S;
        mine = mine < 0 ? 0 : (mine > S ? S : mine);
        rank += before + (xx < x ? mine : 0);
      }
      bid = rank + (j - cl * S);
    } else {
      bid = x * q + (x < r ? x : r) + j;
    }
  }
  const int bc = bid % nbc, br = bid / nbc;
  const int wc = bc * 2 + (wave & 1);
  const int ws = br * 2 + (wave >> 1);
  const bool active = !bookkeeper && wc < nwc && ws < nstrips;
  const int wsa = bookkeeper ? 0 : br * 2;
  const const_int_p sb = (const_int_p)a.strip_bounds;
  const int stopped = *(const_int_p)&a.st->stopped;
  const int b0 = sb[wsa], b1 = sb[wsa + 1 <= nstrips ? wsa + 1 : nstrips], b2 = sb[wsa + 2 <= nstrips ? wsa + 2 : nstrips];
  if (stopped != 0) return;   // sticky flag of an EARLIER launch: the same value for every workgroup of this one
  if (bookkeeper) { persist_bookkeeper(a, sred, s_flag); return; }

  // diagnostic stamps (option "debug_times", tools/persist_timeline.py): 10 words per workgroup, taken around iteration kStampIt
  constexpr int kStampIt = 3;
  auto stamp = [&](int it_now, int it_want, int slot) {
    if (a.dbg_times && it_now == it_want && tid == 0) a.dbg_times[(size_t)blockIdx.x * 10 + slot] = __builtin_amdgcn_s_memrealtime();
  };
  double *xs = smem + L::off_x + wave * L::wave_doubles;
  const double l1 = a.lambda1[0], l2 = a.lambda2[0];
  const double eps = a.eps, eps2 = eps * eps;
  const FarCoef fc = {a.far_k[0], a.far_k[1], a.far_k[2], a.far_k[3], a.far_k[4], a.far_thr};
  for (int q = tid; q < CVH_ATAN2_N; q += CVH_BLOCK) satan[q] = a.atan2_tab[q];   // constant over the launch

  int wg_groups = 0;
  {
    const int la = b1 - b0, lb = b2 - b1;
    wg_groups = ((la > lb ? la : lb) + R - 1) / R;
  }

  // ---- per-wave constants of the march (meaningful for active waves)
  const int s0 = (wave >> 1) ? b1 : b0, s1 = (wave >> 1) ? b2 : b1;
  const int c0 = W2 * wc - 2 + 2 * lane;
  const bool lane_valid = lane >= 1 && c0 < w;
  const int cl = c0 < 0 ? 0 : (c0 > w - 2 ? w - 2 : c0);
  const double fxa = (c0 <= 0) ? 0.0 : 1.0;
  const int pa = 2 * lane;
  const int pw = (lane == 0 || c0 <= 0) ? pa : pa - 1;
  const int pe = (c0 + 2 >= w) ? pa + 1 : pa + 2;
  const double2_t *x_own = reinterpret_cast<const double2_t *>(xs + pa);
  const double *x_w = xs + pw, *x_e = xs + pe;
  double2_t *x_put = reinterpret_cast<double2_t *>(xs + pa);
  const bool xlane = lane < R;
  const int xrow = xlane ? lane : 0;
  const int xcol = clampi(W2 * wc + W2, 0, w - 1);
  double *x_ext = xlane ? xs + xrow * XP2 + 128 : xs + R * XP2 + lane;
  const unsigned rowbytes = (unsigned)w * 8u, ubytes = (unsigned)h * rowbytes;
  const unsigned voff_u = (unsigned)cl * 8u;
  const unsigned voff_st = lane_valid ? (unsigned)c0 * 8u : kOobOffset;
  const unsigned voff_x = ((unsigned)xrow * (unsigned)w + (unsigned)xcol) * 8u;
  const int ulast = s1 < h - 1 ? s1 : h - 1, ilast = s1 - 1;
  unsigned char *simg = reinterpret_cast<unsigned char *>(xs + R * XP2 + 64);
  const int icol0 = (W2 * wc - 2) & ~15;
  const int ipiece = lane % 9, irow = lane / 9;
  const bool ilane = lane < 9 * R;
  int ipc = icol0 + 16 * ipiece;
  ipc = ipc < 0 ? 0 : (ipc > w - 16 ? w - 16 : ipc);
  const unsigned voff_i = (unsigned)(ilane ? irow : 0) * (unsigned)w + (unsigned)ipc;
  unsigned char *ipiece_dst = simg + irow * IMGP2 + ((icol0 + 16 * ipiece) == ipc ? 16 * ipiece : ipc - icol0);
  const int ibyte = cl - icol0;
  const __amdgpu_buffer_rsrc_t ri = make_rsrc(a.img[0], (unsigned)h * (unsigned)w);
  auto lds_fence = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const double vmask = lane_valid ? 1.0 : 0.0;

  const int nit = a.persist_steps;
  for (int it = 0; it < nit; ++it) {
    const int phase = (a.chain_phase + it) & 3;
    const double *const uin = (it & 1) ? a.u_out : a.u_in;
    double *const uout = (it & 1) ? const_cast<double *>(a.u_in) : a.u_out;
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(uin, ubytes);
    const __amdgpu_buffer_rsrc_t ro_live = make_rsrc(uout, ubytes), ro_dead = make_rsrc(uout, 0u);

    // ---- wait A: the neighbours' rows of u(it) exist; drop what this CU / XCD cached of the buffer two iterations ago
    if (it > 0) {
      if (!wg_wait_neighbours(br, bc, nbr, nbc, (unsigned)it, a, s_flag)) return;
      stamp(it, kStampIt + 1, 4);
    }

    auto U = [&](int r) -> double2_t { return coh_load_f64x2(ru, voff_u, (unsigned)clampi(r, 0, ulast) * rowbytes); };
    auto UX = [&](int r0) -> double {
      if (r0 + R - 1 <= ulast) return coh_load_f64(ru, voff_x, (unsigned)r0 * rowbytes);
      return coh_load_f64(ru, ((unsigned)clampi(r0 + xrow, 0, ulast) * (unsigned)w + (unsigned)xcol) * 8u, 0u);
    };
    auto IMQ = [&](int r0) -> u32x4_t {
      if (r0 + R - 1 <= ilast) return buf_load_b128(ri, voff_i, (unsigned)r0 * (unsigned)w);
      return buf_load_b128(ri, (unsigned)clampi(r0 + (ilane ? irow : 0), 0, ilast) * (unsigned)w + (unsigned)ipc, 0u);
    };

    // ---- prologue loads (in flight across wait B)
    double2_t um2 = {0, 0}, um = {0, 0}, u0 = {0, 0}, T0[R];
    double X0 = 0, X1 = 0;
    u32x4_t IQ0 = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < R; ++j) T0[j] = double2_t{0, 0};
    if (active) {
      um2 = U(s0 - 2); um = U(s0 - 1); u0 = U(s0);
#pragma unroll
      for (int j = 0; j < R; ++j) T0[j] = U(s0 + 1 + j);
      X0 = UX(s0);
      X1 = UX(s0 + 1);
      IQ0 = IMQ(s0);
    }

    stamp(it, kStampIt + 1, 5);
    // ---- wait B: every workgroup has finished iteration it - 1 and its bookkeeping is one loop behind at most
    if (it > 0) {
      const unsigned go = wg_wait_go(&ps->go[(size_t)blockIdx.x * kWordStride], it, a, s_flag);
      stamp(it, kStampIt + 1, 6);
      if (go & 1u) return;   // the stop rule fired at iteration it - 2 or earlier (it - 1 was the garbage one), or a wait gave up
    }
    // region means of u(it) from the fixed-point sums, then the table of the variance term
    double c1, c2;
    {
      const long long entry = __hip_atomic_load(&a.chain->v[phase][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      double m1[1], m2[1];
      chain_means<1>(a, entry, m1, m2);
      c1 = m1[0]; c2 = m2[0];
      const double v = (double)tid;
      const double d1 = v - c1, d2 = v - c2;
      const double reg = (d2 * d2) * l2 - (d1 * d1) * l1;
      slut[2 * tid] = __builtin_fma(reg, a.beta, a.gamma);
      slut[2 * tid + 1] = v;
    }
    __syncthreads();
    stamp(it, kStampIt, 0); stamp(it, kStampIt + 1, 7);
    if (a.dbg_times && it == kStampIt && tid == 0) {
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      a.dbg_times[(size_t)blockIdx.x * 10 + 9] = (unsigned long long)(xcc & 0xf) | ((unsigned long long)hwid << 8) | ((unsigned long long)(s1 - s0) << 40) |
                                                 ((unsigned long long)bid << 48);
    }

    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0;
    int groups_done = 0;

    if (active) {
      int im[R];
      auto park = [&](const double2_t (&T)[R], double X, u32x4_t IQ) {
        lds_fence();
#pragma unroll
        for (int j = 0; j < R; ++j) x_put[j * (XP2 / 2)] = T[j];
        *x_ext = X;
        if (ilane) *reinterpret_cast<u32x4_t *>(ipiece_dst) = IQ;
        lds_fence();
#pragma unroll
        for (int k = 0; k < R; ++k) im[k] = (int)*reinterpret_cast<const unsigned short *>(simg + k * IMGP2 + ibyte);
      };
      double uw, ue;
      x_put[0] = u0;
      if (xrow == 0) *x_ext = X0;
      lds_fence();
      uw = x_w[0]; ue = x_e[0];
      park(T0, X1, IQ0);

      auto norm = [&](double fwd, double bwd, double c) -> double { return normalised4(fwd, bwd, c + c); };
      double nypa = norm(u0.x, um2.x, um.x), nypb = norm(u0.y, um2.y, um.y);   // ny at row s0-1
      if (s0 == 0) {   // kappa_y(0, .) = 0 (:372): ny_prev := row 0's own ny
        const double2_t up0 = x_own[0];
        nypa = norm(up0.x, um.x, u0.x); nypb = norm(up0.y, um.y, u0.y);
      }
      auto pixel = [&](double c, double n_, double s_, double nx, double nxl, double fx, double &nyp, int byte,
                       double &ud_out, double &Ik_out) -> double {
        const double ny = norm(s_, n_, c);
        const double kappa = __builtin_fma(nx - nxl, fx, ny - nyp);
        const double2_t e = reinterpret_cast<const double2_t *>(slut)[byte];
        double ud = __builtin_fma(kappa, a.alpha, e.x);                  // :985
        const double qd = __builtin_fma(c, c, eps2) * a.dk1;             // 1/delta_eps(u)
        const double r0 = __builtin_amdgcn_rcp(qd);
        const double er = __builtin_fma(-qd, r0, 1.0);
        ud = ud * __builtin_fma(__builtin_fma(er, er, er), r0, r0);      // :992
        nyp = ny;
        ud_out = ud; Ik_out = e.y;
        return c + ud;                                                   // :994
      };
      double2_t keep[R];
      unsigned long long near_mask[R];
      auto row = [&](int i, int k, bool live) {
        const double2_t up = x_own[k * (XP2 / 2)];
        const double uw_n = x_w[k * XP2], ue_n = x_e[k * XP2];
        const int ba = im[k] & 0xff, bb = (im[k] >> 8) & 0xff;
        const double nxa = norm(u0.y, uw, u0.x);
        const double nxb = norm(ue, u0.x, u0.y);
        const double nxla = dpp_from_left(nxb);
        double uda, udb, Ia, Ib, nya = nypa, nyb = nypb;
        const double va = pixel(u0.x, um.x, up.x, nxa, nxla, fxa, nya, ba, uda, Ia);
        const double vb = pixel(u0.y, um.y, up.y, nxb, nxa, 1.0, nyb, bb, udb, Ib);
        keep[k] = double2_t{va, vb};   // stays live to the end of the group: 16-byte store hazard, see csv_wave2_kernel.hip
        const double hva = heaviside_centred_far(va, fc), hvb = heaviside_centred_far(vb, fc);
        near_mask[k] = __builtin_amdgcn_ballot_w64(fabs(va) < fc.thr || fabs(vb) < fc.thr);
        buf_store_f64x2(keep[k], live ? ro_live : ro_dead, voff_st, (unsigned)i * rowbytes);
        if (live) {
          acc[0] += hva; acc[0] += hvb;
          acc[2] = __builtin_fma(Ia, hva, acc[2]); acc[2] = __builtin_fma(Ib, hvb, acc[2]);
          acc[4] = __builtin_fma(uda, uda, acc[4]); acc[4] = __builtin_fma(udb, udb, acc[4]);
          nypa = nya; nypb = nyb;
        }
        um = u0; u0 = up;
        uw = uw_n; ue = ue_n;
      };

      int prio = 3;
      if (a.wave_prio) __builtin_amdgcn_s_setprio(3);
      auto group = [&](int ib, auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        if (a.wave_sync) { __builtin_amdgcn_s_barrier(); ++groups_done; }
        if (a.wave_prio) {   // waves that are ahead lower their priority (csv_wave2_kernel.hip)
          const int rem = s1 - ib, len = s1 - s0;
          const int pq = (rem * 4 - 1) / len;
          if (pq != prio) {
            prio = pq;
            if (pq >= 3) __builtin_amdgcn_s_setprio(3);
            else if (pq == 2) __builtin_amdgcn_s_setprio(2);
            else if (pq == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
          }
        }
        double2_t T[R];
#pragma unroll
        for (int j = 0; j < R; ++j) T[j] = INTERIOR ? coh_load_f64x2(ru, voff_u, (unsigned)(ib + R + 1 + j) * rowbytes) : U(ib + R + 1 + j);
        const double X = INTERIOR ? coh_load_f64(ru, voff_x, (unsigned)(ib + R + 1) * rowbytes) : UX(ib + R + 1);
        const u32x4_t IQ = INTERIOR ? buf_load_b128(ri, voff_i, (unsigned)(ib + R) * (unsigned)w) : IMQ(ib + R);
#pragma unroll
        for (int k = 0; k < R; ++k) {
          if (INTERIOR || (ib + k) < s1) row(ib + k, k, true);
          else near_mask[k] = 0ull;
        }
        if ((near_mask[0] | near_mask[1] | near_mask[2] | near_mask[3]) != 0ull) {
#pragma unroll
          for (int k = 0; k < R; ++k) {
            if (near_mask[k] != 0ull && (INTERIOR || (ib + k) < s1)) {
              const double xa = keep[k].x, xb = keep[k].y;
              const double da = (fabs(xa) < fc.thr) ? heaviside_centred_near(xa, a.inv_eps, satan) - heaviside_centred_far(xa, fc) : 0.0;
              const double db = (fabs(xb) < fc.thr) ? heaviside_centred_near(xb, a.inv_eps, satan) - heaviside_centred_far(xb, fc) : 0.0;
              acc[0] += da; acc[0] += db;
              acc[2] = __builtin_fma((double)(im[k] & 0xff), da, acc[2]);
              acc[2] = __builtin_fma((double)((im[k] >> 8) & 0xff), db, acc[2]);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) asm volatile("; row %2 of the group: store data still live" :: "v"(keep[k].x), "v"(keep[k].y), "n"(0));
        park(T, X, IQ);
      };
      int ib = s0;
      for (; ib + 2 * R <= ulast; ib += R) group(ib, std::true_type{});
      for (; ib < s1; ib += R) group(ib, std::false_type{});
      if (a.wave_prio) __builtin_amdgcn_s_setprio(3);
#pragma unroll
      for (int s = 0; s < NS; ++s) acc[s] = acc[s] * vmask;
    }
    if (a.wave_sync) {
      for (; groups_done < wg_groups; ++groups_done) __builtin_amdgcn_s_barrier();
    }
    stamp(it, kStampIt, 1);
    const double total = block_reduce<NS>(acc, sred);
    // ---- publish (chain_publish with this iteration's phase; the row of sum u_diff^2 is read by another XCD's workgroup)
    {
      long long *const set = &a.chain->v[(phase + 1) & 3][0];
      const int shard = (int)blockIdx.x % chain_shards<1>();
      if (tid == 0)
        __hip_atomic_fetch_add(&set[shard], __double2ll_rn(total * a.chain_scale[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid == 2)
        __hip_atomic_fetch_add(&set[chain_shards<1>() + shard], __double2ll_rn(total * a.chain_scale[1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid == 4) __hip_atomic_store(&a.chain_s4[(size_t)(phase & 1) * a.nparts + blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (it + 1 < nit) {
      // all rows and sums of this workgroup have reached memory (write-through stores, returned atomics) before the flags rise
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      stamp(it, kStampIt, 2);
      if (wave == 0) {
        unsigned old = 0;
        if (lane == 0) {
          st_agent(&ps->done[(size_t)bid * kWordStride], (unsigned)(it + 1));
          old = __hip_atomic_fetch_add(&ps->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        if (old + 1u == (unsigned)(it + 1) * (unsigned)a.nparts) {
          // LAST to arrive: iteration `it` is complete everywhere.  One word per workgroup says "go" (every workgroup polls its
          // own word) and carries the ONE decision all of them follow: leave if the stop rule fired at iteration it - 1 or
          // earlier (booked by the bookkeeper's loop `it`, which this wave waits for -- it started an iteration ago).
          unsigned val = kAbort;
          {
            int ok = 0;
            for (int i = 0; i < a.persist_poll_cap; ++i) {
              if (ld_agent(&ps->booked) >= (unsigned)(it + 1)) { ok = 1; break; }
              __builtin_amdgcn_s_sleep(8);
            }
            if (ok) {
              const int mark = ld_agent(&ps->stop_mark);   // 0, or 2 + the launch-relative index of the stopping iteration
              val = ((unsigned)(it + 1) << 1) | ((mark != 0 && mark <= it + 1) ? 1u : 0u);
            } else if (lane == 0) st_agent(&ps->error, 1);
          }
          for (int i = lane; i <= a.nparts; i += 64) st_agent(&ps->go[(size_t)i * kWordStride], val);
          stamp(it, kStampIt, 8);
        }
        stamp(it, kStampIt, 3);
      }
    }
  }
}

}  // namespace

// Workgroups one CU holds (167 VGPRs, ~27 KB of LDS each: 3): the host sizes the grid so that all of them are resident.
int cvh_persist_blocks_per_cu()
{
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, csv_persist_kernel, CVH_BLOCK, Wave2Smem<true>::bytes) != hipSuccess) return 0;
  return n;
}

hipError_t cvh_launch_persist(const CvhStepArgs &a, hipStream_t s)
{
  CvhStepArgs copy = a;
  void *params[] = {&copy};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void *>(csv_persist_kernel), dim3(a.nparts + 1), dim3(CVH_BLOCK), params,
                                    (unsigned)Wave2Smem<true>::bytes, s);
}
