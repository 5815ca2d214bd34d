// csv_wave_kernel.hip — wave-streaming variant of the fused CSV step (gfx950, wave64).
//
// Same arithmetic as csv_kernels.hip; data flow chosen from measurements on MI355X
// (tools/dp_rate_probe.hip, profiles/): the step is FP64-VALU-bound, and a wave64 FP64
// instruction costs ~3.3 SIMD-cycles at 4 waves/SIMD but ~2.4 at 8.  So this variant spends
// almost no LDS and few registers to run 7-8 waves per SIMD, and removes every VALU
// instruction that only moved data:
//   * each WAVE owns a strip of 63 output columns (lane 0 is the left halo column; lanes
//     1..63 produce pixels) and marches down `strip_rows` rows on its own: no workgroup
//     barrier inside the loop, waves drift apart and hide each other's memory latency;
//   * u(i-1), u(i), u(i+1) of the lane's own column and the previous row's normalised
//     y-gradient live in registers; rows are loaded straight into registers 4 rows ahead
//     (coalesced 512-byte wave loads), the image 4 rows ahead;
//   * x-neighbours go through a 4-slot, 66-double per-wave LDS row buffer (written when a
//     row arrives, read one row ahead of use: no exposed LDS latency, no VALU);
//     the two extra halo columns of 4 rows are fetched by 8 lanes in one load;
//   * the left neighbour's normalised x-gradient comes by DPP from lane-1; lane 0 computes
//     the halo column's gradient like any other lane, so nothing is recomputed per tile;
//   * borders: clamped column/row indices are BORDER_REPLICATE on u; the second-level rule
//     (kappa_x(.,0) = 0, kappa_y(0,.) = 0, src/main.cpp:371-372) is a 0/1 factor.
// One partial row of sums per workgroup; finalisation as in the other variants.
#include "csv_device.h"

using namespace cvh_dev;

namespace {

constexpr int WCOLS = 63;   // output columns per wave
constexpr int XPITCH = 66;  // exchange row: [0] = col-2 of lane 0, [1..64] = lanes, [65] = col+1 of lane 63
constexpr int XSLOTS = 4;

template <int C, bool FAST, bool LUT>
struct WaveSmem {
  static constexpr int NS = cvh_nsums(C);
  static constexpr int off_x = 0;                                          // 4 waves x XSLOTS x XPITCH
  static constexpr int off_red = off_x + 4 * XSLOTS * XPITCH;              // 4*NS
  static constexpr int off_fin = off_red + 4 * NS + (4 * NS) % 2;          // NS
  static constexpr int off_atan = off_fin + NS + NS % 2;                   // FAST: CVH_ATAN2_N
  static constexpr int off_lut = off_atan + (FAST ? CVH_ATAN2_N + 1 : 0);  // LUT: C*256
  static constexpr int off_flag = off_lut + (LUT ? C * 256 : 0);
  static constexpr int doubles = off_flag + 2;
  static constexpr size_t bytes = (size_t)doubles * sizeof(double);
};

// (pi/4 + atan(c))/pi from the table plus the series of the small remainder — see
// cvh_fill_atan2_table.  H_eps(x) = 1/2 + copysign(atan|x| / pi, x).  No selects:
// atan(a) = pi/4 + atan((a-1)/(a+1)); with y ~ (a-1)/(a+1) rounded to c = j/128,
// atan(y) = atan(c) + atan(z), z = (n - c d)/(d + c n), n = a-1, d = a+1: ONE accurate
// reciprocal (of d + c n) and one raw one (to pick c).  |z| <= 1/256.
__device__ __forceinline__ double heaviside_fast(double x, const double *tab /*LDS, CVH_ATAN2_N*/)
{
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
  return 0.5 + __builtin_copysign(atpi, x);
}

__device__ __forceinline__ double dpp_from_left(double v)
{
  const long long vb = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_mov_dpp((int)vb, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(vb >> 32), 0x138, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int C, bool FAST, bool LUT, int MINW>
__global__ __launch_bounds__(CVH_BLOCK, MINW) void csv_wave_kernel(const CvhStepArgs a)
{
  using L = WaveSmem<C, FAST, LUT>;
  constexpr int NS = cvh_nsums(C);
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *sred = smem + L::off_red;
  double *sfin = smem + L::off_fin;
  double *satan = smem + L::off_atan;
  double *slut = smem + L::off_lut;
  int *s_last = (int *)(smem + L::off_flag);

  if (a.st->stopped) return;  // sticky stop: src/main.cpp:1000

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: row arithmetic stays scalar
  double *xs = smem + L::off_x + wave * (XSLOTS * XPITCH);
  const int h = a.h, w = a.w;
  if (tid == 0) *s_last = 0;

  double c1[C], c2[C], l1[C], l2[C];
#pragma unroll
  for (int k = 0; k < C; ++k) { c1[k] = a.st->c1[k]; c2[k] = a.st->c2[k]; l1[k] = a.lambda1[k]; l2[k] = a.lambda2[k]; }
  const double eps = a.eps;
  const double eps2 = eps * eps;

  if (FAST) {
    for (int q = tid; q < CVH_ATAN2_N; q += CVH_BLOCK) satan[q] = a.atan2_tab[q];
  }
  if (LUT) {
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const double v = (double)tid;
      const double d1 = v - c1[k], d2 = v - c2[k];
      const double reg = (d2 * d2) * l2[k] - (d1 * d1) * l1[k];
      slut[k * 256 + tid] = (k == 0) ? __builtin_fma(reg, a.beta, a.gamma) : reg * a.beta;
    }
  }
  __syncthreads();

  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0;

  // ---- this wave's strip
  const int nwc = a.tiles_x;  // wave-columns per image row
  const int gw = blockIdx.x * 4 + wave;
  const int wc = gw % nwc, ws = gw / nwc;
  const int s0 = ws * a.strip_rows;
  if (s0 < h) {
    const int s1 = (s0 + a.strip_rows) < h ? (s0 + a.strip_rows) : h;
    const int col = WCOLS * wc - 1 + lane;              // lane 0 = left halo column
    const int colc = clampi(col, 0, w - 1);
    const bool lane_valid = (lane >= 1) && (col < w);
    const double vm = lane_valid ? 1.0 : 0.0;           // sums of halo / out-of-image lanes are exact zeros
    const double fx = (col <= 0) ? 0.0 : 1.0;           // kappa_x(i,0) = 0 (:371)
    // the two extra halo columns of 4 consecutive rows, fetched by lanes 0..7
    const int xrow = (lane >> 1) & 3, xside = lane & 1;
    const int xcol = xside ? clampi(WCOLS * wc + 63, 0, w - 1) : clampi(WCOLS * wc - 2, 0, w - 1);
    const bool xlane = lane < 8;
    double *x_own = xs + 1 + lane;
    double *x_ext = xs + xrow * XPITCH + (xside ? 65 : 0);
    const double *x_w = xs + lane, *x_e = xs + lane + 2;

    // row base pointers are wave-uniform (scalar); the lane contributes a constant 32-bit offset
    auto U = [&](int r) -> double { const double *rp = a.u_in + (size_t)clampi(r, 0, h - 1) * w; return rp[colc]; };
    auto UX = [&](int r0) -> double { return a.u_in[(size_t)clampi(r0 + xrow, 0, h - 1) * w + xcol]; };
    auto IM = [&](int k, int r) -> int { const uint8_t *rp = a.img[k] + (size_t)clampi(r, 0, h - 1) * w; return rp[colc]; };

    // ---- prologue
    const double um2 = U(s0 - 2);
    double um = U(s0 - 1), u0 = U(s0);
    double q[4];
    int im[C][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      q[k] = U(s0 + 1 + k);
#pragma unroll
      for (int ch = 0; ch < C; ++ch) im[ch][k] = IM(ch, s0 + k);
    }
    double xq = xlane ? UX(s0) : 0.0;
    if (xlane) *x_ext = xq;                     // extras of rows s0 .. s0+3 -> slots 0..3
    x_own[0 * XPITCH] = u0;                     // row s0 -> slot 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double uw = x_w[0 * XPITCH], ue = x_e[0 * XPITCH];
    double ny_prev = normalised<FAST>(u0 - um, central(um2, u0));  // ny at row s0-1

    for (int ib = s0; ib < s1; ib += 4) {
      xq = xlane ? UX(ib + 4) : 0.0;            // extras of the NEXT four rows
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = ib + k;
        const double up = q[k];
        // row i+1: publish this lane's value, fetch its neighbours for the next step
        if (k == 3) { if (xlane) *x_ext = xq; }
        x_own[((k + 1) & 3) * XPITCH] = up;
        // the row buffer is exchanged between LANES of this wave: LDS operations of one wave
        // execute in order, the fences only stop the compiler from reordering them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double uw_n = x_w[((k + 1) & 3) * XPITCH], ue_n = x_e[((k + 1) & 3) * XPITCH];
        if (i < s1) {
          double nx, ny;
          if (FAST) {
            nx = normalised<true>(ue - u0, 0.5 * (ue - uw));
            ny = normalised<true>(up - u0, 0.5 * (up - um));
          } else {
            nx = normalised<false>(ue - u0, central(uw, ue));  // :365-366
            ny = normalised<false>(up - u0, central(um, up));  // :367-368
          }
          const double nxl = dpp_from_left(nx);
          double kappa;
          if (FAST) {
            kappa = __builtin_fma(nx - nxl, fx, (i == 0) ? 0.0 : ny - ny_prev);
          } else {
            const double kx = (col <= 0) ? 0.0 : nx - nxl;                // :371
            const double ky = (i == 0) ? 0.0 : ny - ny_prev;              // :372
            kappa = kx + ky;                                              // :373
          }
          double Ik[C];
#pragma unroll
          for (int ch = 0; ch < C; ++ch) Ik[ch] = (double)im[ch][k];
          double ud, hv;
          if (FAST) {
            double reg;
            if (LUT) {
              reg = slut[im[0][k]];
#pragma unroll
              for (int ch = 1; ch < C; ++ch) reg += slut[ch * 256 + im[ch][k]];
            } else {
              reg = 0.0;
#pragma unroll
              for (int ch = 0; ch < C; ++ch) {
                const double d1 = Ik[ch] - c1[ch], d2 = Ik[ch] - c2[ch];
                reg += (d2 * d2) * l2[ch] - (d1 * d1) * l1[ch];
              }
              reg = __builtin_fma(reg, a.beta, a.gamma);
            }
            ud = __builtin_fma(kappa, a.alpha, reg);                      // :985
            const double qd = __builtin_fma(u0, u0, eps2) * a.dk1;        // 1/delta_eps(u) = (pi/eps)(eps^2 + u^2)
            const double r0 = __builtin_amdgcn_rcp(qd);
            const double e = __builtin_fma(-qd, r0, 1.0);
            ud = ud * __builtin_fma(__builtin_fma(e, e, e), r0, r0);      // :992
          } else {
            ud = 0.0;  // :965
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
              const double d1 = Ik[ch] - c1[ch], d2 = Ik[ch] - c2[ch];
              const double vin = (d1 * d1) * l1[ch];   // :307-310
              const double vout = (d2 * d2) * l2[ch];
              ud += vout - vin;                         // :979
            }
            ud = kappa * a.alpha + ud * a.beta + a.gamma;   // :985
            ud = ud * (eps / (kPi * (eps2 + u0 * u0)));      // :209, :992
          }
          const double un = u0 + ud;                         // :994
          if (FAST) hv = heaviside_fast(un * a.inv_eps, satan);
          else hv = heaviside_strict(un, eps);
          if (lane_valid) { double *op = a.u_out + (size_t)i * w; op[colc] = un; }
          const double hz = hv * vm, udz = ud * vm;
          acc[0] += hz;
          if (!FAST) acc[1] += (1 - hv) * vm;
#pragma unroll
          for (int ch = 0; ch < C; ++ch) {
            if (FAST) {
              acc[2 + ch] = __builtin_fma(Ik[ch], hz, acc[2 + ch]);
            } else {
              acc[2 + ch] += Ik[ch] * hz;          // :276
              acc[2 + C + ch] += Ik[ch] * ((1 - hv) * vm);
            }
          }
          if (FAST) acc[2 + 2 * C] = __builtin_fma(udz, udz, acc[2 + 2 * C]);
          else acc[2 + 2 * C] += udz * udz;      // :993
          ny_prev = ny;
        }
        // refill the pipeline: row i+5 of u, row i+4 of the image
        q[k] = U(i + 5);
#pragma unroll
        for (int ch = 0; ch < C; ++ch) im[ch][k] = IM(ch, i + 4);
        um = u0; u0 = up; uw = uw_n; ue = ue_n;
      }
    }
  }

  const double total = block_reduce<NS>(acc, sred);
  publish_partials_and_maybe_finalize<C>(a, total, sred, sfin, s_last, gridDim.x);
}

template <int C, bool FAST, bool LUT, int MINW>
hipError_t launch_wave_v(const CvhStepArgs &a, hipStream_t s)
{
  using L = WaveSmem<C, FAST, LUT>;
  static_assert(L::bytes <= 64 * 1024, "dynamic LDS above 64 KiB would need hipFuncSetAttribute");
  hipLaunchKernelGGL((csv_wave_kernel<C, FAST, LUT, MINW>), dim3(a.nparts), dim3(CVH_BLOCK), L::bytes, s, a);
  return hipGetLastError();
}

template <int C>
hipError_t launch_wave_c(const CvhStepArgs &a, int fast, hipStream_t s)
{
  if (!fast) return launch_wave_v<C, false, false, 4>(a, s);
  if (a.wave_minw >= 8) return a.use_lut ? launch_wave_v<C, true, true, 8>(a, s) : launch_wave_v<C, true, false, 8>(a, s);
  if (a.wave_minw == 7) return a.use_lut ? launch_wave_v<C, true, true, 7>(a, s) : launch_wave_v<C, true, false, 7>(a, s);
  if (a.wave_minw == 6) return a.use_lut ? launch_wave_v<C, true, true, 6>(a, s) : launch_wave_v<C, true, false, 6>(a, s);
  return a.use_lut ? launch_wave_v<C, true, true, 5>(a, s) : launch_wave_v<C, true, false, 5>(a, s);
}

}  // namespace

int cvh_wave_cols() { return WCOLS; }

hipError_t cvh_launch_wave(const CvhStepArgs &a, int channels, int fast, hipStream_t s)
{
  return channels == 1 ? launch_wave_c<1>(a, fast, s) : launch_wave_c<3>(a, fast, s);
}
