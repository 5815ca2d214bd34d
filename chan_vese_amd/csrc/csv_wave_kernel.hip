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
constexpr int IMGP = 80;     // bytes per row of the per-wave image tile (5 x 16-byte pieces)

template <int C, bool FAST, bool LUT>
struct WaveSmem {
  static constexpr int NS = cvh_nsums(C);
  static constexpr int off_x = 0;                                          // 4 waves x XSLOTS x XPITCH
  static constexpr int wave_doubles = XSLOTS * XPITCH + 64 + C * 4 * IMGP / 8;  // row slots + scratch + image tile
  static constexpr int off_red = off_x + 4 * wave_doubles;                 // 4*NS
  static constexpr int off_fin = off_red + 4 * NS + (4 * NS) % 2;          // NS
  static constexpr int off_atan = off_fin + NS + NS % 2;                   // FAST: CVH_ATAN2_N
  static constexpr int off_lut = (off_atan + (FAST ? CVH_ATAN2_N + 1 : 0) + 1) & ~1;  // LUT: C*256 x {term, I}, 16-byte aligned
  static constexpr int off_flag = off_lut + (LUT ? C * 512 : 0);
  static constexpr int doubles = off_flag + 2;
  static constexpr size_t bytes = (size_t)doubles * sizeof(double);
};

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
struct FarCoef { double k0, k1, k2, k3, thr; };

// H_eps(u) - 1/2 = copysign(atan(|u|/eps)/pi, u).  The sums carry this CENTRED value (the
// finalisation adds N/2 and sum(I)/2, exact integers or half-integers): one addition less per pixel.
__device__ __forceinline__ double heaviside_centred_fast(double u, double inv_eps, const FarCoef &fc,
                                                         const double *tab /*LDS, CVH_ATAN2_N*/)
{
  // Far field, decided per WAVE (uniform branch): with the reference's default time step the
  // level set sits at |u| >> 64 eps everywhere after a handful of iterations.  There
  // atan(a) = pi/2 - atan(1/a), a = |u|/eps, and the series of atan(t), t = eps/|u| <= 1/64,
  // truncated after t^7/7 (next term < 7e-18) needs one reciprocal and no table.  In terms of
  // r = 1/u (signed): atan(eps r)/pi = r (k0 + r^2 (k1 + r^2 (k2 + r^2 k3))), k_i = (-1)^i eps^(2i+1)/((2i+1) pi).
  if (__builtin_amdgcn_ballot_w64(fabs(u) < fc.thr) == 0ull) {
    const double r0 = __builtin_amdgcn_rcp(u);
    const double r = __builtin_fma(__builtin_fma(-u, r0, 1.0), r0, r0);
    const double r2 = r * r;
    double p = fma3(r2, fc.k3, fc.k2);
    p = fma3(p, r2, fc.k1);
    p = fma3(p, r2, fc.k0);
    return __builtin_fma(-r, p, __builtin_copysign(0.5, u));
  }
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
  const double p = fma3(z2, 0.2, -1.0 / 3.0);
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

// Level-set rows and image pieces move through BUFFER instructions: the row base is a scalar
// offset (soffset), the lane's column a constant VGPR offset -- no per-row vector address
// arithmetic -- and a lane whose offset lies outside the buffer (>= num_records) is dropped
// by the hardware.  That gives a maskless, straight-line, compiler-visible store: hipcc counts
// loads AND stores in its vmcnt waits, so the 4-row load pipeline and the stores stay in flight.
// (An `if (lane_valid)` store is a control-flow diamond; a store hidden in inline assembly is not
// counted, and every counted wait then also drains the stores and the younger loads: measured,
// waves spent 50 % of their cycles in s_waitcnt -- profiles/README.md.)
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOffset = 0x80000000u;   // beyond any buffer this kernel accepts (< 2 GiB)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000 /* raw, 32-bit data format (gfx950) */);
}
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store_f64(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, voff, soff, 0);
}

__device__ __forceinline__ double dpp_from_left(double v)
{
  const long long vb = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_mov_dpp((int)vb, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(vb >> 32), 0x138, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

#ifndef CVH_STORE_MOD
#define CVH_STORE_MOD ""
#endif

template <int C, bool FAST, bool LUT, int MINW, bool IMGV, int G>
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
  const unsigned long long t_start = a.dbg_times ? __builtin_amdgcn_s_memrealtime() : 0ull;
  double *xs = smem + L::off_x + wave * L::wave_doubles;
  const int h = a.h, w = a.w;
  if (tid == 0) *s_last = 0;

  double c1[C], c2[C], l1[C], l2[C];
#pragma unroll
  for (int k = 0; k < C; ++k) { c1[k] = a.st->c1[k]; c2[k] = a.st->c2[k]; l1[k] = a.lambda1[k]; l2[k] = a.lambda2[k]; }
  const double eps = a.eps;
  const double eps2 = eps * eps;
  const FarCoef fc = {a.far_k[0], a.far_k[1], a.far_k[2], a.far_k[3], a.far_thr};

  // the tables are filled while the first rows are in flight: see fill_tables() below
  auto fill_tables = [&]() {
    if (FAST) {
      for (int q = tid; q < CVH_ATAN2_N; q += CVH_BLOCK) satan[q] = a.atan2_tab[q];
    }
    if (LUT) {
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const double v = (double)tid;
        const double d1 = v - c1[k], d2 = v - c2[k];
        const double reg = (d2 * d2) * l2[k] - (d1 * d1) * l1[k];
        slut[2 * (k * 256 + tid)] = (k == 0) ? __builtin_fma(reg, a.beta, a.gamma) : reg * a.beta;
        slut[2 * (k * 256 + tid) + 1] = v;   // the sample as a double rides along (saves the conversion)
      }
    }
  };

  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0;

  // ---- this wave's strip: workgroup = 4 adjacent wave-columns of one strip
  const int nwc = a.tiles_x;           // wave-columns per image row
  const int nbc = (nwc + 3) >> 2;      // workgroups per strip
  const int wc = (blockIdx.x % nbc) * 4 + wave, ws = blockIdx.x / nbc;
  const int s0 = a.strip_bounds[ws];
  const bool active = wc < nwc;        // the last workgroup of a strip may hold idle waves
  const int col = WCOLS * wc - 1 + lane;                // lane 0 = left halo column
  const bool lane_valid = active && (lane >= 1) && (col < w);
  if (active) {
    const int s1 = a.strip_bounds[ws + 1];
    const int colc = clampi(col, 0, w - 1);
    const double fx = (col <= 0) ? 0.0 : 1.0;           // kappa_x(i,0) = 0 (:371)
    // Every vector-memory operation below is issued by ALL lanes on EVERY row (halo / out-of-
    // image lanes are pointed at a dummy location instead of being masked off): the
    // instruction stream is straight-line, so the compiler's counted vmcnt waits let the
    // 4-row-deep load pipeline and the stores stay in flight.
    const bool xlane = lane < 8;
    const int xrow = xlane ? (lane >> 1) & 3 : 0, xside = lane & 1;
    const int xcol = !xlane ? colc : (xside ? clampi(WCOLS * wc + 63, 0, w - 1) : clampi(WCOLS * wc - 2, 0, w - 1));
    double *x_own = xs + 1 + lane;
    double *x_ext = xlane ? xs + xrow * XPITCH + (xside ? 65 : 0) : xs + XSLOTS * XPITCH + lane;  // lanes >= 8: scratch
    const double *x_w = xs + lane, *x_e = xs + lane + 2;
    const unsigned rowbytes = (unsigned)w * 8u, ubytes = (unsigned)h * rowbytes;   // < 2 GiB (launcher)
    const unsigned voff_u = (unsigned)colc * 8u;                  // byte offset of this lane's column in a row
    const unsigned voff_st = lane_valid ? voff_u : kOobOffset;    // lanes that own no output pixel store nowhere
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.u_in, ubytes);

    // row base pointers are wave-uniform (scalar); the lane contributes a constant 32-bit offset
#ifdef CVH_ABLATE_MEMORY   // diagnostic build: no global loads/stores in the loop (results are wrong)
    auto U = [&](int r) -> double { return (double)(r & 15) * 0.37 + (double)colc * 0.001; };
#else
    auto U = [&](int r) -> double { return buf_load_f64(ru, voff_u, (unsigned)clampi(r, 0, h - 1) * rowbytes); };
#endif
    const unsigned voff_x = ((unsigned)xrow * (unsigned)w + (unsigned)xcol) * 8u;
    auto UX = [&](int r0) -> double {   // r0 >= 0; away from the bottom edge the lane's offset is a constant
      if (r0 + 3 < h) return buf_load_f64(ru, voff_x, (unsigned)r0 * rowbytes);
      return buf_load_f64(ru, ((unsigned)clampi(r0 + xrow, 0, h - 1) * (unsigned)w + (unsigned)xcol) * 8u, 0u);
    };
    auto IM = [&](int k, int r) -> int { const uint8_t *rp = a.img[k] + (size_t)clampi(r, 0, h - 1) * w; return rp[colc]; };

    // ---- prologue
    // Rows of this lane's column live in a register ring.  G == 2: ring of 8, slot (t & 7) holds
    // row s0+1+t at step t (`up`), `u0` and `um` are slots t-1 and t-2, and the slot of the row
    // that has just died is refilled with the row 8 ahead: every loop-carried value keeps its
    // register from one iteration to the next, so the loop back-edge needs no copies of
    // in-flight loads (hipcc otherwise waits for them there and drains the pipeline every
    // iteration).  G == 1 keeps the shifted um/u0/up form.
    constexpr bool RING8 = (G == 2);
    const double um2 = U(s0 - 2);
    double um = U(s0 - 1), u0 = U(s0);
    double q[4 * G];
    int im[C][4];
    // Image samples.  IMGV (w % 16 == 0): the 64-byte row segments of 4 rows are fetched as
    // 20 aligned 16-byte pieces by ONE load (lanes 0..19), staged in a per-wave LDS tile and
    // read back as bytes: one vector-memory instruction per 4 rows instead of one 64 x 1-byte
    // load per row (measured: the byte loads alone cost ~18 us of a 4096^2 launch).
    unsigned char *simg = reinterpret_cast<unsigned char *>(xs + XSLOTS * XPITCH + 64);
    const int icol0 = (WCOLS * wc - 1) & ~15;                      // 16-byte aligned start column (may be < 0)
    const int ipiece = lane % 5, irow = lane / 5;                  // lanes 0..19: piece of row irow
    const bool ilane = lane < 20;
    int ipc = icol0 + 16 * ipiece;
    ipc = ipc < 0 ? 0 : (ipc > w - 16 ? w - 16 : ipc);             // clamped pieces only feed clamped columns
    const int ibyte = colc - icol0;                                // this lane's byte within a tile row (0..79)
    typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
    uint4_t iq[G][C];
    const unsigned voff_i = (unsigned)(ilane ? irow : 0) * (unsigned)w + (unsigned)ipc;
    auto IMQ = [&](int g, int r0) {
#pragma unroll
      for (int ch = 0; ch < C; ++ch)
      {
        const __amdgpu_buffer_rsrc_t ri = make_rsrc(a.img[ch], (unsigned)h * (unsigned)w);
        if (r0 + 3 < h) iq[g][ch] = __builtin_amdgcn_raw_buffer_load_b128(ri, voff_i, (unsigned)r0 * (unsigned)w, 0);
        else iq[g][ch] = __builtin_amdgcn_raw_buffer_load_b128(ri, (unsigned)clampi(r0 + (ilane ? irow : 0), 0, h - 1) * (unsigned)w + (unsigned)ipc, 0u, 0);
      }
    };
    auto IMTILE = [&](int g) {  // tile of the group whose pieces are in iq[g] -> im[][]
      if (ilane) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
          // a piece clamped at the image edge lands where its columns are expected
          const int dst = (icol0 + 16 * ipiece) == ipc ? 16 * ipiece : ipc - icol0;
          *reinterpret_cast<uint4_t *>(simg + (ch * 4 + irow) * IMGP + dst) = iq[g][ch];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int ch = 0; ch < C; ++ch)
#pragma unroll
        for (int k = 0; k < 4; ++k) im[ch][k] = simg[(ch * 4 + k) * IMGP + ibyte];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
#pragma unroll
    for (int k = 0; k < 4 * G; ++k) q[k] = U(s0 + 1 + k);
    if (RING8) { q[6] = um; q[7] = u0; }       // slots of rows s0-1, s0 (rows s0+7, s0+8 are requested at steps 0, 1)
    if (IMGV) {
#pragma unroll
      for (int g = 0; g < G; ++g) IMQ(g, s0 + 4 * g);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) im[ch][k] = IM(ch, s0 + k);
    }
    double xq[G];
    {
      const double x0 = UX(s0);
#pragma unroll
      for (int g = 1; g < G; ++g) xq[g] = UX(s0 + 4 * g);  // xq[g]: extras of rows base+4g .. (g >= 1)
      xq[0] = UX(s0 + 4 * G);                               // xq[0]: extras of the next iteration's first group
      fill_tables();                            // overlaps the prologue's loads
      __syncthreads();
      *x_ext = x0;                              // extras of rows s0 .. s0+3 -> slots 0..3
    }
    x_own[0 * XPITCH] = u0;                     // row s0 -> slot 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double uw = x_w[0 * XPITCH], ue = x_e[0 * XPITCH];
    double ny_prev = FAST ? normalised4(u0, um2, um + um) : normalised<false>(u0 - um, central(um2, u0));  // ny at row s0-1
    // kappa_y(0, .) = 0 (:372): on the image's first row ny_prev is set to that row's own ny,
    // computed with the very expression the row uses, so ny - ny_prev is exactly 0 there
    if (s0 == 0) ny_prev = FAST ? normalised4(q[0], um, u0 + u0) : normalised<false>(q[0] - u0, central(um, q[0]));

    // one row of the march; `live` (wave-uniform) is false only for rows past the strip end
    auto row = [&](int i, int g, int k, bool live) {
      const int t = 4 * g + k;                  // step within the loop body (static after unrolling)
      const double up = q[t];
      if (RING8) { u0 = q[(t + 7) & 7]; um = q[(t + 6) & 7]; }
      // row i+1: publish this lane's value, fetch its neighbours for the next step
      if (k == 3) *x_ext = xq[(g + 1) % G];     // extras of the next group's rows
      x_own[((k + 1) & 3) * XPITCH] = up;
      // the row buffer is exchanged between LANES of this wave: LDS operations of one wave
      // execute in order, the fences only stop the compiler from reordering them
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const double uw_n = x_w[((k + 1) & 3) * XPITCH], ue_n = x_e[((k + 1) & 3) * XPITCH];
#ifdef CVH_ABLATE_COMPUTE  // diagnostic build: the memory/LDS instruction stream without the arithmetic
      if (FAST) {
        const double un_ = u0 + (up + um + uw + ue) * 1e-30 + (double)im[0][k] * 1e-30;
        if (lane_valid) (live ? a.u_out + (size_t)i * w : a.dummy)[colc] = un_;
        acc[0] += un_;
        q[(t + 6) & 7] = U(i + 7);
        uw = uw_n; ue = ue_n;
        return;
      }
#endif
      double nx, ny;
      if (FAST) {
        const double u02 = u0 + u0;
        nx = normalised4(ue, uw, u02);
        ny = normalised4(up, um, u02);
      } else {
        nx = normalised<false>(ue - u0, central(uw, ue));  // :365-366
        ny = normalised<false>(up - u0, central(um, up));  // :367-368
      }
      const double nxl = dpp_from_left(nx);
      double kappa;
      if (FAST) {
        kappa = __builtin_fma(nx - nxl, fx, ny - ny_prev);
      } else {
        const double kx = (col <= 0) ? 0.0 : nx - nxl;                // :371
        const double ky = ny - ny_prev;                               // :372 (row 0: see ny_prev above)
        kappa = kx + ky;                                              // :373
      }
      double Ik[C];
      if (!(FAST && LUT)) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) Ik[ch] = (double)im[ch][k];
      }
      double ud, hv;
      if (FAST) {
        double reg;
        if (LUT) {
          typedef double double2_t __attribute__((ext_vector_type(2)));
          const double2_t *lut2 = reinterpret_cast<const double2_t *>(slut);
          const double2_t e0 = lut2[im[0][k]];
          reg = e0.x; Ik[0] = e0.y;
#pragma unroll
          for (int ch = 1; ch < C; ++ch) {
            const double2_t e = lut2[ch * 256 + im[ch][k]];
            reg += e.x; Ik[ch] = e.y;
          }
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
      if (FAST) hv = heaviside_centred_fast(un, a.inv_eps, fc, satan);   // H - 1/2: see finalize()
      else hv = heaviside_strict(un, eps);
      // rows past the strip end (wave-uniform) get an empty buffer: every lane is out of range
      buf_store_f64(un, make_rsrc(live ? a.u_out : a.dummy, live ? ubytes : 0u), voff_st, (unsigned)i * rowbytes);
      if (live) {  // halo / out-of-image lanes are zeroed once after the loop
        acc[0] += hv;
        if (!FAST) acc[1] += (1 - hv);
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
          if (FAST) {
            acc[2 + ch] = __builtin_fma(Ik[ch], hv, acc[2 + ch]);
          } else {
            acc[2 + ch] += Ik[ch] * hv;          // :276
            acc[2 + C + ch] += Ik[ch] * (1 - hv);
          }
        }
        if (FAST) acc[2 + 2 * C] = __builtin_fma(ud, ud, acc[2 + 2 * C]);
        else acc[2 + 2 * C] += ud * ud;        // :993
        ny_prev = ny;
      }
      // refill the pipeline: row i+5 of u, row i+4 of the image
      if (RING8) q[(t + 6) & 7] = U(i + 7);     // row i-1 is dead: its slot takes the row 8 below it
      else q[t] = U(i + 1 + 4 * G);
      if (!IMGV) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) im[ch][k] = IM(ch, i + 4);
      }
      if (!RING8) { um = u0; u0 = up; }
      uw = uw_n; ue = ue_n;
    };

    int prio = 3;
    if (a.wave_prio) __builtin_amdgcn_s_setprio(3);
    for (int ib0 = s0; ib0 < s1; ib0 += 4 * G) {
#pragma unroll
     for (int g = 0; g < G; ++g) {
      const int ib = ib0 + 4 * g;
      if (ib >= s1 && !a.wave_sync) break;
      if (a.wave_sync) __builtin_amdgcn_s_barrier();
      if (ib >= s1) continue;
      if (a.wave_prio) {
        // Equal-work waves drift apart under oldest-first issue arbitration and the tail then
        // runs at 1-2 waves per SIMD.  Waves that are AHEAD lower their priority (by quarter of
        // the strip), so laggards catch up and all waves finish together.
        const int rem = s1 - ib, len = s1 - s0;
        int pq;
        if (a.wave_prio == 1) pq = (rem * 4 - 1) / len;  // 3,2,1,0 by quarters of the strip
        else {  // thresholds crowd towards the end: only the last level's length sets the finishing spread
          const int sh = a.wave_prio == 4 ? 1 : a.wave_prio;  // 2: 1/4,1/8,1/16 of the strip left; 3: 1/2,1/4,1/8; 4: 1/8,1/16,1/32
          pq = (rem << (4 - sh)) > len ? 3 : ((rem << (5 - sh)) > len ? 2 : ((rem << (6 - sh)) > len ? 1 : 0));
        }
        if (pq != prio) {
          prio = pq;
          if (pq >= 3) __builtin_amdgcn_s_setprio(3);
          else if (pq == 2) __builtin_amdgcn_s_setprio(2);
          else if (pq == 1) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
      }
      if (IMGV) {
        IMTILE(g);                              // image bytes of THIS group (requested G groups ago)
        IMQ(g, ib + 4 * G);                     // request the group 4G rows ahead
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) row(ib + k, g, k, (ib + k) < s1);
      // extras 4G rows ahead of the group that follows (consumed at its k == 3 ... one turn later)
      xq[(g + 1) % G] = UX(ib + 4 + 4 * G);
     }
    }
    // exact: valid lanes are multiplied by 1, halo / out-of-image lanes by 0
    const double vmask = lane_valid ? 1.0 : 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = acc[s] * vmask;
  } else {
    fill_tables();
    __syncthreads();
  }
  if (!active && a.wave_sync) {                // idle waves still meet the per-iteration barrier
    const int s1i = a.strip_bounds[ws + 1];
    for (int ib = s0; ib < s1i; ib += 4) __builtin_amdgcn_s_barrier();
  }

  if (a.dbg_times && lane == 0) {  // diagnostic stamps: only ever written to their own buffer
    unsigned long long *d = a.dbg_times + (size_t)(blockIdx.x * 4 + wave) * 4;
    d[0] = t_start;
    d[1] = __builtin_amdgcn_s_memrealtime();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    d[2] = hwid;
    d[3] = xcc;
  }
  const double total = block_reduce<NS>(acc, sred);
  publish_partials_and_maybe_finalize<C>(a, total, sred, sfin, s_last, gridDim.x);
  if (a.dbg_times && tid == 0) a.dbg_times[(size_t)gridDim.x * 16 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

template <int C, bool FAST, bool LUT, int MINW>
hipError_t launch_wave_v(const CvhStepArgs &a, hipStream_t s)
{
  using L = WaveSmem<C, FAST, LUT>;
  static_assert(L::bytes <= 64 * 1024, "dynamic LDS above 64 KiB would need hipFuncSetAttribute");
  // Ask for 1/W of the CU's 160 KiB of LDS: the hardware can then place at most W workgroups on
  // a CU, so a grid of <= W x CUs workgroups spreads evenly instead of packing some CUs fuller
  // than others (the step is VALU-bound: the fullest CU sets the kernel time).
  size_t lds = L::bytes;
  if (a.wave_lds_cap) {
    size_t cap = ((size_t)160 * 1024 / (size_t)(a.wave_minw > 2 ? a.wave_minw : 3)) & ~(size_t)511;
    if (cap > 64 * 1024) cap = 64 * 1024;
    if (cap > lds) lds = cap;
  }
  const bool imgv = a.w % 16 == 0 && a.w >= 80 && a.wave_imgv;
  if (a.wave_depth >= 8) {
    if (imgv) hipLaunchKernelGGL((csv_wave_kernel<C, FAST, LUT, MINW, true, 2>), dim3(a.nparts), dim3(CVH_BLOCK), lds, s, a);
    else hipLaunchKernelGGL((csv_wave_kernel<C, FAST, LUT, MINW, false, 2>), dim3(a.nparts), dim3(CVH_BLOCK), lds, s, a);
  } else {
    if (imgv) hipLaunchKernelGGL((csv_wave_kernel<C, FAST, LUT, MINW, true, 1>), dim3(a.nparts), dim3(CVH_BLOCK), lds, s, a);
    else hipLaunchKernelGGL((csv_wave_kernel<C, FAST, LUT, MINW, false, 1>), dim3(a.nparts), dim3(CVH_BLOCK), lds, s, a);
  }
  return hipGetLastError();
}

template <int C>
hipError_t launch_wave_c(const CvhStepArgs &a, int fast, hipStream_t s)
{
  if (!fast) return launch_wave_v<C, false, false, (C == 1 ? 4 : 2)>(a, s);
  if constexpr (C == 3) {  // 9 accumulators, 3 image tiles: fits 168 registers (3 waves/SIMD) without spilling
    return a.use_lut ? launch_wave_v<C, true, true, 3>(a, s) : launch_wave_v<C, true, false, 3>(a, s);
  } else {
  if (a.wave_minw >= 8) return a.use_lut ? launch_wave_v<C, true, true, 8>(a, s) : launch_wave_v<C, true, false, 8>(a, s);
  if (a.wave_minw == 7) return a.use_lut ? launch_wave_v<C, true, true, 7>(a, s) : launch_wave_v<C, true, false, 7>(a, s);
  if (a.wave_minw == 6) return a.use_lut ? launch_wave_v<C, true, true, 6>(a, s) : launch_wave_v<C, true, false, 6>(a, s);
  if (a.wave_minw == 5) return a.use_lut ? launch_wave_v<C, true, true, 5>(a, s) : launch_wave_v<C, true, false, 5>(a, s);
  return a.use_lut ? launch_wave_v<C, true, true, 4>(a, s) : launch_wave_v<C, true, false, 4>(a, s);
  }
}

}  // namespace

int cvh_wave_cols() { return WCOLS; }

hipError_t cvh_launch_wave(const CvhStepArgs &a, int channels, int fast, hipStream_t s)
{
  return channels == 1 ? launch_wave_c<1>(a, fast, s) : launch_wave_c<3>(a, fast, s);
}
