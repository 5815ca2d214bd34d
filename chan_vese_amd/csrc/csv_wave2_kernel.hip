// csv_wave2_kernel.hip — wave-streaming CSV step with TWO pixels per lane (gfx950, wave64, 1 channel).
//
// Same arithmetic and the same data flow as csv_wave_kernel.hip (buffer loads/stores with dropped
// lanes, next group parked in a per-wave LDS ring, branch-free interior groups), but a lane owns two
// ADJACENT columns: a wave covers 128 columns (lane 0 holds the two west halo columns, lanes 1..63
// produce 126 pixels per row).  Measured on MI355X: one vector-memory INSTRUCTION per
// wave-row costs about as much whatever it moves, so 16-byte row loads/stores halve that cost per
// pixel; the LDS exchange and the DPP move of the west neighbour's normalised gradient are halved
// as well (one of the two x-neighbours of a pixel is in the same lane), the east halo is one column,
// and the two pixels of a lane are independent dependency chains.
//   ring slot (130 doubles): [a0 b0 a1 b1 ... a63 b63 | east extra]; per-lane read addresses implement
//   the BORDER_REPLICATE clamps at the image's left / right edges, so the row values need no selects.
// Workgroup = 2 wave-columns x 2 strips (a 4096-wide image has 33 wave-columns: pairs waste less than
// quads).  Requires w % 16 == 0 and w >= 144 (16-byte image pieces); other shapes use kernel 2.
// C = 3 (round 3, FAST only): one image tile per channel (two piece loads per group), the samples are read from the
// tile inside the row (no per-group sample registers), and the region term sum_k [l2k (I_k - c2k)^2 - l1k (I_k - c1k)^2] beta
// + gamma is three table lookups.  (The table-free quadratic form sum_k (qa_k I_k + qb_k) I_k + qc of round 3 -- 81 against 72 us --
// lives in tools/experiments/pruned_flavours/.)
#include "csv_device.h"
#include "buffer_ops.h"
#include "wave_math.h"
#include "chain_device.h"
#include "wave2_device.h"
#include <type_traits>

using namespace cvh_dev;

namespace {

template <int C, bool FAST, int MINW, int POL, bool ST32>
__global__ __launch_bounds__(CVH_BLOCK, MINW) void csv_wave2_kernel(const CvhStepArgs a)
{
  static_assert(C == 1 || FAST, "the 3-channel flavour exists in FAST arithmetic only (STRICT: kernel 2)");
  static_assert(!ST32 || FAST, "the declared FP32-state mode exists in FAST arithmetic only");
  using IO = StateIO<ST32, POL>;             // the level set's format in HBM (wave2_device.h): FP64, or the declared FP32 state
  static_assert(FAST ? MINW == 3 : MINW == 2, "compiled for 3 waves per SIMD (FAST: branch-free rows) or 2 (STRICT)");
  using L = Wave2Smem<FAST, C>;
  constexpr int NS = cvh_nsums(C), R = R2;
  constexpr int NIQ = (9 * R * C + 63) / 64;   // image piece loads per group (9 pieces x R rows x C channels, one lane each)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *sred = smem + L::off_red;
  double *sfin = smem + L::off_fin;
  double *satan = smem + L::off_atan;
  double *slut = smem + L::off_lut;
  int *s_last = (int *)(smem + L::off_flag);
  constexpr unsigned kLutAddr = (unsigned)(L::off_lut * sizeof(double));   // LDS byte address of the region-term table (the dynamic block starts at 0)
  if (!lds_base_is_zero(smem)) __builtin_trap();                          // (folds away: no static LDS in this kernel)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool chain = FAST && a.chain != nullptr;
  const int h = a.h, w = a.w;

  // ---- this wave's strip: workgroup = 2 adjacent wave-columns x 2 adjacent strips
  const int nwc = a.tiles_x, nstrips = a.tiles_y;
  const int nbc = (nwc + 1) >> 1;
  int bid = (int)blockIdx.x;
  const bool bookkeeper = bid >= a.nparts;   // chain mode: one extra workgroup per launch (chain_bookkeeping)
  if (a.wave_xcd && !bookkeeper) {
    const int nb = a.nparts, x = bid & 7, j = bid >> 3, q = nb >> 3, r = nb & 7;
    if (a.wave_cls > 0) {
      // class-major numbering: all workgroups of dispatch round 0 (the first wave_cls of every XCD: one per CU, the
      // OLDEST wave of their SIMD), then round 1, ...; inside a round XCD by XCD, so an XCD still works on contiguous
      // wave-columns / strips.  The host gives the rounds different strip lengths (api.hip, upload_strip_bounds).
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
    } else {   // XCD-contiguous numbering (see csv_wave_kernel.hip)
      bid = x * q + (x < r ? x : r) + j;
    }
  }
  const int wc = (bid % nbc) * 2 + (wave & 1);
  const int ws = (bid / nbc) * 2 + (wave >> 1);
  const bool active = !bookkeeper && wc < nwc && ws < nstrips;
  // One batch of scalar loads before anything else: the sticky stop flag (src/main.cpp:1000) and the rows of this
  // workgroup's two strips (the exit test reads all of them, so hipcc issues them together: one round trip).
  const int wsa = bookkeeper ? 0 : (bid / nbc) * 2;
  const const_int_p sb = (const_int_p)a.strip_bounds;
  const int stopped = *(const_int_p)&a.st->stopped;
  const int b0 = sb[wsa], b1 = sb[wsa + 1 <= nstrips ? wsa + 1 : nstrips], b2 = sb[wsa + 2 <= nstrips ? wsa + 2 : nstrips];
  if ((stopped != 0) | (b1 < b0) | (b2 < b1)) return;

  const unsigned long long t_start = a.dbg_times ? __builtin_amdgcn_s_memrealtime() : 0ull;
  double *xs = smem + L::off_x + wave * L::wave_doubles;
  if (tid == 0) *s_last = 0;

  // region means of the level set this launch reads: from the fixed-point sum set (chain mode: one 8-byte load per
  // lane, in flight beside the first rows of u) or from the state block the last finaliser wrote
  double c1, c2;
  double cm1[C], cm2[C];                    // C = 3: the region means per channel
  long long chain_entry = 0;
  if (chain) chain_entry = a.chain->v[a.chain_phase][lane];
  else if (C == 1) { c1 = a.st->c1[0]; c2 = a.st->c2[0]; }
  else {
#pragma unroll
    for (int k = 0; k < C; ++k) { cm1[k] = a.st->c1[k]; cm2[k] = a.st->c2[k]; }
  }
  if (bookkeeper) { chain_bookkeeper_block<C>(a, chain_entry, sred); return; }
  const double l1 = a.lambda1[0], l2 = a.lambda2[0];
  const double eps = a.eps, eps2 = eps * eps;
  const FarCoef fc = {a.far_k[0], a.far_k[1], a.far_k[2], a.far_k[3], a.far_k[4], a.far_thr};

  auto fill_tables = [&]() {
    if (C == 1) {
      if (chain) { double m1[1], m2[1]; chain_means<1>(a, chain_entry, m1, m2); c1 = m1[0]; c2 = m2[0]; }
      if (FAST) {
        for (int q = tid; q < CVH_ATAN2_N; q += CVH_BLOCK) satan[q] = a.atan2_tab[q];
        const double v = (double)tid;
        const double d1 = v - c1, d2 = v - c2;
        const double reg = (d2 * d2) * l2 - (d1 * d1) * l1;
        slut[2 * tid] = __builtin_fma(reg, a.beta, a.gamma);
        slut[2 * tid + 1] = v;
      }
    } else {
      if (chain) chain_means<C>(a, chain_entry, cm1, cm2);
      for (int q = tid; q < CVH_ATAN2_N; q += CVH_BLOCK) satan[q] = a.atan2_tab[q];
      {
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const double v = (double)tid;
          const double d1 = v - cm1[k], d2 = v - cm2[k];
          const double reg = (d2 * d2) * a.lambda2[k] - (d1 * d1) * a.lambda1[k];
          slut[2 * (k * 256 + tid)] = (k == 0) ? __builtin_fma(reg, a.beta, a.gamma) : reg * a.beta;
          slut[2 * (k * 256 + tid) + 1] = v;   // the sample as a double rides along (saves the conversion)
        }
      }
    }
  };

  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0;

  // the 4 waves of the workgroup meet at one barrier per group: all run the longer strip's group count
  int wg_groups = 0;
  {
    const int la = b1 - b0, lb = b2 - b1;
    wg_groups = ((la > lb ? la : lb) + R - 1) / R;
  }
  int groups_done = 0;

  if (active) {
    const int s0 = (wave >> 1) ? b1 : b0, s1 = (wave >> 1) ? b2 : b1;
    const int c0 = W2 * wc - 2 + 2 * lane;                 // column of pixel `a`; pixel `b` is c0 + 1
    const bool lane_valid = lane >= 1 && c0 < w;           // w is even: both pixels or none
    const int cl = c0 < 0 ? 0 : (c0 > w - 2 ? w - 2 : c0); // even column of the 16-byte load
    const double fxa = (c0 <= 0) ? 0.0 : 1.0;              // kappa_x(i,0) = 0 (:371); pixel b never is column 0
    // per-lane ring indices: own pair, west neighbour of a, east neighbour of b (replicated at the image edges)
    const int pa = 2 * lane;
    const int pw = (lane == 0 || c0 <= 0) ? pa : pa - 1;   // lane 0 (halo) never uses its west value
    const int pe = (c0 + 2 >= w) ? pa + 1 : pa + 2;        // lane 63 of a full wave: index 128 = the east extra
    const double2_t *x_own = reinterpret_cast<const double2_t *>(xs + pa);
    const double *x_w = xs + pw, *x_e = xs + pe;
    double2_t *x_put = reinterpret_cast<double2_t *>(xs + pa);
    // east extra: column 126 wc + 126 of R rows, one lane per row
    const bool xlane = lane < R;
    const int xrow = xlane ? lane : 0;
    const int xcol = clampi(W2 * wc + W2, 0, w - 1);
    double *x_ext = xlane ? xs + xrow * XP2 + 128 : xs + R * XP2 + lane;   // other lanes: scratch
    using raw2_t = typename IO::raw2_t;
    using raw1_t = typename IO::raw1_t;
    constexpr unsigned SB = IO::kBytes;                          // bytes of a level-set value in HBM
    const unsigned rowbytes = (unsigned)w * SB, ubytes = (unsigned)h * rowbytes;
    const unsigned voff_u = (unsigned)cl * SB;
    const unsigned voff_st = lane_valid ? (unsigned)c0 * SB : kOobOffset;
    const unsigned voff_x = ((unsigned)xrow * (unsigned)w + (unsigned)xcol) * SB;
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.u_in, ubytes);
    const int ulast = s1 < h - 1 ? s1 : h - 1, ilast = s1 - 1;
    auto U = [&](int r) -> raw2_t { return IO::load2(ru, voff_u, (unsigned)clampi(r, 0, ulast) * rowbytes); };
    auto UX = [&](int r0) -> raw1_t {
      if (r0 + R - 1 <= ulast) return IO::load1(ru, voff_x, (unsigned)r0 * rowbytes);
      return IO::load1(ru, ((unsigned)clampi(r0 + xrow, 0, ulast) * (unsigned)w + (unsigned)xcol) * SB, 0u);
    };
    // image: 9 aligned 16-byte pieces per row, R rows by 9R lanes, staged in the per-wave tile
    unsigned char *simg = reinterpret_cast<unsigned char *>(xs + R * XP2 + 64);
    const int icol0 = (W2 * wc - 2) & ~15;                      // may be < 0
    const int ipiece = lane % 9, irow = lane / 9;
    const bool ilane = lane < 9 * R;
    int ipc = icol0 + 16 * ipiece;
    ipc = ipc < 0 ? 0 : (ipc > w - 16 ? w - 16 : ipc);
    const unsigned voff_i = (unsigned)(ilane ? irow : 0) * (unsigned)w + (unsigned)ipc;
    unsigned char *ipiece_dst = simg + irow * IMGP2 + ((icol0 + 16 * ipiece) == ipc ? 16 * ipiece : ipc - icol0);
    const int ibyte = cl - icol0;                               // bytes of (a, b): ibyte, ibyte + 1
    const __amdgpu_buffer_rsrc_t ri = make_rsrc(a.img[0], (unsigned)(C - 1) * a.img_stride + (unsigned)h * (unsigned)w);
    auto IMQ = [&](int r0) -> u32x4_t {
      if (r0 + R - 1 <= ilast) return buf_load_b128(ri, voff_i, (unsigned)r0 * (unsigned)w);
      return buf_load_b128(ri, (unsigned)clampi(r0 + (ilane ? irow : 0), 0, ilast) * (unsigned)w + (unsigned)ipc, 0u);
    };
    // C = 3: piece q = 36 ch + 9 row + piece of the group's 108; load j fetches pieces 64 j + lane (the planes live in one slab)
    bool q_on[NIQ];
    int q_row[NIQ];
    unsigned q_voff[NIQ], q_voff_cl[NIQ];     // interior offset (row folded in); channel + column part for the clamped form
    unsigned char *q_dst[NIQ];
#pragma unroll
    for (int j = 0; j < NIQ; ++j) {
      const int q = 64 * j + lane, qc = q < 9 * R * C ? q : 0;
      const int ch = qc / (9 * R), il = qc % (9 * R), pr = il / 9, pp = il % 9;
      int pcq = icol0 + 16 * pp;
      pcq = pcq < 0 ? 0 : (pcq > w - 16 ? w - 16 : pcq);
      q_on[j] = q < 9 * R * C;
      q_row[j] = pr;
      q_voff_cl[j] = (unsigned)ch * a.img_stride + (unsigned)pcq;
      q_voff[j] = q_voff_cl[j] + (unsigned)pr * (unsigned)w;
      q_dst[j] = simg + (ch * R + pr) * IMGP2 + ((icol0 + 16 * pp) == pcq ? 16 * pp : pcq - icol0);
    }
    auto IMQ3 = [&](int r0, int j) -> u32x4_t {
      if (r0 + R - 1 <= ilast) return buf_load_b128(ri, q_voff[j], (unsigned)r0 * (unsigned)w);
      return buf_load_b128(ri, q_voff_cl[j] + (unsigned)clampi(r0 + q_row[j], 0, ilast) * (unsigned)w, 0u);
    };
    auto lds_fence = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    int im[R];   // the group's samples: byte of a | byte of b << 8   (C = 1; three channels read them inside the row)
    auto park = [&](const raw2_t (&T)[R], raw1_t X, const u32x4_t (&IQ)[NIQ]) {   // (the ring holds doubles whatever the format in HBM)
      lds_fence();
#pragma unroll
      for (int j = 0; j < R; ++j) x_put[j * (XP2 / 2)] = IO::widen(T[j]);
      *x_ext = IO::widen(X);
      if (C == 1) {
        if (ilane) *reinterpret_cast<u32x4_t *>(ipiece_dst) = IQ[0];
      } else {
#pragma unroll
        for (int j = 0; j < NIQ; ++j) if (q_on[j]) *reinterpret_cast<u32x4_t *>(q_dst[j]) = IQ[j];
      }
      lds_fence();
      if (C == 1) {
#pragma unroll
        for (int k = 0; k < R; ++k) im[k] = (int)*reinterpret_cast<const unsigned short *>(simg + k * IMGP2 + ibyte);
      }
    };
    // samples of row k of the group, channel ch: byte of a | byte of b << 8
    auto samples = [&](int ch, int k) -> int { return (int)*reinterpret_cast<const unsigned short *>(simg + (ch * R + k) * IMGP2 + ibyte); };

    // ---- prologue
    const double2_t um2 = IO::widen(U(s0 - 2));
    double2_t um = IO::widen(U(s0 - 1)), u0 = IO::widen(U(s0));
    double uw, ue;
    {
      raw2_t T[R];
#pragma unroll
      for (int j = 0; j < R; ++j) T[j] = U(s0 + 1 + j);
      const double X0 = IO::widen(UX(s0));
      const raw1_t X = UX(s0 + 1);
      u32x4_t IQ[NIQ];
      if (C == 1) IQ[0] = IMQ(s0);
      else {
#pragma unroll
        for (int j = 0; j < NIQ; ++j) IQ[j] = IMQ3(s0, j);
      }
      fill_tables();
      __syncthreads();
      x_put[0] = u0;
      if (xrow == 0) *x_ext = X0;
      lds_fence();
      uw = x_w[0]; ue = x_e[0];
      park(T, X, IQ);
    }
    auto norm = [&](double fwd, double bwd, double c) -> double {
      return FAST ? normalised4(fwd, bwd, c + c) : normalised<false>(fwd - c, central(bwd, fwd));
    };
    double nypa = norm(u0.x, um2.x, um.x), nypb = norm(u0.y, um2.y, um.y);   // ny at row s0-1
    if (s0 == 0) {   // kappa_y(0, .) = 0 (:372): ny_prev := row 0's own ny (see csv_wave_kernel.hip)
      const double2_t up0 = x_own[0];
      nypa = norm(up0.x, um.x, u0.x); nypb = norm(up0.y, um.y, u0.y);
    }

    // update of one pixel from its 4 neighbours, its x-gradient nx and the west neighbour's (nxl)
    auto pixel = [&](double c, double n_, double s_, double nx, double nxl, double fx, double &nyp, int byte,
                     double &ud_out, double &Ik_out) -> double {
      const double ny = norm(s_, n_, c);
      double kappa, ud, Ik;
      if (FAST) {
        kappa = __builtin_fma(nx - nxl, fx, ny - nyp);
        const double2_t e = lds_read_d2(kLutAddr + (unsigned)byte);   // FAST: `byte` is the entry's byte offset (sample x 16)
        Ik = e.y;
        ud = __builtin_fma(kappa, a.alpha, e.x);                       // :985
        const double qd = __builtin_fma(c, c, eps2) * a.dk1;           // 1/delta_eps(u)
        const double r0 = __builtin_amdgcn_rcp(qd);
        const double er = __builtin_fma(-qd, r0, 1.0);
        ud = ud * __builtin_fma(__builtin_fma(er, er, er), r0, r0);    // :992
      } else {
        const double kx = (fx == 0.0) ? 0.0 : nx - nxl;                // :371
        const double ky = ny - nyp;                                    // :372
        kappa = kx + ky;                                               // :373
        Ik = (double)byte;
        const double d1 = Ik - c1, d2 = Ik - c2;
        const double vin = (d1 * d1) * l1, vout = (d2 * d2) * l2;      // :307-310
        ud = 0.0;                                                      // :965
        ud += vout - vin;                                              // :979
        ud = kappa * a.alpha + ud * a.beta + a.gamma;                  // :985
        ud = ud * (eps / (kPi * (eps2 + c * c)));                      // :209, :992
      }
      nyp = ny;
      ud_out = ud; Ik_out = Ik;
      return c + ud;                                                   // :994
    };

    // three channels (FAST): the same update with the region term summed over the channels (:965-985)
    auto pixel3 = [&](double c, double n_, double s_, double nx, double nxl, double fx, double &nyp, const int (&byte)[C],
                      double &ud_out, double (&Ik)[C]) -> double {
      const double ny = norm(s_, n_, c);
      const double kappa = __builtin_fma(nx - nxl, fx, ny - nyp);
      double reg;
      {
        const double2_t e0 = lds_read_d2(kLutAddr + (unsigned)byte[0]);   // `byte[ch]` is the entry's byte offset inside channel ch's table (sample x 16)
        reg = e0.x; Ik[0] = e0.y;
#pragma unroll
        for (int ch = 1; ch < C; ++ch) {
          const double2_t e = lds_read_d2(kLutAddr + (unsigned)(ch * 4096) + (unsigned)byte[ch]);
          reg += e.x; Ik[ch] = e.y;
        }
      }
      double ud = __builtin_fma(kappa, a.alpha, reg);                  // :985
      const double qd = __builtin_fma(c, c, eps2) * a.dk1;             // 1/delta_eps(u)
      const double r0 = __builtin_amdgcn_rcp(qd);
      const double er = __builtin_fma(-qd, r0, 1.0);
      ud = ud * __builtin_fma(__builtin_fma(er, er, er), r0, r0);      // :992
      nyp = ny;
      ud_out = ud;
      return c + ud;                                                   // :994
    };

    // DEFER (FAST, 3 waves/SIMD: register room): rows without a branch, see csv_wave_kernel.hip
    constexpr bool DEFER = FAST;
    double2_t keep[R];
    // lanes that met a pixel below the far-field threshold, per row of the group.  (Measured, round 4, one process: ONE mask OR-ed up row by
    // row costs 3 us per 4096^2 launch -- the rows' dependent chains no longer overlap --, one running minimum of |u| per lane 0.7 us;
    // four independent masks, two of which hipcc parks in VGPRs, are the fastest form.)
    unsigned long long near_mask[R];
    int smp3[C][R];   // NEARFORM, three channels: the group's samples, taken aside before the park refills the image tile
    auto row = [&](int i, int k, bool live, auto near_tag) {
      constexpr bool NEARFORM = decltype(near_tag)::value;   // this group evaluates H_eps in its table form on every lane (below)
      const double2_t up = x_own[k * (XP2 / 2)];
      const double uw_n = x_w[k * XP2], ue_n = x_e[k * XP2];
      int sa[C], sb[C];                              // samples of pixel a / b per channel
      // FAST: sample x 16 = the byte offset of its 16-byte table entry, one SDWA instruction per sample (wave_math.h); STRICT: the sample
      if (C == 1) {
        if (FAST) { sa[0] = (int)byte_x16<0>((unsigned)im[k]); sb[0] = (int)byte_x16<1>((unsigned)im[k]); }
        else { sa[0] = im[k] & 0xff; sb[0] = (im[k] >> 8) & 0xff; }
      } else {
        // (reading the samples one row ahead, so that the table lookups wait for one LDS round trip instead of two: measured, no gain --
        // DESIGN.md 4.1, the 3-channel paragraph)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) { const unsigned s = (unsigned)samples(ch, k); sa[ch] = (int)byte_x16<0>(s); sb[ch] = (int)byte_x16<1>(s); }
      }
      const int ba = sa[0], bb = sb[0];
      // x-gradients first: nx(b) is the west gradient of lane+1's a (DPP), nx(a) the west gradient of b
      const double nxa = norm(u0.y, uw, u0.x);       // east = own b, west = lane-1's b
      const double nxb = norm(ue, u0.x, u0.y);       // east = lane+1's a, west = own a
      const double nxla = dpp_from_left(nxb);
      double uda, udb, Ia, Ib, nya = nypa, nyb = nypb;
      double Ika[C], Ikb[C];
      double va, vb;
      if (C == 1) {
        va = pixel(u0.x, um.x, up.x, nxa, nxla, fxa, nya, ba, uda, Ia);
        vb = pixel(u0.y, um.y, up.y, nxb, nxa, 1.0, nyb, bb, udb, Ib);
      } else {
        va = pixel3(u0.x, um.x, up.x, nxa, nxla, fxa, nya, sa, uda, Ika);
        vb = pixel3(u0.y, um.y, up.y, nxb, nxa, 1.0, nyb, sb, udb, Ikb);
      }
      double hva, hvb;
      // gfx950 wide-store data hazard (found in round 2, root-caused in round 3: tools/store_hazard_probe.hip): a VALU
      // instruction that writes a data register of a 16-byte buffer store in the issue slot right behind it changes what lanes
      // 12-15 of every row of 16 store under memory back-pressure, and hipcc pads that only for stores WITHOUT a register
      // soffset -- these stores have one (the scalar row offset).  One wait state is enough; LDS / vector-memory returns into
      // the registers are harmless.  The stored pair lives in keep[k] until the END of the group (the branch-free flavour needs
      // it there anyway, the others pin it below), so nothing writes it for hundreds of instructions, and
      // tools/isa_store_hazard.py checks the emitted ISA of every instantiation (tests/test_isa_hazard.py).
      if (ST32) { va = IO::stored(va); vb = IO::stored(vb); }   // FP32 state: what the next iteration will load is what H_eps is taken of
      keep[k] = double2_t{va, vb};
      if (FAST && DEFER && NEARFORM) {   // H_eps of the whole group is taken behind its rows, in the table form (group())
        hva = 0.0; hvb = 0.0;
        near_mask[k] = 0ull;
      } else if (FAST && DEFER) {   // far-field form on every lane; near lanes are corrected once per group (no branch in a row)
        hva = heaviside_centred_far(va, fc); hvb = heaviside_centred_far(vb, fc);
        near_mask[k] = __builtin_amdgcn_ballot_w64(fabs(va) < fc.thr || fabs(vb) < fc.thr);
      } else {
        hva = heaviside_strict(va, eps); hvb = heaviside_strict(vb, eps);
      }
      IO::store2(keep[k], make_rsrc(a.u_out, live ? ubytes : 0u), voff_st, (unsigned)i * rowbytes);
      if (live) {
        if (FAST && DEFER && NEARFORM) {   // the sums of H follow behind the rows
          acc[2 + 2 * C] = __builtin_fma(uda, uda, acc[2 + 2 * C]); acc[2 + 2 * C] = __builtin_fma(udb, udb, acc[2 + 2 * C]);
        } else if (FAST && C == 1) {
          acc[0] += hva; acc[0] += hvb;
          acc[2] = __builtin_fma(Ia, hva, acc[2]); acc[2] = __builtin_fma(Ib, hvb, acc[2]);
          acc[4] = __builtin_fma(uda, uda, acc[4]); acc[4] = __builtin_fma(udb, udb, acc[4]);
        } else if (FAST) {
          acc[0] += hva; acc[0] += hvb;
#pragma unroll
          for (int ch = 0; ch < C; ++ch) {
            acc[2 + ch] = __builtin_fma(Ika[ch], hva, acc[2 + ch]); acc[2 + ch] = __builtin_fma(Ikb[ch], hvb, acc[2 + ch]);
          }
          acc[2 + 2 * C] = __builtin_fma(uda, uda, acc[2 + 2 * C]); acc[2 + 2 * C] = __builtin_fma(udb, udb, acc[2 + 2 * C]);
        } else {
          acc[0] += hva; acc[1] += (1 - hva); acc[2] += Ia * hva; acc[3] += Ia * (1 - hva); acc[4] += uda * uda;
          acc[0] += hvb; acc[1] += (1 - hvb); acc[2] += Ib * hvb; acc[3] += Ib * (1 - hvb); acc[4] += udb * udb;
        }
        nypa = nya; nypb = nyb;
      }
      um = u0; u0 = up;
      uw = uw_n; ue = ue_n;
    };

    int prio = 3;
    if (a.wave_prio) __builtin_amdgcn_s_setprio(3);
    auto group = [&](int ib, auto interior_tag, auto near_tag) {
      constexpr bool INTERIOR = decltype(interior_tag)::value;
      constexpr bool NEARFORM = decltype(near_tag)::value;
      if (a.wave_sync) { __builtin_amdgcn_s_barrier(); ++groups_done; }
      if (a.wave_prio) {
        // At equal priority the SIMD arbiter serves its OLDEST wave first: the 4 waves of a SIMD then run almost
        // one after the other (measured: they finish 11 us apart, the last one alone on the SIMD).  Waves that
        // are AHEAD lower their priority, so all finish together and hide each other's latencies to the end.
        const int rem = s1 - ib, len = s1 - s0;
        int pq;
        if (a.wave_prio == 1) pq = (rem * 4 - 1) / len;  // 3,2,1,0 by quarters of the strip
        else {
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
      raw2_t T[R];
#pragma unroll
      for (int j = 0; j < R; ++j) T[j] = INTERIOR ? IO::load2(ru, voff_u, (unsigned)(ib + R + 1 + j) * rowbytes) : U(ib + R + 1 + j);
      const raw1_t X = INTERIOR ? IO::load1(ru, voff_x, (unsigned)(ib + R + 1) * rowbytes) : UX(ib + R + 1);
      u32x4_t IQ[NIQ];
      if (C == 1) IQ[0] = INTERIOR ? buf_load_b128(ri, voff_i, (unsigned)(ib + R) * (unsigned)w) : IMQ(ib + R);
      else {
#pragma unroll
        for (int j = 0; j < NIQ; ++j) IQ[j] = INTERIOR ? buf_load_b128(ri, q_voff[j], (unsigned)(ib + R) * (unsigned)w) : IMQ3(ib + R, j);
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (INTERIOR || (ib + k) < s1) row(ib + k, k, true, near_tag);   // wave-uniform: rows past the strip end cost nothing
        else near_mask[k] = 0ull;
      }
      const bool any_near = DEFER && !NEARFORM && (near_mask[0] | near_mask[1] | near_mask[2] | near_mask[3]) != 0ull;
      // H_eps - 1/2 of row k's pair from keep[k]: TABLE true = the table form of every pixel (the rows of a NEARFORM group added nothing for H),
      // else the per-group correction of the lanes below the far-field threshold (the rows added the far form on every lane)
      auto finish_row = [&](int k, int smp0, auto table_tag) {
        constexpr bool TABLE = decltype(table_tag)::value;
        const double xa = keep[k].x, xb = keep[k].y;
        double da, db;
        if (TABLE) {
          da = heaviside_centred_near(xa, a.inv_eps, satan); db = heaviside_centred_near(xb, a.inv_eps, satan);
        } else {
          da = (fabs(xa) < fc.thr) ? heaviside_centred_near(xa, a.inv_eps, satan) - heaviside_centred_far(xa, fc) : 0.0;
          db = (fabs(xb) < fc.thr) ? heaviside_centred_near(xb, a.inv_eps, satan) - heaviside_centred_far(xb, fc) : 0.0;
        }
        acc[0] += da; acc[0] += db;
        if (C == 1) {
          acc[2] = __builtin_fma((double)(smp0 & 0xff), da, acc[2]);
          acc[2] = __builtin_fma((double)((smp0 >> 8) & 0xff), db, acc[2]);
        } else {
#pragma unroll
          for (int ch = 0; ch < C; ++ch) {
            const int s = TABLE ? smp3[ch][k] : samples(ch, k);
            acc[2 + ch] = __builtin_fma((double)(s & 0xff), da, acc[2 + ch]);
            acc[2 + ch] = __builtin_fma((double)(s >> 8), db, acc[2 + ch]);
          }
        }
      };
      if (DEFER && !NEARFORM && any_near) {
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (near_mask[k] != 0ull && (INTERIOR || (ib + k) < s1)) finish_row(k, im[k], std::false_type{});
      }
      if (DEFER && NEARFORM) {
        // the table forms come BEHIND the park: the prefetched rows (22 registers) are in the ring by then, and the forms of eight pixels
        // have the registers to overlap.  The group's samples are taken aside first (the park refills im[] / the image tile).
        int smp1[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
          smp1[k] = (C == 1) ? im[k] : 0;
          if (C > 1) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) smp3[ch][k] = samples(ch, k);
          }
        }
        park(T, X, IQ);
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (INTERIOR || (ib + k) < s1) finish_row(k, smp1[k], std::true_type{});
#pragma unroll
        for (int k = 0; k < R; ++k) asm volatile("; row %2 of the group: store data still live" :: "v"(keep[k].x), "v"(keep[k].y), "n"(0));
      } else {
#pragma unroll
        for (int k = 0; k < R; ++k) asm volatile("; row %2 of the group: store data still live" :: "v"(keep[k].x), "v"(keep[k].y), "n"(0));
        park(T, X, IQ);
      }
    };
    const unsigned long long t_first = a.dbg_times ? __builtin_amdgcn_s_memrealtime() : 0ull;   // prologue done
    if (a.dbg_times && lane == 0) a.dbg_times[(size_t)(blockIdx.x * 4 + wave) * 4 + 2] = t_first;
    int ib = s0;
    // Which form of H_eps a STRIP takes is decided per wave from its first row: where most of the row's pixels are below the far-field
    // threshold (a level set that is near everywhere: dt << 1, the reference README's second example; the first iterations of a
    // checkerboard start) the wave runs the copy of the march that evaluates the table form on every lane, behind the rows of a group
    // -- valid for any u, nothing to correct: one form per pixel instead of three (far + near + far again in the correction).
    const bool near_strip = DEFER && a.near_switch &&
        __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane_valid && (fabs(u0.x) < fc.thr || fabs(u0.y) < fc.thr))) >= 32;
    if (__builtin_expect(DEFER && near_strip, 0)) {   // cold for the register allocator: whatever has to spill spills in this copy, not in the far-field march
      for (; ib + 2 * R <= ulast; ib += R) group(ib, std::true_type{}, std::true_type{});
      for (; ib < s1; ib += R) group(ib, std::false_type{}, std::true_type{});
    } else {
      for (; ib + 2 * R <= ulast; ib += R) group(ib, std::true_type{}, std::false_type{});
      for (; ib < s1; ib += R) group(ib, std::false_type{}, std::false_type{});
    }
    const double vmask = lane_valid ? 1.0 : 0.0;   // exact: halo / out-of-image lanes contribute nothing
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = acc[s] * vmask;
  } else {
    fill_tables();
    __syncthreads();
  }
  if (a.wave_sync) {   // remaining barriers of the workgroup (shorter strip, idle wave)
    for (; groups_done < wg_groups; ++groups_done) __builtin_amdgcn_s_barrier();
  }
  if (a.dbg_times && lane == 0) {  // diagnostic stamps (tools/wave_timeline.py): only ever written to their own buffer
    unsigned long long *d = a.dbg_times + (size_t)(blockIdx.x * 4 + wave) * 4;
    d[0] = t_start;
    d[1] = __builtin_amdgcn_s_memrealtime();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    d[3] = (unsigned long long)xcc | ((unsigned long long)hwid << 8) | ((unsigned long long)(unsigned)bid << 40);   // d[2] = time the first group started; bid = the workgroup's logical index
  }
  const double total = block_reduce<NS>(acc, sred);
  if (chain) {
    chain_publish<C>(a, total);   // fixed-point atomics + the sum u_diff^2 row: nothing waits (chain_device.h)
  } else {
    publish_partials_and_maybe_finalize<C>(a, total, sred, sfin, s_last, a.nparts);
  }
  if (a.dbg_times && tid == 0) a.dbg_times[(size_t)a.nparts * 16 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

template <int C, bool FAST, int MINW, int POL, bool ST32 = false>
hipError_t launch_wave2(const CvhStepArgs &a, hipStream_t s)
{
  using L = Wave2Smem<FAST, C>;
  static_assert(L::bytes <= 64 * 1024, "dynamic LDS above 64 KiB would need hipFuncSetAttribute");
  const int extra = (FAST && a.chain) ? 1 : 0;   // the bookkeeping workgroup
  CVH_LAUNCH((csv_wave2_kernel<C, FAST, MINW, POL, ST32>), a.nparts + extra, L::bytes, s, a, "csv_wave2_kernel<%d, %s, %d, %d, %s>", C, CVH_TF(FAST), MINW,
             POL, CVH_TF(ST32));
  return hipGetLastError();
}

}  // namespace

int cvh_wave2_cols() { return W2; }

// Instantiations <channels, FAST, waves per SIMD, cache policy of the rows, FP32 state>: <1, false, 2, 1, false> STRICT; <1 | 3, true, 3,
// POL, false> FAST (wave2_device.h: POL 1 write-through stores + sc0 loads, 0 plain, 2 plain stores + non-temporal loads -- diagnostic);
// <1 | 3, true, 3, POL, true> the DECLARED FP32-state mode (option "state" = 32).  Round 3 also shipped a 4-waves/SIMD flavour (95.7
// against 57.3 us at 4096^2) and a table-free 3-channel region term (81 against 72 us): neither was ever chosen, neither is a fallback --
// tools/experiments/pruned_flavours/README.md.
hipError_t cvh_launch_wave2(const CvhStepArgs &a, int channels, int fast, hipStream_t s)
{
  if (a.state32) {   // FAST only (api.hip refuses the combination with STRICT)
    if (channels == 3) return a.wave_pol ? launch_wave2<3, true, 3, 1, true>(a, s) : launch_wave2<3, true, 3, 0, true>(a, s);
    return a.wave_pol ? launch_wave2<1, true, 3, 1, true>(a, s) : launch_wave2<1, true, 3, 0, true>(a, s);
  }
  if (channels == 3) return a.wave_pol ? launch_wave2<3, true, 3, 1>(a, s) : launch_wave2<3, true, 3, 0>(a, s);   // FAST only (api.hip routes STRICT to kernel 2)
  if (!fast) return launch_wave2<1, false, 2, 1>(a, s);
  if (a.wave_pol == 2) return launch_wave2<1, true, 3, 2>(a, s);
  return a.wave_pol ? launch_wave2<1, true, 3, 1>(a, s) : launch_wave2<1, true, 3, 0>(a, s);
}
