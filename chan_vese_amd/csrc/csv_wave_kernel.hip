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
#include "buffer_ops.h"
#include "wave_math.h"
#include "chain_device.h"
#include <type_traits>

using namespace cvh_dev;

namespace {

constexpr int WCOLS = 63;   // output columns per wave
constexpr int XPITCH = 66;  // exchange row: [0] = col-2 of lane 0, [1..64] = lanes, [65] = col+1 of lane 63
constexpr int IMGP = 80;     // bytes per row of the per-wave image tile (5 x 16-byte pieces)

template <int C, bool FAST, bool LUT, int G>
struct WaveSmem {
  static constexpr int R = 4 * G;                                          // rows per group = ring slots
  static constexpr int NS = cvh_nsums(C);
  static constexpr int off_x = 0;                                          // 4 waves x XSLOTS x XPITCH
  static constexpr int wave_doubles = R * XPITCH + 64 + C * R * IMGP / 8;  // row slots + scratch + image tile
  static constexpr int off_red = off_x + 4 * wave_doubles;                 // 4*NS
  static constexpr int off_fin = off_red + 4 * NS + (4 * NS) % 2;          // NS
  static constexpr int off_atan = off_fin + NS + NS % 2;                   // FAST: CVH_ATAN2_N
  static constexpr int off_lut = (off_atan + (FAST ? CVH_ATAN2_N + 1 : 0) + 1) & ~1;  // LUT: C*256 x {term, I}, 16-byte aligned
  static constexpr int off_flag = off_lut + (LUT ? C * 512 : 0);
  static constexpr int doubles = off_flag + 2;
  static constexpr size_t bytes = (size_t)doubles * sizeof(double);
};

// POL: cache policy of the level-set stores (1 = sc1 write-through while the pair fits the Infinity Cache: 1000^2 12.4 -> 11.65 us; beyond
// it write-through costs -- 6144^2: 144 -> 171 us; wave2_device.h has the 2-pixel kernel's figures), chosen by the host (wave_pol)
template <int C, bool FAST, bool LUT, int MINW, bool IMGV, int G, int POL = 0>
__global__ __launch_bounds__(CVH_BLOCK, MINW) void csv_wave_kernel(const CvhStepArgs a)
{
  using L = WaveSmem<C, FAST, LUT, G>;
  constexpr int R = 4 * G;   // rows per group
  constexpr int NS = cvh_nsums(C);
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *sred = smem + L::off_red;
  double *sfin = smem + L::off_fin;
  double *satan = smem + L::off_atan;
  double *slut = smem + L::off_lut;
  int *s_last = (int *)(smem + L::off_flag);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: row arithmetic stays scalar
  const bool chain = FAST && a.chain != nullptr;               // chain_device.h
  const int h = a.h, w = a.w;

  // ---- this wave's strip: workgroup = 4 adjacent wave-columns of one strip
  const int nwc = a.tiles_x;           // wave-columns per image row
  const int nbc = (nwc + 3) >> 2;      // workgroups per strip
  // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  wave_xcd:
  // renumber them so that an XCD works on a contiguous run of workgroups (neighbouring wave-columns
  // of the same strips): halo columns and shared image pieces then hit in that XCD's L2.
  int bid = (int)blockIdx.x;
  const bool bookkeeper = bid >= a.nparts;   // chain mode: one extra workgroup per launch
  if (a.wave_xcd && !bookkeeper) {
    const int nb = a.nparts, x = bid & 7, j = bid >> 3, q = nb >> 3, r = nb & 7;
    if (a.wave_cls > 0) {   // class-major numbering by dispatch round (see csv_wave2_kernel.hip)
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
  const int wc = (bid % nbc) * 4 + wave;
  const int ws = bookkeeper ? 0 : (a.wave_rev ? a.tiles_y - 1 - bid / nbc : bid / nbc);
  // one batch of scalar loads: the sticky stop flag (src/main.cpp:1000) and the strip's rows (the exit test reads all)
  const const_int_p sb = (const_int_p)a.strip_bounds;
  const int stopped = *(const_int_p)&a.st->stopped;
  const int s0 = sb[ws], s1 = sb[ws + 1];
  if ((stopped != 0) | (s1 < s0)) return;

  const unsigned long long t_start = a.dbg_times ? __builtin_amdgcn_s_memrealtime() : 0ull;
  double *xs = smem + L::off_x + wave * L::wave_doubles;
  if (tid == 0) *s_last = 0;

  double c1[C], c2[C], l1[C], l2[C];
  long long chain_entry = 0;
  if (chain) chain_entry = a.chain->v[a.chain_phase][lane];
#pragma unroll
  for (int k = 0; k < C; ++k) {
    if (!chain) { c1[k] = a.st->c1[k]; c2[k] = a.st->c2[k]; }
    l1[k] = a.lambda1[k]; l2[k] = a.lambda2[k];
  }
  if (bookkeeper) { chain_bookkeeper_block<C>(a, chain_entry, sred); return; }
  const double eps = a.eps;
  const double eps2 = eps * eps;
  const FarCoef fc = {a.far_k[0], a.far_k[1], a.far_k[2], a.far_k[3], a.far_k[4], a.far_thr};

  // the tables are filled while the first rows are in flight: see fill_tables() below
  auto fill_tables = [&]() {
    if (chain) chain_means<C>(a, chain_entry, c1, c2);
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

  const bool active = wc < nwc;        // the last workgroup of a strip may hold idle waves
  const int col = WCOLS * wc - 1 + lane;                // lane 0 = left halo column
  const bool lane_valid = active && (lane >= 1) && (col < w);
  if (active) {
    const int colc = clampi(col, 0, w - 1);
    const double fx = (col <= 0) ? 0.0 : 1.0;           // kappa_x(i,0) = 0 (:371)
    // Every vector-memory operation below is issued by ALL lanes on EVERY row (halo / out-of-
    // image lanes are pointed at a dummy location instead of being masked off): the
    // instruction stream is straight-line, so the compiler's counted vmcnt waits let the
    // 4-row-deep load pipeline and the stores stay in flight.
    const bool xlane = lane < 2 * R;
    const int xrow = xlane ? (lane >> 1) & (R - 1) : 0, xside = lane & 1;
    const int xcol = !xlane ? colc : (xside ? clampi(WCOLS * wc + 63, 0, w - 1) : clampi(WCOLS * wc - 2, 0, w - 1));
    double *x_own = xs + 1 + lane;
    double *x_ext = xlane ? xs + xrow * XPITCH + (xside ? 65 : 0) : xs + R * XPITCH + lane;  // other lanes: scratch
    const double *x_w = xs + lane, *x_e = xs + lane + 2;
    const unsigned rowbytes = (unsigned)w * 8u, ubytes = (unsigned)h * rowbytes;   // < 2 GiB (launcher)
    const unsigned voff_u = (unsigned)colc * 8u;                  // byte offset of this lane's column in a row
    const unsigned voff_st = lane_valid ? voff_u : kOobOffset;    // lanes that own no output pixel store nowhere
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.u_in, ubytes);

    // row base pointers are wave-uniform (scalar); the lane contributes a constant 32-bit offset
    // rows past the strip's last neighbour row (s1) are never used: requests for them (the pipeline runs
    // up to 8 rows ahead) are pointed at row s1, which is cached -- no HBM traffic beyond the strip
    const int ulast = s1 < h - 1 ? s1 : h - 1, ilast = s1 - 1;
    auto U = [&](int r) -> double { return buf_load_f64(ru, voff_u, (unsigned)clampi(r, 0, ulast) * rowbytes); };
    const unsigned voff_x = ((unsigned)xrow * (unsigned)w + (unsigned)xcol) * 8u;
    auto UX = [&](int r0) -> double {   // r0 >= 0; away from the bottom edge the lane's offset is a constant
      if (r0 + R - 1 <= ulast) return buf_load_f64(ru, voff_x, (unsigned)r0 * rowbytes);
      return buf_load_f64(ru, ((unsigned)clampi(r0 + xrow, 0, ulast) * (unsigned)w + (unsigned)xcol) * 8u, 0u);
    };
    auto IM = [&](int k, int r) -> int { const uint8_t *rp = a.img[k] + (size_t)clampi(r, 0, ilast) * w; return rp[colc]; };

    // ---- data flow of the march (G = 1)
    // Global loads never stay in flight across the loop back-edge.  Each iteration handles a GROUP of
    // 4 rows ib..ib+3: at its start it requests the level-set rows, halo extras and image pieces of
    // the NEXT group into temporaries; at its end it waits for them and parks them in this wave's LDS
    // (4-slot row ring, image tile).  The rows themselves (own column and both neighbours) are read
    // back from the ring, `um`/`u0` rotate through registers.  hipcc counts every vector-memory
    // operation here in its vmcnt waits and nothing loop-carried is pending at the back-edge, where
    // it would otherwise copy registers and wait for ALL outstanding loads (vmcnt(0)) -- the earlier
    // register-ring form drained its pipeline once per group that way.
    // Ring slot j holds row ib+1+j (66 doubles: west extra, 64 lanes, east extra).
    int im[C][R];
    unsigned char *simg = reinterpret_cast<unsigned char *>(xs + R * XPITCH + 64);
    const int icol0 = (WCOLS * wc - 1) & ~15;                      // 16-byte aligned start column (may be < 0)
    // lanes 20 ch .. 20 ch + 19: the 5 pieces x 4 rows of channel ch -- the planes live in ONE slab (a.img_stride bytes apart), so a
    // single load instruction fetches the group's pieces of all channels (3 channels: 60 lanes; 12 -> 10 vector-memory
    // instructions per group)
    const int ich = lane / (5 * R), il = lane % (5 * R);
    const int ipiece = il % 5, irow = il / 5;
    const bool ilane = lane < 5 * R * C;
    int ipc = icol0 + 16 * ipiece;
    ipc = ipc < 0 ? 0 : (ipc > w - 16 ? w - 16 : ipc);             // clamped pieces only feed clamped columns
    const int ibyte = colc - icol0;                                // this lane's byte within a tile row (0..79)
    const unsigned ich_off = (unsigned)(ilane ? ich : 0) * a.img_stride;
    const unsigned voff_i = ich_off + (unsigned)(ilane ? irow : 0) * (unsigned)w + (unsigned)ipc;
    // a piece clamped at the image edge lands where its columns are expected
    unsigned char *ipiece_dst = simg + (ilane ? ich : 0) * R * IMGP + irow * IMGP + ((icol0 + 16 * ipiece) == ipc ? 16 * ipiece : ipc - icol0);
    const __amdgpu_buffer_rsrc_t ri_all = make_rsrc(a.img[0], (unsigned)(C - 1) * a.img_stride + (unsigned)h * (unsigned)w);
    // Image samples.  IMGV (w % 16 == 0): the 64-byte row segments of 4 rows are fetched as
    // 20 aligned 16-byte pieces by ONE load (lanes 0..19), staged in a per-wave LDS tile and
    // read back as bytes: one vector-memory instruction per 4 rows instead of one 64 x 1-byte
    // load per row (measured: the byte loads alone cost ~18 us of a 4096^2 launch).
    auto IMQ = [&](int r0) -> u32x4_t {
      if (r0 + R - 1 <= ilast) return buf_load_b128(ri_all, voff_i, (unsigned)r0 * (unsigned)w);
      return buf_load_b128(ri_all, ich_off + (unsigned)clampi(r0 + (ilane ? irow : 0), 0, ilast) * (unsigned)w + (unsigned)ipc, 0u);
    };
    auto lds_fence = [&]() {
      // the ring is exchanged between LANES of this wave: LDS operations of one wave execute in
      // order, the fences only stop the compiler from reordering them
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // parks the next group's data: rows T[j] -> slot j, extras, image pieces / bytes
    auto park = [&](const double (&T)[R], double X, u32x4_t IQ, const int (&IB)[C][R]) {
      lds_fence();                              // all reads of the current group are done
#pragma unroll
      for (int j = 0; j < R; ++j) x_own[j * XPITCH] = T[j];
      *x_ext = X;
      if (IMGV) {
        if (ilane) *reinterpret_cast<u32x4_t *>(ipiece_dst) = IQ;
      }
      lds_fence();
#pragma unroll
      for (int ch = 0; ch < C; ++ch)
#pragma unroll
        for (int k = 0; k < R; ++k) im[ch][k] = IMGV ? (int)simg[(ch * R + k) * IMGP + ibyte] : IB[ch][k];
    };

    // ---- prologue: rows s0-2 .. s0 in registers, rows s0+1 .. s0+4 and the image rows s0 .. s0+3 in LDS
    const double um2 = U(s0 - 2);
    double um = U(s0 - 1), u0 = U(s0);
    double uw, ue;
    {
      double T[R];
      u32x4_t IQ = {0, 0, 0, 0};
      int IB[C][R];
#pragma unroll
      for (int j = 0; j < R; ++j) T[j] = U(s0 + 1 + j);
      const double X0 = UX(s0);                 // extras of rows s0 .. s0+3: only row s0's are used
      const double X = UX(s0 + 1);              // extras of rows s0+1 .. s0+4
      if (IMGV) IQ = IMQ(s0);
      else {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
#pragma unroll
          for (int k = 0; k < R; ++k) IB[ch][k] = IM(ch, s0 + k);
        }
      }
      fill_tables();                            // overlaps the prologue's loads
      __syncthreads();
      // neighbours of row s0 through slot 0, before the ring takes rows s0+1 ..
      x_own[0] = u0;
      if (xrow == 0) *x_ext = X0;
      lds_fence();
      uw = x_w[0]; ue = x_e[0];
      park(T, X, IQ, IB);
    }
    double ny_prev = FAST ? normalised4(u0, um2, um + um) : normalised<false>(u0 - um, central(um2, u0));  // ny at row s0-1
    // kappa_y(0, .) = 0 (:372): on the image's first row ny_prev is set to that row's own ny,
    // computed with the very expression the row uses, so ny - ny_prev is exactly 0 there
    if (s0 == 0) {
      const double up0 = x_own[0];              // row 1
      ny_prev = FAST ? normalised4(up0, um, u0 + u0) : normalised<false>(up0 - u0, central(um, up0));
    }

    // DEFER (builds with register room, <= 4 waves/SIMD): no branch inside a row -- the far-field form is
    // evaluated on every lane and the lanes near the contour are corrected once per group; a group is
    // then one basic block and hipcc overlaps the rows.  At 5 waves/SIMD the 8 extra registers would spill,
    // so that build decides far/near per row with a wave-uniform branch.
    constexpr bool DEFER = FAST && MINW <= 4;
    double un_keep[4];                  // DEFER: the group's new values and which lanes were near the contour
    unsigned long long near_mask[4];
    // one row of the march; `live` (wave-uniform) is false only for rows past the strip end
    auto row = [&](int i, int k, bool live) {
      // row i+1 (own column and its x-neighbours, the latter for the next step) from ring slot k
      const double up = x_own[k * XPITCH];
      const double uw_n = x_w[k * XPITCH], ue_n = x_e[k * XPITCH];
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
      if (FAST && DEFER) {
        hv = heaviside_centred_far(un, fc);     // H - 1/2 (see finalize()); near lanes are corrected after the group
        un_keep[k & 3] = un;
        near_mask[k & 3] = __builtin_amdgcn_ballot_w64(fabs(un) < fc.thr);
      } else if (FAST) {                        // decided per WAVE (uniform branch)
        if (__builtin_amdgcn_ballot_w64(fabs(un) < fc.thr) == 0ull) hv = heaviside_centred_far(un, fc);
        else hv = heaviside_centred_near(un, a.inv_eps, satan);
      }
      else hv = heaviside_strict(un, eps);
      // rows past the strip end (wave-uniform) get an empty buffer: every lane is out of range
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, un), make_rsrc(a.u_out, live ? ubytes : 0u), voff_st, (unsigned)i * rowbytes, POL ? 16 : 0);
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
      um = u0; u0 = up;
      uw = uw_n; ue = ue_n;
    };

    int prio = 3;
    if (a.wave_prio) __builtin_amdgcn_s_setprio(3);
    // INTERIOR groups (all four rows inside the strip, every requested row inside the image) run a
    // branch-free copy of the body: no index clamps, no dead rows -- and no control flow around the
    // vector-memory operations, which keeps hipcc's vmcnt counts exact (at a join it assumes the
    // path with the fewest operations in flight).  The groups at the end of a strip take the
    // general copy.
    auto group = [&](int ib, auto interior_tag) {
      constexpr bool INTERIOR = decltype(interior_tag)::value;
      if (a.wave_sync) __builtin_amdgcn_s_barrier();
      if (a.wave_prio) {
        // Equal-work waves drift apart under oldest-first issue arbitration and the tail then
        // runs at 1-2 waves per SIMD.  Waves that are AHEAD lower their priority (by quarter of
        // the strip), so laggards catch up and all waves finish together.
        const int rem = s1 - ib, len = s1 - s0;
        int pq;
        if (a.wave_prio == 1) pq = (rem * 4 - 1) / len;  // 3,2,1,0 by quarters of the strip
        else {  // thresholds crowd towards the end
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
      // requests for the next group (rows ib+4 .. ib+7; its `up` rows are ib+5 .. ib+8)
      double T[R];
      u32x4_t IQ = {0, 0, 0, 0};
      int IB[C][R];
#pragma unroll
      for (int j = 0; j < R; ++j) T[j] = INTERIOR ? buf_load_f64(ru, voff_u, (unsigned)(ib + R + 1 + j) * rowbytes) : U(ib + R + 1 + j);
      const double X = INTERIOR ? buf_load_f64(ru, voff_x, (unsigned)(ib + R + 1) * rowbytes) : UX(ib + R + 1);
      if (IMGV) IQ = INTERIOR ? buf_load_b128(ri_all, voff_i, (unsigned)(ib + R) * (unsigned)w) : IMQ(ib + R);
      else {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
#pragma unroll
          for (int k = 0; k < R; ++k) IB[ch][k] = IM(ch, ib + R + k);
        }
      }
#pragma unroll
      for (int half = 0; half < G; ++half) {
#pragma unroll
        for (int k = 4 * half; k < 4 * half + 4; ++k) row(ib + k, k, INTERIOR ? true : (ib + k) < s1);
        if (FAST && DEFER && (near_mask[0] | near_mask[1] | near_mask[2] | near_mask[3]) != 0ull) {
          // near_field_correction: replace the clamped far-field value by the table value on the lanes
          // with |u| < 64 eps (the sums take the difference)
#pragma unroll
          for (int k = 4 * half; k < 4 * half + 4; ++k) {
            if (near_mask[k & 3] != 0ull && (INTERIOR || (ib + k) < s1)) {
              const double x = un_keep[k & 3];
              const double d = (fabs(x) < fc.thr) ? heaviside_centred_near(x, a.inv_eps, satan) - heaviside_centred_far(x, fc) : 0.0;
              acc[0] += d;
#pragma unroll
              for (int ch = 0; ch < C; ++ch) acc[2 + ch] = __builtin_fma((double)im[ch][k], d, acc[2 + ch]);
            }
          }
        }
      }
      park(T, X, IQ, IB);
    };
    int ib = s0;
    for (; ib + 2 * R <= ulast; ib += R) group(ib, std::true_type{});   // rows up to ib+2R requested, all needed
    for (; ib < s1; ib += R) group(ib, std::false_type{});
    // exact: valid lanes are multiplied by 1, halo / out-of-image lanes by 0
    const double vmask = lane_valid ? 1.0 : 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = acc[s] * vmask;
  } else {
    fill_tables();
    __syncthreads();
  }
  if (!active && a.wave_sync) {                // idle waves still meet the per-iteration barrier
    for (int ib = s0; ib < s1; ib += R) __builtin_amdgcn_s_barrier();
  }

  if (a.dbg_times && lane == 0) {  // diagnostic stamps: only ever written to their own buffer
    unsigned long long *d = a.dbg_times + (size_t)(blockIdx.x * 4 + wave) * 4;
    d[0] = t_start;
    d[1] = __builtin_amdgcn_s_memrealtime();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    d[2] = d[0];   // (the 2-pixel kernel stamps its first group here)
    d[3] = (unsigned long long)xcc | ((unsigned long long)hwid << 8);
  }
  const double total = block_reduce<NS>(acc, sred);
  if (chain) chain_publish<C>(a, total);   // fixed-point atomics + the sum u_diff^2 row: nothing waits (chain_device.h)
  else publish_partials_and_maybe_finalize<C>(a, total, sred, sfin, s_last, a.nparts);
  if (a.dbg_times && tid == 0) a.dbg_times[(size_t)a.nparts * 16 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

template <int C, bool FAST, bool LUT, int MINW, int G>
hipError_t launch_wave_g(const CvhStepArgs &a, hipStream_t s)
{
  using L = WaveSmem<C, FAST, LUT, G>;
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
  const int extra = (FAST && a.chain) ? 1 : 0;   // the bookkeeping workgroup of chain mode
  if constexpr (FAST && LUT && G == 1) {   // the shipped flavours exist with write-through stores too
    if (a.wave_pol == 1) {
      if (imgv) CVH_LAUNCH((csv_wave_kernel<C, FAST, LUT, MINW, true, G, 1>), a.nparts + extra, lds, s, a, "csv_wave_kernel<%d, %s, %s, %d, true, %d, 1>", C, CVH_TF(FAST), CVH_TF(LUT), MINW, G);
      else CVH_LAUNCH((csv_wave_kernel<C, FAST, LUT, MINW, false, G, 1>), a.nparts + extra, lds, s, a, "csv_wave_kernel<%d, %s, %s, %d, false, %d, 1>", C, CVH_TF(FAST), CVH_TF(LUT), MINW, G);
      return hipGetLastError();
    }
  }
  if (imgv) CVH_LAUNCH((csv_wave_kernel<C, FAST, LUT, MINW, true, G>), a.nparts + extra, lds, s, a, "csv_wave_kernel<%d, %s, %s, %d, true, %d, 0>", C, CVH_TF(FAST), CVH_TF(LUT), MINW, G);
  else CVH_LAUNCH((csv_wave_kernel<C, FAST, LUT, MINW, false, G>), a.nparts + extra, lds, s, a, "csv_wave_kernel<%d, %s, %s, %d, false, %d, 0>", C, CVH_TF(FAST), CVH_TF(LUT), MINW, G);
  return hipGetLastError();
}

template <int C, bool FAST, bool LUT, int MINW>
hipError_t launch_wave_v(const CvhStepArgs &a, hipStream_t s)
{
  // 8-row groups (wave_depth 8) double the time a request has to land; built where registers allow
  if constexpr (C == 1 && FAST && LUT) { if (a.wave_depth >= 8) return launch_wave_g<C, FAST, LUT, MINW, 2>(a, s); }
  return launch_wave_g<C, FAST, LUT, MINW, 1>(a, s);
}

template <int C>
hipError_t launch_wave_c(const CvhStepArgs &a, int fast, hipStream_t s)
{
  if (!fast) return launch_wave_v<C, false, false, (C == 1 ? 3 : 2)>(a, s);
  if constexpr (C == 3) {  // 9 accumulators, 3 image tiles: fits 168 registers (3 waves/SIMD) without spilling
    return a.use_lut ? launch_wave_v<C, true, true, 3>(a, s) : launch_wave_v<C, true, false, 3>(a, s);
  } else {
    // 5 waves/SIMD (96 registers) is the most this kernel reaches without spilling; 4 is kept for comparison
    if (a.wave_minw >= 5) return a.use_lut ? launch_wave_v<C, true, true, 5>(a, s) : launch_wave_v<C, true, false, 4>(a, s);
    return a.use_lut ? launch_wave_v<C, true, true, 4>(a, s) : launch_wave_v<C, true, false, 4>(a, s);
  }
}

}  // namespace

int cvh_wave_cols() { return WCOLS; }

hipError_t cvh_launch_wave(const CvhStepArgs &a, int channels, int fast, hipStream_t s)
{
  return channels == 1 ? launch_wave_c<1>(a, fast, s) : launch_wave_c<3>(a, fast, s);
}
