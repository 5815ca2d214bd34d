// csv_strip_kernel.hip — streaming variant of the fused CSV step (gfx950, wave64).
//
// Same arithmetic as csv_kernels.hip (see its header), different data flow:
//   * a 256-thread workgroup owns a 256-column STRIP and marches down it in chunks of R
//     rows; the (R+3) x (256+4) FP64 window of u lives in an LDS ring of R+3 row slots, so
//     the 3 halo rows between consecutive chunks are never re-read from HBM and the
//     previous row's normalised y-gradient stays in registers across chunks;
//   * the NEXT chunk's R rows of u (16-byte coalesced loads) and of the image (one 16-byte
//     load per thread) are requested into registers BEFORE the current chunk is computed,
//     and written to LDS behind a barrier afterwards: HBM latency hides under ~R rows of
//     FP64 work of the same workgroup instead of relying on other workgroups' phases;
//   * one partial row of sums per workgroup (fewer, fatter workgroups => a short
//     finalisation), grid sized to one balanced round of resident workgroups.
// Requires w % 16 == 0 (aligned 16-byte pieces of u rows and image rows); other widths
// use the tile kernel.
#include "csv_device.h"

using namespace cvh_dev;

namespace {

template <int C, int R, bool FAST, bool LUT>
struct StripSmem {
  static constexpr int NS = cvh_nsums(C);
  static constexpr int RING = R + 3;
  static constexpr int off_u = 0;                                         // RING x PITCH doubles
  static constexpr int off_img = off_u + RING * PITCH;                    // C x R x 256 bytes
  static constexpr int off_red = off_img + C * R * TW / 8;                // 4*NS
  static constexpr int off_fin = off_red + 4 * NS + (4 * NS) % 2;         // NS
  static constexpr int off_atan = off_fin + NS + NS % 2;                  // FAST: 2*CVH_ATAN_N
  static constexpr int off_lut = off_atan + (FAST ? 2 * CVH_ATAN_N : 0);  // LUT: C*256
  static constexpr int off_flag = off_lut + (LUT ? C * 256 : 0);
  static constexpr int doubles = off_flag + 2;
  static constexpr size_t bytes = (size_t)doubles * sizeof(double);
};

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

constexpr int CPR = PITCH / 2;  // 16-byte pieces per window row
#ifndef ROW_UNROLL
#define ROW_UNROLL 2
#endif

// global piece `p` (row t of NROWS, piece cc of the row) of rows [row0, row0+NROWS)
template <int NROWS>
__device__ __forceinline__ void load_rows(const double *u_in, int row0, int j0, int h, int w, int tid,
                                          double2_t (&v)[(NROWS * CPR + CVH_BLOCK - 1) / CVH_BLOCK])
{
  constexpr int NP = NROWS * CPR;
#pragma unroll
  for (int rd = 0; rd < (NP + CVH_BLOCK - 1) / CVH_BLOCK; ++rd) {
    const int p = rd * CVH_BLOCK + tid;
    const int q = p < NP ? p : NP - 1;
    const int t = q / CPR, cc = q - t * CPR;
    const int gi = clampi(row0 + t, 0, h - 1), gj = clampi(j0 - 2 + 2 * cc, 0, w - 2);
    v[rd] = *reinterpret_cast<const double2_t *>(u_in + ((size_t)gi * w + gj));
  }
}

// rows land in ring slots (slot0 + t) % RING
template <int NROWS, int RING>
__device__ __forceinline__ void store_rows(double *su, int slot0, int tid,
                                           const double2_t (&v)[(NROWS * CPR + CVH_BLOCK - 1) / CVH_BLOCK])
{
  constexpr int NP = NROWS * CPR;
#pragma unroll
  for (int rd = 0; rd < (NP + CVH_BLOCK - 1) / CVH_BLOCK; ++rd) {
    const int p = rd * CVH_BLOCK + tid;
    if (p < NP) {
      const int t = p / CPR, cc = p - t * CPR;
      int sl = slot0 + t;
      sl = sl >= RING ? sl - RING : sl;
      *reinterpret_cast<double2_t *>(su + sl * PITCH + 2 * cc) = v[rd];
    }
  }
}

template <int C, int R, bool FAST, bool LUT>
__global__ __launch_bounds__(CVH_BLOCK, (FAST ? (R == 12 ? 4 : 3) : 2)) void csv_strip_kernel(const CvhStepArgs a)
{
  using L = StripSmem<C, R, FAST, LUT>;
  constexpr int NS = cvh_nsums(C);
  constexpr int RING = R + 3;
  constexpr int IMG_ROWS_PER_LOAD = CVH_BLOCK / (TW / 16);  // 16 rows of 256 bytes per 256 x 16 B
  constexpr int IMG_LOADS = (R + IMG_ROWS_PER_LOAD - 1) / IMG_ROWS_PER_LOAD;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *su = smem + L::off_u;
  unsigned char *simg = reinterpret_cast<unsigned char *>(smem + L::off_img);
  double *sred = smem + L::off_red;
  double *sfin = smem + L::off_fin;
  double *satan = smem + L::off_atan;
  double *slut = smem + L::off_lut;
  int *s_last = (int *)(smem + L::off_flag);

  if (a.st->stopped) return;  // sticky stop: src/main.cpp:1000

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = a.h, w = a.w;
  const int bx = blockIdx.x % a.tiles_x, by = blockIdx.x / a.tiles_x;
  const int j0 = bx * TW;
  const int s0 = by * a.strip_rows;
  const int s1 = (s0 + a.strip_rows) < h ? (s0 + a.strip_rows) : h;
  const int nchunks = (s1 - s0 + R - 1) / R;
  if (tid == 0) *s_last = 0;

  // image piece owned by this thread in each 16-row image load
  const int irow = tid >> 4, icb = tid & 15;
  int icol = j0 + 16 * icb;
  icol = icol <= w - 16 ? icol : w - 16;

  // ---- prologue: window rows s0-2 .. s0+R into ring slots 0..R+2, image rows of chunk 0
  {
    double2_t v[(RING * CPR + CVH_BLOCK - 1) / CVH_BLOCK];
    load_rows<RING>(a.u_in, s0 - 2, j0, h, w, tid, v);
    uint4_t iv[C][IMG_LOADS];
#pragma unroll
    for (int k = 0; k < C; ++k)
#pragma unroll
      for (int q = 0; q < IMG_LOADS; ++q) {
        const int gi = clampi(s0 + q * IMG_ROWS_PER_LOAD + irow, 0, h - 1);
        iv[k][q] = *reinterpret_cast<const uint4_t *>(a.img[k] + ((size_t)gi * w + icol));
      }
    store_rows<RING, RING>(su, 0, tid, v);
#pragma unroll
    for (int k = 0; k < C; ++k)
#pragma unroll
      for (int q = 0; q < IMG_LOADS; ++q) {
        const int rr = q * IMG_ROWS_PER_LOAD + irow;
        if (rr < R) *reinterpret_cast<uint4_t *>(simg + (k * R + rr) * TW + 16 * icb) = iv[k][q];
      }
  }

  double c1[C], c2[C], l1[C], l2[C];
#pragma unroll
  for (int k = 0; k < C; ++k) { c1[k] = a.st->c1[k]; c2[k] = a.st->c2[k]; l1[k] = a.lambda1[k]; l2[k] = a.lambda2[k]; }
  const double eps = a.eps;
  const double eps2 = eps * eps;

  if (FAST) {
    static_assert(2 * CVH_ATAN_N <= 2 * CVH_BLOCK, "atan table copy assumes two loads per thread");
    const double t0 = a.atan_tab[tid];
    const double t1 = a.atan_tab[tid + CVH_BLOCK < 2 * CVH_ATAN_N ? tid + CVH_BLOCK : 0];
    satan[tid] = t0;
    if (tid + CVH_BLOCK < 2 * CVH_ATAN_N) satan[tid + CVH_BLOCK] = t1;
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

  const int c = tid + 2;
  const int gj = j0 + tid;
  // rows s0-1, s0 and the normalised y-gradient at row s0-1 (ring slots 1, 2, 0)
  double um = su[1 * PITCH + c], u0 = su[2 * PITCH + c];
  double ny_prev = normalised<FAST>(u0 - um, central(su[c], u0));

  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0;

  int win0 = 0;  // ring slot of image row ibase-2
  for (int ch = 0; ch < nchunks; ++ch) {
    const int ibase = s0 + ch * R;
    const bool more = (ch + 1) < nchunks;

    // ---- [A] request the next chunk's rows ibase+R+1 .. ibase+2R and its image rows
    double2_t nv[(R * CPR + CVH_BLOCK - 1) / CVH_BLOCK];
    uint4_t niv[C][IMG_LOADS];
    if (more) {
      load_rows<R>(a.u_in, ibase + R + 1, j0, h, w, tid, nv);
#pragma unroll
      for (int k = 0; k < C; ++k)
#pragma unroll
        for (int q = 0; q < IMG_LOADS; ++q) {
          const int gi = clampi(ibase + R + q * IMG_ROWS_PER_LOAD + irow, 0, h - 1);
          niv[k][q] = *reinterpret_cast<const uint4_t *>(a.img[k] + ((size_t)gi * w + icol));
        }
    }

    // ---- [B] this chunk.  Left-edge column of the wave, lanes <-> rows
    double nx_edge;
    {
      const int er = lane < R ? lane : R - 1;
      int sl = win0 + er + 2;
      sl = sl >= RING ? sl - RING : sl;
      const double *p = &su[sl * PITCH + wave * 64 + 1];
      nx_edge = normalised<FAST>(p[1] - p[0], central(p[-1], p[1]));
    }
    int sl_cur = win0 + 2;  // slot of row ibase
    sl_cur = sl_cur >= RING ? sl_cur - RING : sl_cur;
#pragma unroll ROW_UNROLL
    for (int r = 0; r < R; ++r) {
      const int gi = ibase + r;
      int sl_next = sl_cur + 1;
      sl_next = sl_next >= RING ? 0 : sl_next;
      const double up = su[sl_next * PITCH + c];
      const double uw = (gj == 0) ? u0 : su[sl_cur * PITCH + c - 1];      // BORDER_REPLICATE in x
      const double ue = (gj >= w - 1) ? u0 : su[sl_cur * PITCH + c + 1];
      const double nx = normalised<FAST>(ue - u0, central(uw, ue));  // :365-366
      const double ny = normalised<FAST>(up - u0, central(um, up));  // :367-368
      const double nxl = from_left_lane(nx, read_lane(nx_edge, r));
      const double kx = (gj == 0) ? 0.0 : nx - nxl;                   // :371
      const double ky = (gi == 0) ? 0.0 : ny - ny_prev;               // :372
      const double kappa = kx + ky;                                   // :373

      const bool valid = (gi < s1) && (gj < w);
      int Iv[C];
      double Ik[C];
#pragma unroll
      for (int k = 0; k < C; ++k) { Iv[k] = simg[(k * R + r) * TW + tid]; Ik[k] = (double)Iv[k]; }

      double ud, hv;
      if (FAST) {
        double reg;
        if (LUT) {
          reg = slut[Iv[0]];
#pragma unroll
          for (int k = 1; k < C; ++k) reg += slut[k * 256 + Iv[k]];
        } else {
          reg = 0.0;
#pragma unroll
          for (int k = 0; k < C; ++k) {
            const double d1 = Ik[k] - c1[k], d2 = Ik[k] - c2[k];
            reg += (d2 * d2) * l2[k] - (d1 * d1) * l1[k];
          }
          reg = __builtin_fma(reg, a.beta, a.gamma);
        }
        ud = __builtin_fma(kappa, a.alpha, reg);                      // :985
        const double q = __builtin_fma(u0 * u0, a.dk1, a.dk2);        // 1/delta_eps(u)
        const double r0 = __builtin_amdgcn_rcp(q);
        const double e = __builtin_fma(-q, r0, 1.0);
        ud = ud * __builtin_fma(__builtin_fma(e, e, e), r0, r0);      // :992
      } else {
        ud = 0.0;  // :965
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const double d1 = Ik[k] - c1[k], d2 = Ik[k] - c2[k];
          const double vin = (d1 * d1) * l1[k];   // :307-310
          const double vout = (d2 * d2) * l2[k];
          ud += vout - vin;                        // :979
        }
        ud = kappa * a.alpha + ud * a.beta + a.gamma;   // :985
        ud = ud * (eps / (kPi * (eps2 + u0 * u0)));      // :209, :992
      }
      const double un = u0 + ud;                         // :994
      if (FAST)
        hv = __builtin_fma(atan_table(un * a.inv_eps, satan), 1.0 / kPi, 0.5);
      else
        hv = heaviside_strict(un, eps);
      if (valid) a.u_out[(size_t)gi * w + gj] = un;
      const double hz = valid ? hv : 0.0;
      const double udz = valid ? ud : 0.0;
      acc[0] += hz;
      if (!FAST) acc[1] += valid ? 1 - hv : 0.0;
#pragma unroll
      for (int k = 0; k < C; ++k) {
        if (FAST) {
          acc[2 + k] = __builtin_fma(Ik[k], hz, acc[2 + k]);
        } else {
          acc[2 + k] += Ik[k] * hz;          // :276
          acc[2 + C + k] += valid ? Ik[k] * (1 - hv) : 0.0;
        }
      }
      if (FAST) acc[2 + 2 * C] = __builtin_fma(udz, udz, acc[2 + 2 * C]);
      else acc[2 + 2 * C] += udz * udz;      // :993
      um = u0; u0 = up; ny_prev = ny;
      sl_cur = sl_next;
    }

    // ---- [C][D][E] retire the window: the next chunk's rows overwrite slots win0 .. win0+R-1
    if (more) {
      __syncthreads();
      store_rows<R, RING>(su, win0, tid, nv);
#pragma unroll
      for (int k = 0; k < C; ++k)
#pragma unroll
        for (int q = 0; q < IMG_LOADS; ++q) {
          const int rr = q * IMG_ROWS_PER_LOAD + irow;
          if (rr < R) *reinterpret_cast<uint4_t *>(simg + (k * R + rr) * TW + 16 * icb) = niv[k][q];
        }
      __syncthreads();
      win0 += R;
      win0 = win0 >= RING ? win0 - RING : win0;
    }
  }

  const double total = block_reduce<NS>(acc, sred);
  publish_partials_and_maybe_finalize<C>(a, total, sred, sfin, s_last, gridDim.x);
}

template <int C, int R, bool FAST, bool LUT>
hipError_t launch_strip_v(const CvhStepArgs &a, hipStream_t s)
{
  using L = StripSmem<C, R, FAST, LUT>;
  static_assert(L::bytes <= 64 * 1024, "dynamic LDS above 64 KiB would need hipFuncSetAttribute");
  const int nseg = (a.h + a.strip_rows - 1) / a.strip_rows;
  CVH_LAUNCH((csv_strip_kernel<C, R, FAST, LUT>), a.tiles_x * nseg, L::bytes, s, a, "csv_strip_kernel<%d, %d, %s, %s>", C, R, CVH_TF(FAST), CVH_TF(LUT));
  return hipGetLastError();
}

template <int C>
hipError_t launch_strip_c(const CvhStepArgs &a, int fast, hipStream_t s)
{
  if (a.tile_rows == 12) {
    if (!fast) return launch_strip_v<C, 12, false, false>(a, s);
    return a.use_lut ? launch_strip_v<C, 12, true, true>(a, s) : launch_strip_v<C, 12, true, false>(a, s);
  }
  if (!fast) return launch_strip_v<C, 16, false, false>(a, s);
  return a.use_lut ? launch_strip_v<C, 16, true, true>(a, s) : launch_strip_v<C, 16, true, false>(a, s);
}

}  // namespace

hipError_t cvh_launch_strip(const CvhStepArgs &a, int channels, int fast, hipStream_t s)
{
  return channels == 1 ? launch_strip_c<1>(a, fast, s) : launch_strip_c<3>(a, fast, s);
}
