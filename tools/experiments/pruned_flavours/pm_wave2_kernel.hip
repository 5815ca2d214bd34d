// pm_wave2_kernel.hip — wave-streaming Perona-Malik step with TWO pixels per lane (gfx950, wave64).
//
// Same arithmetic as pm_kernels.hip (src/main.cpp:500-555), data flow of csv_wave2_kernel.hip: a lane
// owns two adjacent columns (16-byte row loads / stores), a wave covers 128 columns of which lanes 0 and
// 63 are the 2-column halos (the flux needs g one column out, g needs I one column further): 124 output
// columns per wave.  Each wave marches down a strip; the rows of the NEXT group of 4 are requested at the
// start of a group and parked in the wave's LDS ring at its end (no load in flight across the loop
// back-edge: hipcc's vmcnt counts stay exact); x-neighbours of I and of g are exchanged through LDS,
// g(i-1..i+1) and the 3 rows of I around the current row stay in registers.  No workgroup barrier.
// Requires w % 2 == 0 and w >= 128; other shapes use pm_wave_kernel.
#include "csv_device.h"
#include "buffer_ops.h"
#include <type_traits>

using namespace cvh_dev;

namespace {

constexpr int PW2 = 124;     // output columns per wave
constexpr int PXP = 130;     // LDS row pitch in doubles (128 used)
constexpr int PR = 4;        // rows per group = ring slots

typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double2_t pm_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void pm_store2(double2_t v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, voff, soff, 0);
}

struct Quad { double w, a, b, e; };   // one row around a lane: west neighbour, own pair, east neighbour

template <bool FAST>
__global__ __launch_bounds__(CVH_BLOCK) void pm_wave2_kernel(const CvhPmArgs a)
{
  __shared__ __attribute__((aligned(16))) double sx[4][(PR + 1) * PXP];   // per wave: 4 row slots of I + 1 slot of g
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = a.h, w = a.w;
  const int nwc = a.tiles_x, nbc = (nwc + 1) >> 1;       // workgroup = 2 wave-columns x 2 strips
  const int nstrips = (h + a.strip_rows - 1) / a.strip_rows;
  const int wc = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % nbc) * 2 + (wave & 1));
  const int ws = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / nbc) * 2 + (wave >> 1));
  if (wc >= nwc || ws >= nstrips) return;
  const int s0 = ws * a.strip_rows;
  const int s1 = (s0 + a.strip_rows) < h ? (s0 + a.strip_rows) : h;
  const int c0 = PW2 * wc - 2 + 2 * lane;                 // column of pixel a; pixel b is c0 + 1
  const bool lane_out = lane >= 1 && lane <= 62 && c0 < w;
  const int cl = c0 < 0 ? 0 : (c0 > w - 2 ? w - 2 : c0);  // even column of the 16-byte load
  // g is forced to 1 on the image's border ring (:518-519); clamped rows / columns sit on that ring
  const bool bord_a = c0 <= 0 || c0 >= w - 1, bord_b = c0 + 1 <= 0 || c0 + 1 >= w - 1;
  double *sI = sx[wave], *sG = sx[wave] + PR * PXP;
  // per-lane LDS indices: own pair, west neighbour of a, east neighbour of b (replicated at the image's sides and
  // at the wave's outermost lanes, whose outer values are never used)
  const int pa = 2 * lane;
  const int pw = (lane == 0 || c0 <= 0) ? pa : pa - 1;
  const int pe = (lane == 63 || c0 + 2 >= w) ? pa + 1 : pa + 2;
  const unsigned rowbytes = (unsigned)w * 8u, nbytes = (unsigned)h * rowbytes;
  const unsigned voff = (unsigned)cl * 8u, voff_st = lane_out ? (unsigned)c0 * 8u : kOobOffset;
  const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, nbytes), rout = make_rsrc(a.out, nbytes);
  auto LD = [&](int r) -> double2_t { return pm_load2(rin, voff, (unsigned)clampi(r, 0, h - 1) * rowbytes); };
  auto fence = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // edge-stopping coefficient of one pixel from its 3x3 window (rows m, 0, p; columns 0, 1, 2): :503-520
  auto g_of = [&](double a00, double a01, double a02, double a10, double a12, double a20, double a21, double a22,
                  bool border) -> double {
    const double rm = a02 - a00, r0 = a12 - a10, rp = a22 - a20;
    const double gx = rm + r0 * 2 + rp;
    const double sm = a00 + a01 * 2 + a02;
    const double sp = a20 + a21 * 2 + a22;
    const double gy = sp - sm;
    double g;
    if (FAST) g = rcp_refined(__builtin_fma(__builtin_fma(gx, gx, gy * gy), a.invK2, 1.0));
    else g = 1.0 / (1.0 + (gx * gx + gy * gy) / a.K2);
    return border ? 1.0 : g;
  };
  auto g_pair = [&](const Quad &m, const Quad &c, const Quad &p, int gi) -> double2_t {
    const bool rb = gi <= 0 || gi >= h - 1;
    return double2_t{g_of(m.w, m.a, m.b, c.w, c.b, p.w, p.a, p.b, rb || bord_a),
                     g_of(m.a, m.b, m.e, c.a, c.e, p.a, p.b, p.e, rb || bord_b)};
  };
  auto quad_from = [&](int slot) -> Quad {
    const double2_t v = *reinterpret_cast<const double2_t *>(sI + slot * PXP + pa);
    return Quad{sI[slot * PXP + pw], v.x, v.y, sI[slot * PXP + pe]};
  };

  // ---- prologue: rows s0-2 .. s0+1 through the ring for their x-neighbours, then rows s0+2 .. s0+5 parked
  Quad qm, q0, q1;          // rows i-1 (only its pair is used after the prologue), i, i+1
  double2_t gm, g0;         // g of rows i-1, i
  double gw, ge;            // g(i, col -/+ 1) of the pair's outer neighbours
  {
    double2_t P[4], T[PR];
#pragma unroll
    for (int j = 0; j < 4; ++j) P[j] = LD(s0 - 2 + j);
#pragma unroll
    for (int j = 0; j < PR; ++j) T[j] = LD(s0 + 2 + j);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<double2_t *>(sI + j * PXP + pa) = P[j];
    fence();
    const Quad qa = quad_from(0), qb = quad_from(1);
    q0 = quad_from(2); q1 = quad_from(3);
    qm = qb;
    gm = g_pair(qa, qb, q0, s0 - 1);
    g0 = g_pair(qb, q0, q1, s0);
    *reinterpret_cast<double2_t *>(sG + pa) = g0;
    fence();
    gw = sG[pw]; ge = sG[pe];
    fence();
#pragma unroll
    for (int j = 0; j < PR; ++j) *reinterpret_cast<double2_t *>(sI + j * PXP + pa) = T[j];
    fence();
  }

  // gfx950 wide-store data hazard (csv_wave2_kernel.hip; tools/store_hazard_probe.hip): a VALU write of a 16-byte buffer store's data
  // registers in the issue slot right behind the store corrupts it (hipcc does not pad stores with a register soffset).  The stored
  // pairs stay live to the group's end; tools/isa_store_hazard.py checks the emitted ISA.
  double2_t stored[PR];
  // one output row; ring slot k holds row i+2
  auto row = [&](int i, int k, bool live) {
    const Quad q2 = quad_from(k);
    const double2_t g1 = g_pair(q0, q1, q2, i + 1);
    *reinterpret_cast<double2_t *>(sG + pa) = g1;       // g(i+1): its x-neighbours are read after this row's arithmetic
    double oa, ob;
    {
      // pixel a: west = lane-1's b, east = own b; pixel b: west = own a, east = lane+1's a   (:527-547)
      const double cna = gm.x, c0a = g0.x, csa = g1.x, cwa = gw, cea = g0.y;
      const double cnb = gm.y, c0b = g0.y, csb = g1.y, cwb = g0.x, ceb = ge;
      const double Ina = qm.a, I0a = q0.a, Isa = q1.a, Iwa = q0.w, Iea = q0.b;
      const double Inb = qm.b, I0b = q0.b, Isb = q1.b, Iwb = q0.a, Ieb = q0.e;
      if (FAST) {
        double s = (csa + c0a) * (Isa - I0a);
        s = __builtin_fma(cea + c0a, Iea - I0a, s);
        s = __builtin_fma(cna + c0a, Ina - I0a, s);
        s = __builtin_fma(cwa + c0a, Iwa - I0a, s);
        oa = __builtin_fma(a.L4, s, I0a);
        double t = (csb + c0b) * (Isb - I0b);
        t = __builtin_fma(ceb + c0b, Ieb - I0b, t);
        t = __builtin_fma(cnb + c0b, Inb - I0b, t);
        t = __builtin_fma(cwb + c0b, Iwb - I0b, t);
        ob = __builtin_fma(a.L4, t, I0b);
      } else {
        const double s = (csa + c0a) * (Isa - I0a) + (cea + c0a) * (Iea - I0a) + (cna + c0a) * (Ina - I0a) + (cwa + c0a) * (Iwa - I0a);
        oa = I0a + a.L * s / 4;  // :544-547
        const double t = (csb + c0b) * (Isb - I0b) + (ceb + c0b) * (Ieb - I0b) + (cnb + c0b) * (Inb - I0b) + (cwb + c0b) * (Iwb - I0b);
        ob = I0b + a.L * t / 4;
      }
    }
    stored[k] = double2_t{oa, ob};
    pm_store2(stored[k], rout, live ? voff_st : kOobOffset, (unsigned)i * rowbytes);
    fence();
    gw = sG[pw]; ge = sG[pe];                           // g(i+1, col -/+ 1) for the next row
    fence();
    qm = q0; q0 = q1; q1 = q2;
    gm = g0; g0 = g1;
  };

  auto group = [&](int ib, auto interior_tag) {
    constexpr bool INTERIOR = decltype(interior_tag)::value;
    double2_t T[PR];
#pragma unroll
    for (int j = 0; j < PR; ++j) T[j] = INTERIOR ? pm_load2(rin, voff, (unsigned)(ib + PR + 2 + j) * rowbytes) : LD(ib + PR + 2 + j);
#pragma unroll
    for (int k = 0; k < PR; ++k) row(ib + k, k, INTERIOR ? true : (ib + k) < s1);
#pragma unroll
    for (int k = 0; k < PR; ++k) asm volatile("; store data of the group's rows still live" :: "v"(stored[k].x), "v"(stored[k].y));
    fence();
#pragma unroll
    for (int j = 0; j < PR; ++j) *reinterpret_cast<double2_t *>(sI + j * PXP + pa) = T[j];
    fence();
  };
  int ib = s0;
  for (; ib + PR <= s1 && ib + 2 * PR + 1 <= h - 1; ib += PR) group(ib, std::true_type{});
  for (; ib < s1; ib += PR) group(ib, std::false_type{});
}

}  // namespace

int cvh_pm_wave2_cols() { return PW2; }

hipError_t cvh_launch_pm_wave2(const CvhPmArgs &a, hipStream_t s)
{
  const int nbc = (a.tiles_x + 1) / 2, nstr = (a.h + a.strip_rows - 1) / a.strip_rows;
  const int grid = nbc * ((nstr + 1) / 2);
  if (a.fast) CVH_LAUNCH(pm_wave2_kernel<true>, grid, 0, s, a, "pm_wave2_kernel<true>");
  else CVH_LAUNCH(pm_wave2_kernel<false>, grid, 0, s, a, "pm_wave2_kernel<false>");
  return hipGetLastError();
}
