// store_hazard_probe.hip — which writer of a wide store's data registers corrupts the store on gfx950, and what guard suffices?
//
// Round 2 found: `buffer_store_dwordx4 v[2:5]` directly followed by `ds_read_b128 v[2:5]` stored wrong values in
// lanes 12-15 of every row of 16 lanes, only under memory back-pressure.  This probe isolates the pair in inline assembly (the
// registers are the SAME operand, so no compiler choice is involved) and varies
//   the writer class:  LDS return (ds_read_b128), vector-memory return (buffer_load_dwordx4), VALU (4 x v_mov_b32), none (control)
//   the guard between store and writer: nothing, s_nop of 1 / 2 / 8 / 64 wait states, 16 independent VALU instructions,
//                      s_waitcnt vmcnt(0)
//   the load on the memory system: 8192 waves (8 per SIMD) stream 1 GiB of 1 KiB wave-rows (back-pressure) or 64 waves stream 128 MiB
// and counts stored 16-byte elements that differ from what the registers held when the store was issued, by lane % 16.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/store_hazard_probe tools/store_hazard_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ unsigned pat(unsigned long long elem, unsigned j)
{
  unsigned x = (unsigned)(elem * 2654435761ull) ^ (unsigned)(elem >> 17) ^ (j * 0x9e3779b9u);
  return x | 1u;   // never equals the poisons below (even)
}
constexpr unsigned kPoisonLds = 0xDEADBEE0u, kPoisonMem = 0xFEEDFAC0u, kPoisonValu = 0x0BADC0D0u;

#define NOP8 "s_nop 7\n"
#define NOP64 NOP8 NOP8 NOP8 NOP8 NOP8 NOP8 NOP8 NOP8
#define VALU16 "v_mov_b32 v104, v105\n v_mov_b32 v105, v104\n v_mov_b32 v104, v105\n v_mov_b32 v105, v104\n" \
               "v_mov_b32 v104, v105\n v_mov_b32 v105, v104\n v_mov_b32 v104, v105\n v_mov_b32 v105, v104\n" \
               "v_mov_b32 v104, v105\n v_mov_b32 v105, v104\n v_mov_b32 v104, v105\n v_mov_b32 v105, v104\n" \
               "v_mov_b32 v104, v105\n v_mov_b32 v105, v104\n v_mov_b32 v104, v105\n v_mov_b32 v105, v104\n"

// One kernel per (writer, guard): the asm string must be a literal.  The store data lives in the FIXED registers v[100:103]
// (declared clobbered), so store and writer name the same physical registers whatever the compiler allocates around them.
#define PROBE_KERNEL_S(NAME, STORE, GUARD, WRITER, TAIL)                                                                                 \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, const unsigned *poison, unsigned long long rows_per_wave,      \
                                              unsigned long long nwaves, unsigned *sink)                                     \
  {                                                                                                                          \
    __shared__ u32x4 lds[256];                                                                                               \
    lds[threadIdx.x] = u32x4{kPoisonLds, kPoisonLds, kPoisonLds, kPoisonLds};                                                \
    __syncthreads();                                                                                                         \
    const unsigned lane = threadIdx.x & 63;                                                                                  \
    const unsigned long long gw = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                 \
    const unsigned ldsaddr = threadIdx.x * 16;                                                                               \
    unsigned acc = 0;                                                                                                        \
    for (unsigned long long r = 0; r < rows_per_wave; ++r) {                                                                 \
      const unsigned long long row = gw + r * nwaves;                             /* 1 KiB per wave-row, written ONCE */      \
      const unsigned long long elem = row * 64 + lane;                                                                       \
      const unsigned d0 = pat(elem, 0), d1 = pat(elem, 1), d2 = pat(elem, 2), d3 = pat(elem, 3);                             \
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + row * 256, 0, 1024, 0x00020000);             \
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(poison), 0, 1024, 0x00020000); \
      const unsigned voff = lane * 16, soff = 0;                                                                             \
      unsigned o;                                                                                                            \
      asm volatile("v_mov_b32 v100, %[d0]\n v_mov_b32 v101, %[d1]\n v_mov_b32 v102, %[d2]\n v_mov_b32 v103, %[d3]\n"          \
                   "v_mov_b32 v104, %[d0]\n v_mov_b32 v105, %[d1]\n"                                                          \
                   "s_nop 4\n"                                                                                                \
                   STORE GUARD WRITER                            \
                   TAIL                                                                                                       \
                   "v_xor_b32 %[o], v100, v103\n v_xor_b32 %[o], %[o], v104\n"                                                \
                   : [o] "=&v"(o)                                                                                            \
                   : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [voff] "v"(voff), [rs] "s"(rs), [rp] "s"(rp),    \
                     [soff] "s"(soff), [la] "v"(ldsaddr), [pv] "s"(kPoisonValu)                                              \
                   : "memory", "v100", "v101", "v102", "v103", "v104", "v105");                                              \
      acc ^= o;                                                                                                              \
    }                                                                                                                        \
    if (acc == 0x12345678u) *sink = acc;                                                                                     \
  }

#define ST_X4_SGPR "buffer_store_dwordx4 v[100:103], %[voff], %[rs], %[soff] offen\n"   /* soffset in an SGPR: what the kernels use */
#define ST_X4_IMM "buffer_store_dwordx4 v[100:103], %[voff], %[rs], 0 offen\n"         /* soffset immediate: the form hipcc's recogniser pads */
#define ST_X3_SGPR "buffer_store_dwordx3 v[100:102], %[voff], %[rs], %[soff] offen\n buffer_store_dword v103, %[voff], %[rs], %[soff] offen offset:12\n"
#define ST_X2_SGPR "buffer_store_dwordx2 v[102:103], %[voff], %[rs], %[soff] offen offset:8\n buffer_store_dwordx2 v[100:101], %[voff], %[rs], %[soff] offen\n"
#define PROBE_KERNEL_T(NAME, GUARD, WRITER, TAIL) PROBE_KERNEL_S(NAME, ST_X4_SGPR, GUARD, WRITER, TAIL)
#define PROBE_KERNEL(NAME, GUARD, WRITER) PROBE_KERNEL_T(NAME, GUARD, WRITER, "s_waitcnt vmcnt(0) lgkmcnt(0)\n")
#define W_LDS "ds_read_b128 v[100:103], %[la]\n"
#define W_MEM "buffer_load_dwordx4 v[100:103], %[voff], %[rp], %[soff] offen\n"
#define W_VALU "v_mov_b32 v100, %[pv]\n v_mov_b32 v101, %[pv]\n v_mov_b32 v102, %[pv]\n v_mov_b32 v103, %[pv]\n"
#define W_VALU_REV "v_mov_b32 v103, %[pv]\n v_mov_b32 v102, %[pv]\n v_mov_b32 v101, %[pv]\n v_mov_b32 v100, %[pv]\n"
#define W_NONE ""

PROBE_KERNEL(k_ctrl, "", W_NONE)
PROBE_KERNEL(k_lds_g0, "", W_LDS)
PROBE_KERNEL(k_lds_nop1, "s_nop 0\n", W_LDS)
PROBE_KERNEL(k_lds_nop2, "s_nop 1\n", W_LDS)
PROBE_KERNEL(k_lds_nop8, NOP8, W_LDS)
PROBE_KERNEL(k_lds_nop64, NOP64, W_LDS)
PROBE_KERNEL(k_lds_nop512, NOP64 NOP64 NOP64 NOP64 NOP64 NOP64 NOP64 NOP64, W_LDS)
PROBE_KERNEL(k_lds_valu16, VALU16, W_LDS)
PROBE_KERNEL(k_lds_vm0, "s_waitcnt vmcnt(0)\n", W_LDS)
// stores left in flight across iterations (as in the real kernels): only the LDS return is waited for
PROBE_KERNEL_T(k_lds_g0_fly, "", W_LDS, "s_waitcnt lgkmcnt(0)\n")
PROBE_KERNEL_T(k_lds_nop64_fly, NOP64, W_LDS, "s_waitcnt lgkmcnt(0)\n")
PROBE_KERNEL_T(k_ctrl_fly, "", W_NONE, "s_nop 0\n")
PROBE_KERNEL(k_mem_g0, "", W_MEM)
PROBE_KERNEL(k_mem_nop8, NOP8, W_MEM)
PROBE_KERNEL(k_valu_g0, "", W_VALU)
PROBE_KERNEL(k_valu_rev_g0, "", W_VALU_REV)
PROBE_KERNEL(k_valu_nop1, "s_nop 0\n", W_VALU_REV)
PROBE_KERNEL(k_valu_nop2, "s_nop 1\n", W_VALU_REV)

#define TAILW "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
PROBE_KERNEL_S(k_valu_imm_g0, ST_X4_IMM, "", W_VALU, TAILW)
PROBE_KERNEL_S(k_valu_imm_nop1, ST_X4_IMM, "s_nop 0\n", W_VALU, TAILW)
PROBE_KERNEL_S(k_valu_x3_g0, ST_X3_SGPR, "", "v_mov_b32 v103, %[pv]\n v_mov_b32 v100, %[pv]\n", TAILW)      /* x3 store then dword: v103 belongs to the dword store */
PROBE_KERNEL_S(k_valu_x2_g0, ST_X2_SGPR, "", W_VALU, TAILW)
PROBE_KERNEL_S(k_valu_1between, ST_X4_SGPR, "v_mov_b32 v104, v105\n", W_VALU, TAILW)    /* ONE unrelated VALU instruction between store and writer */

__global__ void check_kernel(const unsigned *out, unsigned long long nrows, unsigned long long *bad_total, unsigned long long *bad_lane16,
                             unsigned long long *bad_dword)
{
  const unsigned long long n = nrows * 64;
  for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (unsigned long long)gridDim.x * blockDim.x) {
    bool bad = false;
    for (unsigned j = 0; j < 4; ++j) {
      if (out[e * 4 + j] != pat(e, j)) { bad = true; atomicAdd(&bad_dword[j], 1ull); }
    }
    if (bad) { atomicAdd(bad_total, 1ull); atomicAdd(&bad_lane16[e % 16], 1ull); }
  }
}

typedef void (*kern_t)(unsigned *, const unsigned *, unsigned long long, unsigned long long, unsigned *);
struct Variant { const char *name; kern_t k; };

int main(int argc, char **argv)
{
  const int reps = argc > 1 ? atoi(argv[1]) : 3;
  unsigned *out, *poison, *sink;
  unsigned long long *cnt;
  const unsigned long long max_rows = (1ull << 30) / 1024;   // 1 GiB of 1 KiB wave-rows
  CHECK(hipMalloc(&out, max_rows * 1024));
  CHECK(hipMalloc(&poison, 1024));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMalloc(&cnt, 21 * 8));
  std::vector<unsigned> hp(256, kPoisonMem);
  CHECK(hipMemcpy(poison, hp.data(), 1024, hipMemcpyHostToDevice));
  const Variant vs[] = {{"control (no writer)", k_ctrl}, {"lds return, no guard", k_lds_g0}, {"lds return, s_nop 1 state", k_lds_nop1},
                        {"lds return, s_nop 2 states", k_lds_nop2}, {"lds return, s_nop 8 states", k_lds_nop8},
                        {"lds return, s_nop 64 states", k_lds_nop64}, {"lds return, s_nop 512 states", k_lds_nop512},
                        {"lds return, 16 VALU between", k_lds_valu16}, {"lds return, s_waitcnt vmcnt(0)", k_lds_vm0},
                        {"lds return, no guard, in flight", k_lds_g0_fly}, {"lds return, s_nop 64, in flight", k_lds_nop64_fly},
                        {"control, stores in flight", k_ctrl_fly},
                        {"vmem return, no guard", k_mem_g0}, {"vmem return, s_nop 8", k_mem_nop8},
                        {"valu v100..v103, no guard", k_valu_g0}, {"valu v103..v100, no guard", k_valu_rev_g0},
                        {"valu v103..v100, s_nop 1 state", k_valu_nop1}, {"valu v103..v100, s_nop 2 states", k_valu_nop2},
                        {"valu, 1 other VALU between", k_valu_1between},
                        {"valu, x4 store IMM soffset, no g.", k_valu_imm_g0}, {"valu, x4 IMM soffset, s_nop 1", k_valu_imm_nop1},
                        {"valu, x3+x1 stores, no guard", k_valu_x3_g0}, {"valu, 2 x2 stores, no guard", k_valu_x2_g0}};
  // load on the memory system: light = 64 waves (one per four CUs) streaming 128 MiB; full = 8192 waves (8 per SIMD) streaming 1 GiB
  struct Mode { const char *name; int blocks, threads; unsigned long long rows_per_wave; } modes[] = {
      {"light: 64 waves", 64, 64, 2048}, {"full: 8192 waves", 2048, 256, 128}};
  printf("%-34s %-18s %14s %12s  bad by lane%%16 [0..15] | bad by dword\n", "variant", "load", "elements", "bad");
  for (const Variant &v : vs) {
    for (const Mode &m : modes) {
      const unsigned long long waves = (unsigned long long)m.blocks * (m.threads / 64), nrows = waves * m.rows_per_wave;
      unsigned long long tot_bad = 0, tot = 0, lane16[16] = {0}, dw[4] = {0};
      for (int rep = 0; rep < reps; ++rep) {
        CHECK(hipMemset(out, 0, nrows * 1024));
        CHECK(hipMemset(cnt, 0, 21 * 8));
        hipLaunchKernelGGL(v.k, dim3(m.blocks), dim3(m.threads), 0, 0, out, poison, m.rows_per_wave, waves, sink);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, out, nrows, cnt, cnt + 1, cnt + 17);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[21];
        CHECK(hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost));
        tot_bad += h[0]; tot += nrows * 64;
        for (int i = 0; i < 16; ++i) lane16[i] += h[1 + i];
        for (int j = 0; j < 4; ++j) dw[j] += h[17 + j];
      }
      printf("%-34s %-18s %14llu %12llu  ", v.name, m.name, tot, tot_bad);
      for (int i = 0; i < 16; ++i) printf("%llu ", lane16[i]);
      printf("| %llu %llu %llu %llu\n", dw[0], dw[1], dw[2], dw[3]);
      fflush(stdout);
    }
  }
  return 0;
}
