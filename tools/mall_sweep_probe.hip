// diagnostic: ping-pong passes over 2 x 128 MiB (+ 16 MiB image) in the access pattern of the CSV kernel -- 768 workgroups, each
// marching through its own contiguous chunk, all at once -- with the march direction (a) the same in every pass, (b) alternating:
// a pass that starts where the previous one ENDED reads the lines written last, which the 256 MiB Infinity Cache may still hold.
// Also: sc1 (write-through) stores as the shipped kernel uses, and chunk counts 768 / 96.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes)
{ return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000); }
// one workgroup = 256 threads x 16 B = 4 KiB per step; chunk = contiguous range of steps
template <int LD, int ST>
__global__ void pass_k(const double *in, double *out, const unsigned char *img, unsigned steps_per_chunk, int backward)
{
  const __amdgpu_buffer_rsrc_t ri = rsrc(in, 4096u * 4096u * 8u), ro = rsrc(out, 4096u * 4096u * 8u);   // exact size: anything outside is dropped
  const unsigned base = blockIdx.x * steps_per_chunk;
  const unsigned voff = threadIdx.x * 16u;
  for (unsigned s0 = 0; s0 < steps_per_chunk; s0 += 4) {
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned s = backward ? steps_per_chunk - 1 - (s0 + k) : s0 + k;
      v[k] = __builtin_amdgcn_raw_buffer_load_b128(ri, voff, (base + s) * 4096u, LD);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned s = backward ? steps_per_chunk - 1 - (s0 + k) : s0 + k;
      v[k].x += img[((size_t)(base + s) * 512 + threadIdx.x * 2) & (size_t)(4096 * 4096 - 1)] & 1;
      __builtin_amdgcn_raw_buffer_store_b128(v[k], ro, voff, (base + s) * 4096u, ST);
    }
  }
}
// the CSV kernel's pattern: a wave owns a 1 KiB-wide column piece and marches down its strip (row pitch 32 KiB), 4 rows in flight;
// workgroup = 2 adjacent column pieces x 2 adjacent strips; 32 column pieces x `nstrips` strips of `rows` rows
template <int LD, int ST, bool TILED = false>
__global__ void strided_k(const double *in, double *out, const unsigned char *img, int nstrips, int rows)
{
  const __amdgpu_buffer_rsrc_t ri = rsrc(in, 4096u * 4096u * 8u), ro = rsrc(out, 4096u * 4096u * 8u);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bc = blockIdx.x % 16, br = blockIdx.x / 16;
  const int wc = bc * 2 + (wave & 1), st = br * 2 + (wave >> 1);
  if (st >= nstrips) return;
  // TILED: the level set stored column piece by column piece ([piece][row][1 KiB]): a wave's rows are contiguous
  const unsigned voff = TILED ? (unsigned)wc * 4096u * 1024u + (unsigned)lane * 16u : (unsigned)wc * 1024u + (unsigned)lane * 16u;
  const unsigned pitch = TILED ? 1024u : 32768u;
  const unsigned r0 = (unsigned)st * (unsigned)rows;
  for (int r = 0; r < rows; r += 4) {
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(ri, voff, (r0 + r + k) * pitch, LD);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k].x += img[(((size_t)(r0 + r + k) * 4096 + wc * 128 + lane * 2)) & (size_t)(4096 * 4096 - 1)] & 1;
      __builtin_amdgcn_raw_buffer_store_b128(v[k], ro, voff, (r0 + r + k) * pitch, ST);
    }
  }
}
int main()
{
  const size_t n = (size_t)4096 * 4096;
  double *a, *b; unsigned char *img;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&img, n);
  hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8); hipMemset(img, 0, n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200;
  auto run = [&](auto kern, const char *name, int chunks, int mode) {
    const unsigned steps = (unsigned)(n * 8 / 4096 / chunks) & ~3u;   // 4 KiB steps per chunk, a multiple of the unroll (32768 steps in all)
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < iters; ++it) {
        const int back = mode ? (it & 1) : 0;
        if (it & 1) hipLaunchKernelGGL(kern, dim3(chunks), dim3(256), 0, 0, b, a, img, steps, back);
        else hipLaunchKernelGGL(kern, dim3(chunks), dim3(256), 0, 0, a, b, img, steps, back);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-28s %4d chunks x %3u steps (%.0f MB per pass), %s: %.2f us per pass, %.2f TB/s\n", name, chunks, steps, 17.0 / 8.0 * 4096.0 * steps * chunks / 1e6,
             mode ? "alternating" : "same direction", ms * 1e3 / iters, (17.0 / 8.0 * 4096.0 * steps * chunks) / (ms * 1e3 / iters) / 1e6);
    }
  };
  auto run_s = [&](auto kern, const char *name, int nstrips, int rows) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < iters; ++it) {
        if (it & 1) hipLaunchKernelGGL(kern, dim3(16 * ((nstrips + 1) / 2)), dim3(256), 0, 0, b, a, img, nstrips, rows);
        else hipLaunchKernelGGL(kern, dim3(16 * ((nstrips + 1) / 2)), dim3(256), 0, 0, a, b, img, nstrips, rows);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 17.0 * 4096.0 * nstrips * rows;
      printf("%-28s strided: %3d strips x %3d rows (%.0f MB per pass): %.2f us per pass, %.2f TB/s\n", name, nstrips, rows, bytes / 1e6, ms * 1e3 / iters,
             bytes / (ms * 1e3 / iters) / 1e6);
    }
  };
  run_s(strided_k<1, 16>, "sc0 loads / sc1 stores", 96, 40);    // 3840 rows: 267 MB
  run_s(strided_k<1, 16>, "sc0 loads / sc1 stores", 128, 32);   // 4096 rows: 285 MB, 1024 workgroups
  run_s(strided_k<1, 16>, "sc0 loads / sc1 stores", 64, 64);    // 4096 rows, 512 workgroups
  run_s(strided_k<0, 0>, "plain", 128, 32);
  run_s(strided_k<1, 16, true>, "TILED sc0 / sc1", 96, 40);
  run_s(strided_k<1, 16, true>, "TILED sc0 / sc1", 128, 32);
  run_s(strided_k<1, 16, true>, "TILED sc0 / sc1", 64, 64);
  run_s(strided_k<0, 0, true>, "TILED plain", 128, 32);
  for (int chunks : {1024, 768}) {
    run(pass_k<0, 0>, "plain loads / stores", chunks, 0);
    run(pass_k<1, 16>, "sc0 loads / sc1 stores", chunks, 0);
    run(pass_k<1, 16>, "sc0 loads / sc1 stores", chunks, 1);
    run(pass_k<0, 16>, "plain loads / sc1 stores", chunks, 0);
  }
  return 0;
}
