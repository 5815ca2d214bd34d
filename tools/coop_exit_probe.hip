// Probe for the abort recorded in profiles/r03_C4/summary.txt: every rocprofv3 pass of a process that issued COOPERATIVE launches
// ended in "Segmentation fault" inside exit().  This program has nothing of libchanvese_hip.so in it: an empty kernel, launched
// either plainly or with hipLaunchCooperativeKernel, with everything it created torn down explicitly before main returns.
//   ./coop_exit_probe plain | coop | coop_null | coop_leak  [maps]     (maps: print the executable mappings before returning)
// plain      <<<>>> on a non-blocking stream
// coop       hipLaunchCooperativeKernel on a non-blocking stream; stream synchronised and destroyed, buffer freed
// coop_null  the same on the null stream
// coop_leak  the same as coop but nothing is destroyed (what a process that exits with live contexts looks like)
// Run it bare and under `rocprofv3 --kernel-trace -- ./coop_exit_probe <mode>`; the exit status of each says whether the profiler's
// finaliser + ANY cooperative launch is enough (then the library is not the owner of the fault) or not.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/coop_exit_probe tools/coop_exit_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void touch(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = 42; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv)
{
  const char *mode = argc > 1 ? argv[1] : "coop";
  int *d = nullptr, h = 0;
  hipStream_t s = nullptr;
  CK(hipMalloc(&d, sizeof(int)));
  const bool null_stream = !std::strcmp(mode, "coop_null");
  if (!null_stream) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int rep = 0; rep < 3; ++rep) {
    if (!std::strcmp(mode, "plain")) {
      touch<<<dim3(256), dim3(512), 0, s>>>(d);
      CK(hipGetLastError());
    } else {
      void *params[] = {&d};
      CK(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(touch), dim3(256), dim3(512), params, 0u, s));
    }
  }
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost));
  if (std::strcmp(mode, "coop_leak")) {
    if (s) CK(hipStreamDestroy(s));
    CK(hipFree(d));
  }
  if (argc > 2 && !std::strcmp(argv[2], "maps")) {   // executable mappings of this process: which object owns a faulting frame
    if (FILE *f = std::fopen("/proc/self/maps", "r")) {
      char line[512];
      while (std::fgets(line, sizeof(line), f)) if (std::strstr(line, " r-xp ") || std::strstr(line, " r-x")) std::fputs(line, stdout);
      std::fclose(f);
    }
  }
  std::printf("probe %s: value %d, returning from main\n", mode, h);
  std::fflush(stdout);
  return h == 42 ? 0 : 1;
}
