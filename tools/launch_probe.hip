// diagnostic: back-to-back launch cost of (nearly) empty kernels on one stream, by grid size
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void tiny_k(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1); }
int main()
{
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int *d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {1, 256, 1275, 5120};
  for (int lds : {0, 18000}) for (int g : grids) {
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_k, dim3(g), dim3(256), lds, s, d);
    hipEventRecord(e0, s);
    const int n = 2000;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_k, dim3(g), dim3(256), lds, s, d);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel grid %5d lds %5d: %.2f us per launch\n", g, lds, ms * 1e3 / n);
  }
  for (int g : {1275}) {
    hipEventRecord(e0, s);
    const int n = 2000;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny_k, dim3(g), dim3(256), 0, s, d);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("tiny (one atomic) grid %5d: %.2f us per launch\n", g, ms * 1e3 / n);
  }
  // graph of 100 launches
  {
    hipGraph_t graph; hipGraphExec_t exec;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_k, dim3(1275), dim3(256), 18000, s, d);
    hipStreamEndCapture(s, &graph);
    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphLaunch(exec, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < 20; ++i) hipGraphLaunch(exec, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("graph of 100 empty kernels (grid 1275, lds 18000): %.2f us per kernel\n", ms * 1e3 / 2000);
  }
  return 0;
}
