// acc_probe.hip — relative accuracy of v_rsq_f64 / v_rcp_f64 and of one Newton / one Halley step (diagnostic).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; } } while (0)
__global__ void k(const double *x, double *o, int n)
{
  int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  const double s = x[i];
  const double r = __builtin_amdgcn_rsq(s);
  const double e = __builtin_fma(-(s * r), r, 1.0);
  o[i] = r;
  o[n + i] = __builtin_fma(r * 0.5, e, r);                                   // Newton (quadratic)
  o[2 * n + i] = __builtin_fma(r * e, __builtin_fma(e, 0.375, 0.5), r);      // Halley (cubic)
  const double c = __builtin_amdgcn_rcp(s);
  const double f = __builtin_fma(-s, c, 1.0);
  o[3 * n + i] = c;
  o[4 * n + i] = __builtin_fma(f, c, c);                                     // Newton
  o[5 * n + i] = __builtin_fma(__builtin_fma(f, f, f), c, c);                // cubic
}
int main()
{
  const int n = 1 << 20;
  std::vector<double> x(n), o(6 * n);
  unsigned long long st = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x[i] = ldexp(1.0 + (st >> 11) * (1.0 / 9007199254740992.0), (int)(st % 80) - 40); }
  double *dx, *dout; CHK(hipMalloc(&dx, n * 8)); CHK(hipMalloc(&dout, 6 * n * 8));
  CHK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  CHK(hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost));
  const char *nm[6] = {"v_rsq_f64 raw", "rsq + Newton", "rsq + Halley", "v_rcp_f64 raw", "rcp + Newton", "rcp + cubic"};
  for (int v = 0; v < 6; ++v) {
    long double worst = 0;
    for (int i = 0; i < n; ++i) {
      const long double ref = v < 3 ? 1.0L / sqrtl((long double)x[i]) : 1.0L / (long double)x[i];
      const long double err = fabsl(((long double)o[v * n + i] - ref) / ref);
      if (err > worst) worst = err;
    }
    printf("%-14s max relative error %.3Le  (%.2Lf bits, %.2Lf ulp)\n", nm[v], worst, -log2l(worst), worst / 1.1102230246251565e-16L);
  }
  return 0;
}
