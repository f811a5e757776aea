// How accurate are v_rcp_f64 / v_rsq_f64 on gfx950, and how many Newton steps does the solver need?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *a, double *r0, double *r1, double *s0, double *s1, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = a[i];
  double r = __builtin_amdgcn_rcp(x);
  r0[i] = r;
  r1[i] = fma(fma(-x, r, 1.0), r, r);
  double q = __builtin_amdgcn_rsq(x);
  s0[i] = q;
  s1[i] = q * fma(-0.5 * x, q * q, 1.5);
}
int main() {
  const int n = 1 << 20;
  std::mt19937_64 g(1);
  std::vector<double> h(n);
  for (auto &v : h) { double e = std::uniform_real_distribution<double>(-20, 30)(g); v = std::ldexp(std::uniform_real_distribution<double>(1, 2)(g), (int)e); }
  double *a, *r0, *r1, *s0, *s1;
  hipMalloc(&a, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&s0, n * 8); hipMalloc(&s1, n * 8);
  hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, r0, r1, s0, s1, n);
  std::vector<double> o0(n), o1(n), q0(n), q1(n);
  hipMemcpy(o0.data(), r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), r1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(q0.data(), s0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(q1.data(), s1, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, f0 = 0, f1 = 0;
  for (int i = 0; i < n; ++i) {
    long double ex = 1.0L / (long double)h[i], es = 1.0L / sqrtl((long double)h[i]);
    e0 = fmax(e0, (double)fabsl((o0[i] - ex) / ex)); e1 = fmax(e1, (double)fabsl((o1[i] - ex) / ex));
    f0 = fmax(f0, (double)fabsl((q0[i] - es) / es)); f1 = fmax(f1, (double)fabsl((q1[i] - es) / es));
  }
  printf("v_rcp_f64 max rel err: raw %.3e, +1 Newton %.3e\nv_rsq_f64 max rel err: raw %.3e, +1 Newton %.3e (eps = 1.1e-16)\n", e0, e1, f0, f1);
}
