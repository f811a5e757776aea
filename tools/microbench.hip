// Latency micro-benchmarks that inform the solver design (one wave per CU alone / 4 waves per CU):
// dependent f64 FMA chain, v_rcp_f64 + Newton, LDS b128 read-after-write round trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_fma(double *out, unsigned long long *cyc, int n) {
  double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) a = fma(a, b, c);
  }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_fma_indep(double *out, unsigned long long *cyc, int n) {
  double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 1e-9;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); }
  }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a0 + a1 + a2 + a3;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_rcp(double *out, unsigned long long *cyc, int n) {
  double a = out[threadIdx.x] + 1.5;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      double r = __builtin_amdgcn_rcp(a);
      r = fma(fma(-a, r, 1.0), r, r);
      r = fma(fma(-a, r, 1.0), r, r);
      a = r + 1.25;
    }
  }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_rcp_raw(double *out, unsigned long long *cyc, double *err, int n) {
  double a = out[threadIdx.x] + 1.5;
  double worst = 0;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      double r = __builtin_amdgcn_rcp(a);
      double e = fabs(fma(-a, r, 1.0));
      worst = e > worst ? e : worst;
      double r1 = fma(fma(-a, r, 1.0), r, r);
      double e1 = fabs(fma(-a, r1, 1.0));
      err[1] = e1 > err[1] ? e1 : err[1];
      a = r + 1.25 + 1e-3 * j;
    }
  }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; err[0] = worst; }
}
__global__ void k_lds(double *out, unsigned long long *cyc, int n) {
  __shared__ __attribute__((aligned(16))) double buf[64 * 8 * 2];
  typedef double d2 __attribute__((ext_vector_type(2)));
  int lane = threadIdx.x & 63, g = lane & 7, grp = lane >> 3;
  double a = out[lane];
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      buf[grp * 16 + g] = a;                          // post own entry
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const d2 *p = (const d2 *)(buf + grp * 16);     // read the 8 entries of the group back
      d2 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
      a = v0[0] + v0[1] + v1[0] + v1[1] + v2[0] + v2[1] + v3[0] + v3[1] + 1e-3 * g;
      __builtin_amdgcn_wave_barrier();
    }
  }
  unsigned long long t1 = clock64();
  out[lane] = a;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double *d; unsigned long long *c; double *err;
  hipMalloc(&d, 64 * 8); hipMalloc(&c, 2048 * 8); hipMalloc(&err, 16);
  std::vector<double> h(64, 0.5); hipMemcpy(d, h.data(), 64 * 8, hipMemcpyHostToDevice);
  hipMemset(err, 0, 16);
  const int n = 200;
  auto report = [&](const char *name, int blocks, int per) {
    hipDeviceSynchronize();
    std::vector<unsigned long long> hc(blocks);
    hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : hc) s += v;
    printf("%-28s blocks=%4d  %7.1f cycles per iteration-unit\n", name, blocks, s / blocks / (double)(n * per));
  };
  for (int blocks : {1, 256, 1024}) {
    hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(64), 0, 0, d, c, n); report("dependent v_fma_f64", blocks, 16);
    hipLaunchKernelGGL(k_fma_indep, dim3(blocks), dim3(64), 0, 0, d, c, n); report("4 independent v_fma_f64", blocks, 16);
    hipLaunchKernelGGL(k_rcp, dim3(blocks), dim3(64), 0, 0, d, c, n); report("rcp_f64 + 2 Newton + add", blocks, 16);
    hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(64), 0, 0, d, c, n); report("LDS post b64 + read 4xb128 + 8 adds", blocks, 8);
  }
  hipLaunchKernelGGL(k_rcp_raw, dim3(1), dim3(64), 0, 0, d, c, err, n);
  hipDeviceSynchronize();
  double he[2]; hipMemcpy(he, err, 16, hipMemcpyDeviceToHost);
  printf("v_rcp_f64 raw relative error (|1 - a*r|): %.3e ; after one Newton step: %.3e\n", he[0], he[1]);
  return 0;
}
