// MFMA-f64 probe for the banded LDL^T of the step kernel: would a 4x4 block-pivot window on
// v_mfma_f64_4x4x4_4b_f64 (one 4x4x4 block per 16-lane env, four envs per wave instruction) beat the rank-1 FMA
// updates?  Measured with ONE wave per SIMD (grid 1024 x 64 threads), which is how the step kernel runs at 4096 envs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_f64_probe.hip && /tmp/mfma_probe
// MODE 0  dependent chain of MFMA (D feeds C of the next)        -> latency per instruction
// MODE 1  4 independent accumulators, MFMAs back to back          -> issue cost per instruction
// MODE 2  8 independent v_fma_f64 per "pivot" (what one lane does now per pivot: 7 window FMAs + the right-hand side)
// MODE 3  a block step of 4 pivots as MFMA would do it for BOTH teams of an env: 8 tile updates (4 per team: the
//         2 x 2 tiles of the trailing window), accumulators independent, plus the 4 reciprocal chains of the 4 x 4
//         diagonal block (v_rcp_f64 + 2 Newton steps each, sequential: pivot k+1 of the block needs pivot k)
// MODE 4  the same 4 pivots as they run now: per pivot one reciprocal chain + 8 FMAs (the LDS exchange of the real
//         kernel is left out on both sides: this isolates the arithmetic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double rcp_newton(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc, int n) {
  const int lane = threadIdx.x;
  double a = out[lane] * 1e-3 + 1.0, b = 1.0 + 1e-9 * lane, c0 = 0.5, c1 = 0.25, c2 = 0.125, c3 = 0.0625;
  double r[8];
  for (int i = 0; i < 8; ++i) r[i] = a + i;
  double d = 2.0 + 1e-6 * lane;
  unsigned long long t0 = clock64();
  for (int it = 0; it < n; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = fma(-b, r[(i + 1) & 7] * 1e-9 + a, r[i]);
    } else if (MODE == 3) {
      // 4 sequential pivots of the diagonal block (reciprocal chain each), then 8 independent tile updates
#pragma unroll
      for (int p = 0; p < 4; ++p) d = fma(-b * 1e-9, rcp_newton(d), d);
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, d, r[i], 0, 0, 0);
    } else if (MODE == 4) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const double inv = rcp_newton(d);
        const double l = r[p] * inv;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i != p) r[i] = fma(-l, r[(i + 1) & 7] * 1e-9 + a, r[i]);
        d = fma(-l, 1e-9, d);
      }
    }
  }
  unsigned long long t1 = clock64();
  double s = c0 + c1 + c2 + c3 + d;
  for (int i = 0; i < 8; ++i) s += r[i];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *what, double per) {
  const int blocks = 1024, n = 2000;
  double *out;
  unsigned long long *cyc;
  hipMalloc(&out, blocks * 64 * sizeof(double));
  hipMemset(out, 0, blocks * 64 * sizeof(double));
  hipMalloc(&cyc, blocks * sizeof(unsigned long long));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, n);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, n);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double m = 0;
  for (auto v : h) m += (double)v;
  m /= blocks;
  printf("%-92s %8.1f cycles per %s\n", what, m / n / per, per == 1 ? "iteration" : "instruction");
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0>("v_mfma_f64_4x4x4_4b, dependent chain (latency)", 16);
  run<1>("v_mfma_f64_4x4x4_4b, 4 independent accumulators (issue)", 16);
  run<2>("v_fma_f64 + v_mul_f64 pair, 8 independent (issue, per pair)", 16);
  run<3>("block step on MFMA: 4 chained reciprocals + 8 tile updates (4 pivots, both teams)", 1);
  run<4>("the same 4 pivots as now: 4 x (reciprocal chain + 8 FMAs)", 1);
  return 0;
}
