// row_newbcast DPP check on gfx950: broadcast lane SRC of each 8-lane half of every 16-lane row.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SRC>
__device__ __forceinline__ double half_row_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  int rlo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + SRC, 0xf, 0x3, false);
  rlo = __builtin_amdgcn_update_dpp(rlo, lo, 0x150 + 8 + SRC, 0xf, 0xc, false);
  int rhi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + SRC, 0xf, 0x3, false);
  rhi = __builtin_amdgcn_update_dpp(rhi, hi, 0x150 + 8 + SRC, 0xf, 0xc, false);
  return __hiloint2double(rhi, rlo);
}
template <int SRC>
__device__ __forceinline__ double quad_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  constexpr int qp = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
  int rlo = __builtin_amdgcn_update_dpp(lo, lo, qp, 0xf, 0xf, false);
  int rhi = __builtin_amdgcn_update_dpp(hi, hi, qp, 0xf, 0xf, false);
  return __hiloint2double(rhi, rlo);
}
__global__ void k(double *out) {
  double v = threadIdx.x * 1.5 + 0.25;
  out[threadIdx.x] = half_row_bcast<3>(v);
  out[64 + threadIdx.x] = quad_bcast<2>(v);
  // dependent chain timing: 64 broadcasts + fma
  double a = v;
  unsigned long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; ++i) a = fma(half_row_bcast<5>(a), 1.0000001, 1e-9);
  unsigned long long t1 = clock64();
  out[128 + threadIdx.x] = a;
  if (threadIdx.x == 0) out[192] = (double)(t1 - t0) / 64;
}
int main() {
  double *d, h[193];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    double e1 = ((l & ~7) + 3) * 1.5 + 0.25, e2 = ((l & ~3) + 2) * 1.5 + 0.25;
    if (h[l] != e1 || h[64 + l] != e2) { ++bad; printf("lane %d: got %g / %g expected %g / %g\n", l, h[l], h[64 + l], e1, e2); }
  }
  printf("dpp row_newbcast/quad_perm broadcast: %s; bcast+fma dependent chain: %.1f cycles per link\n", bad ? "MISMATCH" : "ok", h[192]);
  return bad != 0;
}
