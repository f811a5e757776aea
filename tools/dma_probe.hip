// dma_probe.hip -- semantics check of global_load_lds_dwordx4 (gfx950 LDS-DMA) as the fused GCN layer kernel uses it: lane i of a
// wave-instruction provides its own global address; its 16 bytes land at LDS base (M0) + 16 i.   hipcc -O3 --offload-arch=gfx950 -o tools/bin/dma_probe tools/dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glob_void;
__global__ void k(const char *src, char *dst) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char *g = src + (wave * 64 + (lane ^ 1)) * 16;                  // permuted source: lane i fetches chunk i ^ 1 of the wave's 1 KB
  __builtin_amdgcn_global_load_lds((glob_void *)g, (lds_void *)(sm + wave * 1024), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  ((uint4 *)dst)[threadIdx.x] = ((uint4 *)sm)[threadIdx.x];
}
int main() {
  const int n = 256 * 16;
  std::vector<unsigned char> h(n), o(n);
  for (int i = 0; i < n; ++i) h[i] = (unsigned char)(i * 7 + i / 16);
  char *d, *e;
  hipMalloc(&d, n); hipMalloc(&e, n);
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d, e);
  hipMemcpy(o.data(), e, n, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t)
    for (int b = 0; b < 16; ++b) bad += o[t * 16 + b] != h[((t & ~63) + ((t & 63) ^ 1)) * 16 + b];
  printf("global_load_lds_dwordx4: lane i -> LDS base + 16 i: %s (%d mismatching bytes)\n", bad ? "NO" : "yes", bad);
  return bad != 0;
}
