// Does straight-line code cost more per instruction than a loop?  (The step kernel is ~56 KB of mostly
// straight-line code executed once per wave; the instruction cache is 64 KB per 2 CUs.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/icache tools/icache.hip && /tmp/icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int UNROLL, int ITERS>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = out[threadIdx.x] + i;
  const double b = 1.0000001, c = 1e-9;
  unsigned long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = fma(a[i], b, c + j * 1e-12);   // distinct constants: no CSE, 8-byte+ encodings
    }
  }
  unsigned long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int UNROLL, int ITERS>
void run(double *d, unsigned long long *c, int blocks) {
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<UNROLL, ITERS>), dim3(blocks), dim3(64), 0, 0, d, c);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hc(blocks);
    hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0, mx = 0;
    for (auto v : hc) { s += v; if (v > mx) mx = v; }
    printf("unroll %5d x iters %4d  blocks=%4d  launch %d: %6.2f cycles per fma (mean), %6.2f (slowest wave)\n", UNROLL, ITERS,
           blocks, rep, s / blocks / (double)(UNROLL * ITERS * 8), mx / (double)(UNROLL * ITERS * 8));
  }
}
int main() {
  double *d; unsigned long long *c;
  hipMalloc(&d, 64 * 8); hipMalloc(&c, 4096 * 8);
  std::vector<double> h(64, 0.5); hipMemcpy(d, h.data(), 64 * 8, hipMemcpyHostToDevice);
  for (int blocks : {1024}) {
    run<8, 512>(d, c, blocks);      // 64 fma in the loop body (~1 KB)
    run<64, 64>(d, c, blocks);      // 512 fma (~6 KB)
    run<256, 16>(d, c, blocks);     // 2048 fma (~24 KB)
    run<512, 8>(d, c, blocks);      // 4096 fma (~48 KB)
    run<1024, 4>(d, c, blocks);     // 8192 fma (~96 KB)
    run<512, 1>(d, c, blocks);      // 48 KB once
    run<1024, 1>(d, c, blocks);     // 96 KB once
  }
  return 0;
}
