// Issue-rate micro-benchmarks: what does ONE wave per SIMD get per instruction?  (The step kernel runs
// one wave per SIMD at 4096 envs, so its run time is  sum over instructions of the lone-wave issue cost.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate tools/issue_rate.hip && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define REP 16
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc, int n) {
  __shared__ __attribute__((aligned(16))) double buf[64 * 16];
  const int lane = threadIdx.x;
  double a[8];
  int u[8];
  for (int i = 0; i < 8; ++i) { a[i] = out[lane] + i; u[i] = lane + i; }
  for (int i = lane; i < 64 * 16; i += 64) buf[i] = i;
  __syncthreads();
  const double b = 1.0000001, c = 1e-9;
  const d2 *rp = (const d2 *)(buf + (lane >> 4) * 64);   // 16 lanes share an address: broadcast reads
  d2 *wp = (d2 *)(buf + lane * 8);
  unsigned long long t0 = clock64();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int j = 0; j < REP; ++j) {
      if (MODE == 0) {  // 8 independent f64 fma
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = fma(a[i], b, c);
      } else if (MODE == 1) {  // 8 independent int adds
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = u[i] * 3 + 1;
      } else if (MODE == 2) {  // 8 independent f64 mul (v_mul_f64)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = a[i] * b;
      } else if (MODE == 3) {  // 8 ds_read_b128 (broadcast within 16 lanes) consumed by 8 fma (2 per 2 loads..)
        d2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = rp[(i + j) & 31];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = a[i] + v[i][0];
      } else if (MODE == 4) {  // 8 ds_write_b128
#pragma unroll
        for (int i = 0; i < 4; ++i) { d2 v = {a[2 * i], a[2 * i + 1]}; wp[i] = v; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) { d2 v = {a[2 * i + 1], a[2 * i]}; wp[i] = v; }
        __builtin_amdgcn_wave_barrier();
      } else if (MODE == 5) {  // 8 ds_write_b64
#pragma unroll
        for (int i = 0; i < 8; ++i) { buf[lane * 8 + i] = a[i]; }
        __builtin_amdgcn_wave_barrier();
      } else if (MODE == 6) {  // 8 ds_read_b64 per-lane + add
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = buf[((lane + i * 64 + j * 8) & 1023)];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = a[i] + v[i];
      } else if (MODE == 7) {  // 4 fma f64 interleaved with 4 int ops
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = fma(a[i], b, c); u[i] = u[i] * 3 + 1; }
      } else if (MODE == 8) {  // 8 independent f32 fma
        float *f = (float *)a;
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = fmaf(f[i], 1.0001f, 1e-9f);
      }
    }
  }
  unsigned long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + u[i];
  out[lane] = s + buf[lane];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double *d; unsigned long long *c;
  hipMalloc(&d, 64 * 8); hipMalloc(&c, 4096 * 8);
  std::vector<double> h(64, 0.5); hipMemcpy(d, h.data(), 64 * 8, hipMemcpyHostToDevice);
  const int n = 100;
  const char *names[] = {"8 indep v_fma_f64", "8 indep int mad", "8 indep v_mul_f64", "8 ds_read_b128 bcast + 8 add",
                         "8 ds_write_b128", "8 ds_write_b64", "8 ds_read_b64 + 8 add", "4 fma_f64 + 4 int", "8 indep v_fma_f32"};
  for (int blocks : {1, 256, 1024, 2048}) {
#define RUN(M)                                                                             \
  {                                                                                        \
    hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(64), 0, 0, d, c, n);                       \
    hipDeviceSynchronize();                                                                \
    std::vector<unsigned long long> hc(blocks);                                            \
    hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost);                            \
    double s = 0;                                                                          \
    for (auto v : hc) s += v;                                                              \
    printf("%-32s blocks=%4d  %7.2f cycles per group of 8\n", names[M], blocks, s / blocks / (double)(n * REP)); \
  }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
  }
  return 0;
}
