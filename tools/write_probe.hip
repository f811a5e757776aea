// write_probe.hip -- what the MI355X memory system does with streamed STORES (round 3).
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/write_probe tools/write_probe.hip
// Questions (the fused state-emitting step is bound by the drain of ~80 MB written after the solve):
//   1. store flavour (plain / nontemporal) x footprint (16 MB ... 2 GB, rewritten back to back): does the Infinity Cache absorb
//      re-written lines, is there anything above the ~6.3 TB/s of a plain fill?
//   2. partial lines: only the even / odd 64-byte halves (or 32-byte quarters) of every 128-byte line, as one pass and as two
//      passes (first the even sectors, later the odd ones): what does a line cost that is written in two instalments?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// every thread writes 16-byte chunks q = tid, tid + nthreads, ... of the SELECTED sectors: sector size `sec` chunks (2 = 32 B, 4 = 64 B,
// 8 = 128 B = everything), of every `period` sectors the one at `phase`
template <int NT>
__global__ __launch_bounds__(256) void k_write(f4 *dst, size_t n_chunks_sel, int sec, int period, int phase, float val) {
  const size_t nth = (size_t)gridDim.x * blockDim.x;
  const f4 v = {val, val + 1.0f, val + 2.0f, val + 3.0f};
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_chunks_sel; q += nth) {
    const size_t s = q / sec, r = q % sec;                   // selected sector, chunk inside
    const size_t chunk = (s * period + phase) * sec + r;
    if (NT) __builtin_nontemporal_store(v, &dst[chunk]);
    else dst[chunk] = v;
  }
}

// the same selection, all `period` phases one after the other inside ONE launch (no launch gap between the instalments)
__global__ __launch_bounds__(256) void k_write_phases(f4 *dst, size_t n_chunks_sel, int sec, int period, float val) {
  const size_t nth = (size_t)gridDim.x * blockDim.x;
  const f4 v = {val, val + 1.0f, val + 2.0f, val + 3.0f};
  for (int phase = 0; phase < period; ++phase)
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_chunks_sel; q += nth) {
      const size_t s = q / sec, r = q % sec;
      __builtin_nontemporal_store(v, &dst[(s * period + phase) * sec + r]);
    }
}

static float time_launches(int reps, void (*fn)(void *), void *ctx) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn(ctx);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) fn(ctx);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

struct Ctx { f4 *buf; size_t bytes; int nt, sec, period, passes, grid; };
static void run(void *p) {
  Ctx *c = (Ctx *)p;
  const size_t chunks = c->bytes / 16;
  if (c->passes < 0) {     // all phases inside one launch
    hipLaunchKernelGGL(k_write_phases, dim3(c->grid), dim3(256), 0, 0, c->buf, chunks / c->period, c->sec, c->period, 1.0f);
    return;
  }
  for (int ph = 0; ph < c->passes; ++ph) {
    const size_t sel = chunks / c->period;
    if (c->nt) hipLaunchKernelGGL(k_write<1>, dim3(c->grid), dim3(256), 0, 0, c->buf, sel, c->sec, c->period, ph, 1.0f);
    else hipLaunchKernelGGL(k_write<0>, dim3(c->grid), dim3(256), 0, 0, c->buf, sel, c->sec, c->period, ph, 1.0f);
  }
}

int main() {
  const size_t MAXB = (size_t)2048 << 20;
  f4 *buf;
  CK(hipMalloc(&buf, MAXB));
  CK(hipMemset(buf, 0, MAXB));
  printf("== 1. whole lines: footprint x flavour x grid (us per pass, TB/s)\n");
  for (size_t mb : {16, 64, 100, 200, 400, 2048}) {
    for (int nt = 0; nt < 2; ++nt)
      for (int grid : {1024, 4096, 16384}) {
        Ctx c{buf, mb << 20, nt, 8, 1, 1, grid};
        const int reps = mb >= 400 ? 20 : 100;
        float us = time_launches(reps, run, &c);
        printf("  %5zu MB  %s  grid %5d : %8.2f us  %5.2f TB/s\n", mb, nt ? "nt   " : "plain", grid, us, (double)(mb << 20) / us / 1e6);
      }
  }
  printf("== 2. partial lines of a 100 MB / 2 GB footprint (nt, grid 4096): bytes actually written per us\n");
  for (size_t mb : {100, 2048}) {
    struct { const char *name; int sec, period, passes; } pat[] = {
        {"all 128 B of a line, one pass            ", 8, 1, 1},
        {"even 64-B halves only                    ", 4, 2, 1},
        {"even halves, then odd halves (2 launches)", 4, 2, 2},
        {"even halves, then odd halves (ONE launch) ", 4, 2, -1},
        {"one 32-B quarter of every 64 B           ", 2, 2, 1},
        {"32-B quarters: even, then odd (ONE launch)", 2, 2, -1},
        {"32-B quarters: even, then odd (2 launches)", 2, 2, 2},
        {"one 32-B quarter of every 128 B          ", 2, 4, 1},
        {"32-B quarters in 4 launches              ", 2, 4, 4},
        {"one 16-B chunk of every 32 B             ", 1, 2, 1},
        {"16-B chunks: even, then odd (2 launches) ", 1, 2, 2},
    };
    for (auto &p : pat) {
      Ctx c{buf, mb << 20, 1, p.sec, p.period, p.passes, 4096};
      float us = time_launches(mb >= 400 ? 20 : 100, run, &c);
      const double written = (double)(mb << 20) / p.period * (p.passes < 0 ? p.period : p.passes);
      printf("  %5zu MB  %s : %8.2f us  %5.2f TB/s of written bytes\n", mb, p.name, us, written / us / 1e6);
    }
  }
  CK(hipFree(buf));
  return 0;
}
