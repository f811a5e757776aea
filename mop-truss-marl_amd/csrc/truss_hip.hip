// truss_hip.hip -- gfx950 (MI355X / CDNA4) backend of include/truss_mi355.h.
//
// One 64-lane wavefront per workgroup; G lanes own one env (64/G envs per wave).  The lane program
// and its phase schedule live in truss_body.h, the host logic in truss_host.h.  Everything an env
// needs between the first load and the last store stays in LDS/registers: the assembled band of K
// (n_pad x W float64), its L*D factor (in place), the load vector and the solution.  HBM sees only
// the algorithmic bytes of the step (DESIGN.md "bytes per env-step").
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#pragma clang fp contract(off)  // float32 decode arithmetic must round like numpy: no implicit FMA

#define TRUSS_HD __device__ __forceinline__
#define TRUSS_UNROLL _Pragma("unroll")
#define TB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// result rows are written once and not read again by this launch: streaming (non-temporal) stores leave
// no dirty lines behind for the end-of-kernel write-back
#define TB_STREAM_STORE(p, v) __builtin_nontemporal_store((v), (p))

// LDS float64 scatter-add (ds_add_f64 on gfx950)
__device__ __forceinline__ void tb_lds_add(double *p, double v) { unsafeAtomicAdd(p, v); }

// Broadcast of a double from lane SRC of every WL-lane team to the lanes of that team, through the DPP
// cross-lane path (no LDS): row_newbcast inside 16-lane rows, one bank-masked move per 8-lane half when
// a row holds two teams; quad_perm for 4-lane teams.
template <int WL, int SRC>
__device__ __forceinline__ double tb_dpp_bcast(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  int rlo, rhi;
  if constexpr (WL == 16) {
    rlo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + SRC, 0xf, 0xf, false);
    rhi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + SRC, 0xf, 0xf, false);
  } else if constexpr (WL == 8) {
    rlo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + SRC, 0xf, 0x3, false);
    rlo = __builtin_amdgcn_update_dpp(rlo, lo, 0x150 + 8 + SRC, 0xf, 0xc, false);
    rhi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + SRC, 0xf, 0x3, false);
    rhi = __builtin_amdgcn_update_dpp(rhi, hi, 0x150 + 8 + SRC, 0xf, 0xc, false);
  } else {
    static_assert(WL == 4, "team width");
    constexpr int qp = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
    rlo = __builtin_amdgcn_update_dpp(lo, lo, qp, 0xf, 0xf, false);
    rhi = __builtin_amdgcn_update_dpp(hi, hi, qp, 0xf, 0xf, false);
  }
  return __hiloint2double(rhi, rlo);
}
// src is a compile-time constant after unrolling; the switch folds to one case
template <class LN>
__device__ __forceinline__ double tb_team_bcast(LN &ln, int src) {
  constexpr int WL = LN::WL_;
  switch (src) {
#define TB_CASE(i) \
  case i:          \
    if constexpr (i < WL) return tb_dpp_bcast<WL, i>(ln.bx); else break;
    TB_CASE(0) TB_CASE(1) TB_CASE(2) TB_CASE(3) TB_CASE(4) TB_CASE(5) TB_CASE(6) TB_CASE(7)
    TB_CASE(8) TB_CASE(9) TB_CASE(10) TB_CASE(11) TB_CASE(12) TB_CASE(13) TB_CASE(14) TB_CASE(15)
#undef TB_CASE
  }
  return ln.bx;
}

// 1/d for the pivot: v_rcp_f64 seed + two Newton steps (full double accuracy for normal d)
__device__ __forceinline__ double tb_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

#ifdef TRUSS_STAMPS
// Diagnostic build only (make diag): lane 0 of one mid-grid workgroup records s_memtime at the phase
// boundaries into a buffer nothing else reads.  Never enabled in libtruss_mi355.so.
__device__ unsigned long long g_truss_stamps[16];
// g_truss_span: every workgroup's first and last stamp as (shader clock, 100 MHz wall clock): spread of
// the workgroups over the launch, effective shader frequency (tools/span.py).
__device__ unsigned long long g_truss_span[4096][4];
#define TRUSS_ST(i)                                                  \
  do {                                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) g_truss_stamps[i] = clock64(); \
    if (threadIdx.x == 0 && ((i) == 0 || (i) == 9) && blockIdx.x < 4096) {              \
      g_truss_span[blockIdx.x][(i) == 0 ? 0 : 2] = clock64();                            \
      g_truss_span[blockIdx.x][(i) == 0 ? 1 : 3] = wall_clock64();                       \
    }                                                                \
    __builtin_amdgcn_sched_barrier(0);                               \
  } while (0)
#endif

// 1/sqrt(x): v_rsq_f64 seed (measured 5.2e-8 relative on gfx950, tools/rcp_accuracy.hip) + two Newton
// steps (one step leaves 4e-15)
__device__ __forceinline__ double tb_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x, r * r, 1.5);
  r = r * fma(-0.5 * x, r * r, 1.5);
  return r;
}

#include "truss_body.h"

template <int G, int WL, int RPL, int EPL>
__global__ __launch_bounds__(64) void truss_step_kernel(const TopoDev T, const StepArgsDev A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  StepLane<G, WL, RPL, EPL> ln;
  ln.init(threadIdx.x, blockIdx.x, T, A, smem);
  constexpr int W_ = StepLane<G, WL, RPL, EPL>::W;
  // The workgroup IS one wavefront: its LDS instructions execute in issue order, so a later ds_read
  // sees an earlier ds_write of any lane without waiting for it.  A wavefront-scope fence plus
  // wave_barrier keeps the compiler from reordering across a phase boundary and emits no s_waitcnt /
  // s_barrier (a __syncthreads() here costs a full LDS round trip per pivot of the factorisation).
#define TB_WAVE_SYNC()                                    \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
#define PH(call) \
  ln.call;       \
  TB_WAVE_SYNC()
#define PH_NS(call) ln.call
#define BAR() TB_WAVE_SYNC()
  TRUSS_STEP_SCHEDULE(PH, PH_NS, BAR, T, A)
#undef PH
#undef PH_NS
#undef BAR
}

__global__ __launch_bounds__(64) void truss_obs_kernel(const TopoDev T, const ObsArgsDev A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ObsLane ln;
  ln.init(threadIdx.x, blockIdx.x, T, A, smem);
#define PH(call) \
  ln.call;       \
  __syncthreads()
#define PH_NS(call) ln.call
  TRUSS_OBS_SCHEDULE(PH, PH_NS, T, A)
#undef PH
#undef PH_NS
}

// ---- host backend ---------------------------------------------------------------------------
#define TRUSS_BACKEND_NAME "hip"
struct truss_topo;
static void *tb_dev_alloc(size_t n) {
  void *p = nullptr;
  return hipMalloc(&p, n) == hipSuccess ? p : nullptr;
}
static void tb_dev_free(void *p) { (void)hipFree(p); }
static bool tb_dev_upload(void *dst, const void *src, size_t n) {
  return hipMemcpy(dst, src, n, hipMemcpyHostToDevice) == hipSuccess;
}
static int tb_launch_step(const truss_topo *t, const StepArgsDev &A, void *stream);
static int tb_launch_obs(const truss_topo *t, const ObsArgsDev &A, void *stream);

#include "truss_host.h"

static int tb_launch_obs(const truss_topo *t, const ObsArgsDev &A, void *stream) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)truss_obs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return tb_fail(TRUSS_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    attr_set = true;
  }
  hipLaunchKernelGGL(truss_obs_kernel, dim3((unsigned)A.B), dim3(64), tb_obs_lds_bytes(t->N), (hipStream_t)stream, t->dev, A);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("obs kernel launch failed: ") + hipGetErrorString(e));
  return TRUSS_OK;
}

template <int G, int WL, int RPL, int EPL>
static int hip_run(const truss_topo *t, const StepArgsDev &A, hipStream_t st) {
  static bool attr_set = false;
  static size_t attr_bytes = 0;
  auto kern = truss_step_kernel<G, WL, RPL, EPL>;
  if (!attr_set || t->lds_bytes > attr_bytes) {  // dynamic LDS beyond 64 KiB needs the opt-in
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return tb_fail(TRUSS_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    attr_set = true;
    attr_bytes = 160 * 1024;
  }
  constexpr int EPB = 64 / G;
  const unsigned grid = (unsigned)((A.B + EPB - 1) / EPB);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), t->lds_bytes, st, t->dev, A);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return TRUSS_OK;
}

static int tb_launch_step(const truss_topo *t, const StepArgsDev &A, void *stream) {
  const TbVariant &v = kVariants[t->variant];
  hipStream_t st = (hipStream_t)stream;
#define X(g, wl, r, e) \
  if (v.G == g && v.WL == wl && v.RPL == r && v.EPL == e) return hip_run<g, wl, r, e>(t, A, st);
  TRUSS_VARIANTS(X)
#undef X
  return tb_fail(TRUSS_EUNSUPPORTED, "variant not compiled");
}

#ifdef TRUSS_STAMPS
extern "C" int truss_debug_span(unsigned long long *out, int nblocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_truss_span), (size_t)nblocks * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
extern "C" int truss_debug_stamps(unsigned long long *out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_truss_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif
