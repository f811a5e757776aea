// truss_hip.hip -- gfx950 (MI355X / CDNA4) backend of include/truss_mi355.h.
//
// One 64-lane wavefront per workgroup; G lanes own one env (64/G envs per wave).  The lane program
// and its phase schedule live in truss_body.h, the host logic in truss_host.h.  Everything an env
// needs between the first load and the last store stays in LDS/registers: the assembled band of K
// (n_pad x W float64), its L*D factor (in place), the load vector and the solution.  HBM sees only
// the algorithmic bytes of the step (DESIGN.md "bytes per env-step").
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "../../include/truss_mi355.h"

#pragma clang fp contract(off)  // float32 decode arithmetic must round like numpy: no implicit FMA

#define TRUSS_HD __device__ __forceinline__
#define TRUSS_UNROLL _Pragma("unroll")
#define TB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// result rows are written once and not read again by this launch: streaming (non-temporal) stores leave
// no dirty lines behind for the end-of-kernel write-back
#define TB_STREAM_STORE(p, v) __builtin_nontemporal_store((v), (p))
// observation tensors of the fused step (~100 MB per 4096-env launch, re-written every step): see DESIGN.md section 4.1b / tools/write_probe.hip
#ifdef TRUSS_OBS_STORE_NT
#define TB_OBS_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define TB_OBS_STORE(p, v) (*(p) = (v))
#endif

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

// 1/d in float32: v_rcp_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float tb_rcpf(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}

#ifdef TRUSS_STOP_AT
// Diagnostic build only (tools/lds_by_phase.sh): the step kernel returns at phase boundary TRUSS_STOP_AT (a TRUSS_ST index of the
// schedule), so that PMC counters of builds with increasing boundaries give per-phase differences.  Results are wrong by construction.
#define TRUSS_ST(i) do { if ((i) == TRUSS_STOP_AT) return; } while (0)
#endif
#ifdef TRUSS_STAMPS
// Diagnostic build only (make diag): lane 0 of one mid-grid workgroup records s_memtime at the phase
// boundaries into a buffer nothing else reads.  Never enabled in libtruss_mi355.so.
__device__ unsigned long long g_truss_stamps[32];
// g_truss_span: every workgroup's first and last stamp as (shader clock, 100 MHz wall clock): spread of
// the workgroups over the launch, effective shader frequency (tools/span.py).
__device__ unsigned long long g_truss_span[4096][6];   // [4], [5]: the streaming wave's start / end wall clock (EMIT)
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

// min / max over the G lanes of an env (all lanes get the result), DPP only: xor 1, xor 2 inside a quad,
// row_half_mirror (other quad of the 8), row_mirror (other half of the 16-lane row); beyond a row: ds_bpermute
template <int G, bool MAX>
__device__ __forceinline__ float tb_group_reduce(float v) {
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); };
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  if constexpr (G >= 2) v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));   // quad_perm [1,0,3,2]
  if constexpr (G >= 4) v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));   // quad_perm [2,3,0,1]
  if constexpr (G >= 8) v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));  // row_half_mirror
  if constexpr (G >= 16) v = op(v, dpp(v, std::integral_constant<int, 0x140>{})); // row_mirror
  if constexpr (G >= 32) v = op(v, __shfl_xor(v, 16));
  if constexpr (G >= 64) v = op(v, __shfl_xor(v, 32));
  return v;
}
template <class LN>
__device__ __forceinline__ float tb_group_min(LN &ln, int c) { return tb_group_reduce<LN::G_, false>(ln.pmn[c]); }
template <class LN>
__device__ __forceinline__ float tb_group_max(LN &ln, int c) { return tb_group_reduce<LN::G_, true>(ln.pmx[c]); }

// sum of a double over the G lanes of an env (all lanes get the result): the same butterfly as tb_group_reduce
template <int G>
__device__ __forceinline__ double tb_group_reduce_sum(double v) {
  auto dpp = [](double x, auto ctrl) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), decltype(ctrl)::value, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), decltype(ctrl)::value, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  };
  if constexpr (G >= 2) v = v + dpp(v, std::integral_constant<int, 0xB1>{});
  if constexpr (G >= 4) v = v + dpp(v, std::integral_constant<int, 0x4E>{});
  if constexpr (G >= 8) v = v + dpp(v, std::integral_constant<int, 0x141>{});
  if constexpr (G >= 16) v = v + dpp(v, std::integral_constant<int, 0x140>{});
  if constexpr (G >= 32) v = v + __shfl_xor(v, 16);
  if constexpr (G >= 64) v = v + __shfl_xor(v, 32);
  return v;
}
// objective partials of the step lane: 0 volume, 1 target distance, 2 strain energy / 0 stress ratio, 1 deflection ratio, 2 bad pivot
template <class LN>
__device__ __forceinline__ double tb_group_sum_d(LN &ln, int which) {
  return tb_group_reduce_sum<LN::G_>(which == 0 ? ln.p_vol : which == 1 ? ln.p_dt : ln.p_en);
}
template <class LN>
__device__ __forceinline__ float tb_group_max_f(LN &ln, int which) {
  return tb_group_reduce<LN::G_, true>(which == 0 ? ln.p_c1 : which == 1 ? ln.p_c2 : (float)ln.bad);
}

// EMIT workgroups: the word behind the progress word is set by a streaming wave that gave up waiting (tb_obs_timeout)
template <class LN, class TD>
__device__ __forceinline__ bool tb_obs_timed_out(LN &ln, const TD &T) {
  if constexpr (LN::EMIT_) return *((const __attribute__((address_space(3))) int *)(ln.TB + T.o_flag) + 1) != 0;
  return false;
}

#include "truss_body.h"

// Progress word of an EMIT workgroup (LDS): the compute wave raises it, the streaming wave sleeps on it.  Everything the two waves
// hand over lives in LDS, so the hand-off is a workgroup-scope release store / acquire load of an LDS word: on gfx950 (waves of a
// workgroup share the CU, LDS operations of a wave retire in issue order) that is `s_waitcnt lgkmcnt(0); ds_write_b32` on one side
// and `ds_read_b32; s_waitcnt lgkmcnt(0)` on the other.  The pointer MUST carry the LDS address space: through a generic
// `volatile int *` (rounds 1-2) the compiler emitted flat_store / flat_load ... sc0 sc1 + s_waitcnt vmcnt(0), i.e. every publish and
// every poll went through the vector-memory path and waited for ALL outstanding global stores of the wave.
typedef __attribute__((address_space(3))) int tb_lds_word;
__device__ __forceinline__ tb_lds_word *tb_flag_ptr(char *lds, int o_flag) { return (tb_lds_word *)(lds + o_flag); }
__device__ __forceinline__ void tb_publish(char *lds, int o_flag, int k) {
  __builtin_amdgcn_wave_barrier();
#ifdef TRUSS_FLAG_RELAXED   // A/B only (tools/experiments/README.md): rounds 1-2's reliance on the in-order LDS pipeline, without the release's lgkmcnt(0)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  if (threadIdx.x == 0) __hip_atomic_store(tb_flag_ptr(lds, o_flag), k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  if (threadIdx.x == 0) __hip_atomic_store(tb_flag_ptr(lds, o_flag), k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
// Bounded wait of the streaming wave for progress >= k.  false = gave up (the compute wave never got there -- it faulted or the
// launch is being torn down); the caller then flags the env in status[] (TRUSS_STATUS_OBS_TIMEOUT) instead of streaming garbage.
// The bound (~2^22 polls of >= 64 clocks: >= 0.1 s) keeps the grid draining in every case; a compute wave publishes within tens of
// microseconds.  `tb_await_limit` is lowered by the emulator-side test of the timeout path only.
#ifndef TB_AWAIT_SPINS
#define TB_AWAIT_SPINS (1 << 22)
#endif
__device__ __forceinline__ bool tb_await(char *lds, int o_flag, int k) {
  for (int spin = 0; spin < TB_AWAIT_SPINS; ++spin) {
    const int v = __hip_atomic_load(tb_flag_ptr(lds, o_flag), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (__builtin_amdgcn_readfirstlane(v) >= k) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// the streaming wave gave up (tb_await): leave the timeout word for the compute wave's status store and raise the bit directly too
template <class LN>
__device__ __forceinline__ void tb_obs_timeout(LN &ln, char *lds, int o_flag, int32_t *status) {
  if ((threadIdx.x & 63) == 0) *(tb_flag_ptr(lds, o_flag) + 1) = 1;
  if (ln.g == 0 && ln.active && status) atomicOr(&status[ln.env], TRUSS_STATUS_OBS_TIMEOUT);
}

template <int G, int WL, int RPL, int EPL, bool EMIT>
__global__ __launch_bounds__(EMIT ? 128 : 64) void truss_step_kernel(const TopoDev T, const StepArgsDev A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  StepLane<G, WL, RPL, EPL, EMIT> ln;
  ln.init(threadIdx.x & 63, blockIdx.x, T, A, smem);
  constexpr int W_ = StepLane<G, WL, RPL, EPL, EMIT>::W;
  constexpr bool EMIT_ = EMIT;
  if constexpr (EMIT) {
    // two wavefronts: 0 computes the step, 1 streams the observation tensors (truss_body.h, "WHO streams")
    if (threadIdx.x == 0) {
      *tb_flag_ptr(smem, T.o_flag) = 0;
      *(tb_flag_ptr(smem, T.o_flag) + 1) = 0;   // timeout word
    }
    __syncthreads();   // the only workgroup barrier of the kernel: the progress word starts at 0
    if (threadIdx.x >= 64) {
#define SPH(call) \
  ln.call;        \
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
  __builtin_amdgcn_wave_barrier();                        \
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront")
#ifdef TRUSS_STAMPS
#define SST(i)                                                                                        \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (threadIdx.x == 64 && blockIdx.x == gridDim.x / 2) g_truss_stamps[i] = clock64();              \
    if (threadIdx.x == 64 && ((i) == 20 || (i) == 26) && blockIdx.x < 4096)                           \
      g_truss_span[blockIdx.x][(i) == 20 ? 4 : 5] = wall_clock64();                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
#else
#define SST(i)
#endif
      ln.emit_tables_load(T);   // ahead of every store of this wave in the vmcnt order ...
      __builtin_amdgcn_s_waitcnt(0);   // ... and waited for HERE, while this wave has nothing else to do: gfx9 counts loads and stores
                                       // in one vmcnt, so a first use of a table register behind stores would wait for all of them
                                       // (the compiler put s_waitcnt vmcnt(0) in front of the row tensors' gathers: segment 3 could
                                       // not start before segment 2's 32 KB per workgroup had retired)
      bool ok = tb_await(smem, T.o_flag, 1);
      if (ok) {
        SST(20);
        TRUSS_STREAM_SEG1(SPH, T, A)
        SST(21);
        ok = tb_await(smem, T.o_flag, 2);
      }
      if (ok) {
        SST(22);
        TRUSS_STREAM_SEG2(SPH, T, A)
        SST(23);
        ok = tb_await(smem, T.o_flag, 3);
      }
      if (ok) {
        SST(24);
        TRUSS_STREAM_SEG3(SPH, T, A)
        SST(25);
      } else {
        tb_obs_timeout(ln, smem, T.o_flag, A.status);   // one exit for the three waits
      }
#ifdef TRUSS_STAMPS
      __builtin_amdgcn_s_waitcnt(0);   // all stores of this wave retired (vmcnt(0))
      SST(26);
#endif
#undef SST
#undef SPH
      return;
    }
  }
  // The compute wave has the env's LDS bytes to itself: its LDS instructions execute in issue order, so a later
  // ds_read sees an earlier ds_write of any lane without waiting for it.  A wavefront-scope fence plus
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
#ifdef TRUSS_FAULT_DROP_PUBLISH3   // fault-injection build (make faultinj; tests only): progress 3 is never announced
#define EMIT_POINT(k) do { if ((k) != 3) tb_publish(smem, T.o_flag, k); } while (0)
#else
#define EMIT_POINT(k) tb_publish(smem, T.o_flag, k)
#endif
  TRUSS_STEP_SCHEDULE(PH, PH_NS, BAR, T, A)
#undef PH
#undef PH_NS
#undef BAR
#undef EMIT_POINT
#undef TB_WAVE_SYNC
}

// truss_rollout as ONE launch: every workgroup plays its envs through all n_steps chained steps.  Topology tables,
// per-env constants and the design state (the previous step's result) stay in LDS between steps; the next step's
// actions are fetched while the current step assembles and solves (rollout_prefetch / rollout_stash), so after the
// first step no step waits for HBM.  Per step the same result rows are written as by truss_step (design state into
// the alternating buffers, everything else overwritten), envs are independent: nothing crosses workgroups.
struct RolloutDev {
  TopoDev T;
  StepArgsDev A;                  // the first step's arguments (buffers 0 -> 1, action set 0)
  int32_t n_steps, n_sets;
  size_t gstride, tstride;        // floats between consecutive action sets
};
template <int G, int WL, int RPL, int EPL>
__global__ __launch_bounds__(64) void truss_rollout_kernel(const RolloutDev P_) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int W_ = StepLane<G, WL, RPL, EPL, false>::W;
  constexpr bool EMIT_ = false;
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
#define EMIT_POINT(k) (void)0
  const int n_steps = P_.n_steps;
  for (int s = 0; s < n_steps; ++s) {
    // The arguments are read through a pointer the optimiser cannot see through, once per step: hoisting the ~150
    // loop-invariant kernel-argument loads (and the addresses derived from them) out of the step loop cost 492 SGPR +
    // 199 VGPR spills and 760 bytes of scratch per lane; a step re-reads what it needs from the scalar cache instead.
    // (the pointer keeps the constant address space: through a generic pointer the reads became 500 flat_load's)
    typedef const char __attribute__((address_space(4))) *tb_kernarg_ptr;
    tb_kernarg_ptr ka = (tb_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const RolloutDev &P = *(const RolloutDev *)ka;
    const TopoDev &T = P.T;
    const StepArgsDev &A = P.A;
    StepLane<G, WL, RPL, EPL, false> ln;   // per step: no lane state is carried from one step to the next except through LDS
    ln.init(threadIdx.x, blockIdx.x, T, A, smem);
    const int nset = (s + 1) % P.n_sets;
    ln.rs_first_step = s;
    ln.rs_y_out = (s & 1) ? (float *)A.y_in : A.y_out;          // step s reads buffer s & 1, writes the other one
    ln.rs_sec_out = (s & 1) ? (int32_t *)A.sec_in : A.sec_out;
    ln.rs_next_geo = s + 1 < n_steps ? A.a_geo + (size_t)nset * P.gstride : nullptr;
    ln.rs_next_topo = s + 1 < n_steps ? A.a_topo + (size_t)nset * P.tstride : nullptr;
    TRUSS_STEP_SCHEDULE(PH, PH_NS, BAR, T, A)
    TB_WAVE_SYNC();
  }
#undef PH
#undef PH_NS
#undef BAR
#undef EMIT_POINT
#undef TB_WAVE_SYNC
}

__global__ __launch_bounds__(64) void truss_obs_kernel(const TopoDev T, const ObsArgsDev A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ObsLane ln;
  ln.init(threadIdx.x, blockIdx.x, blockIdx.y, T, A, smem);
#define PH(call) \
  ln.call;       \
  __syncthreads()
#define PH_NS(call) ln.call
  const int r0_tile = ln.role == 1 ? ((int)blockIdx.y - 1) * A.tile_rows : 0;   // uniform over the workgroup: the barriers stay convergent
  TRUSS_OBS_SCHEDULE(PH, PH_NS, T, A, r0_tile)
#undef PH
#undef PH_NS
}

// ---- batched Pareto front + hypervolume (truss_front) ------------------------------------------------
// One 64-lane wave per env, lane i <-> input row i (P <= 64 rows of [obj1, obj2, con1, con2]).  Everything
// quadratic in P is a loop over LDS broadcast reads; the order-sensitive float64 sums (distances, area)
// are accumulated by lane 0 in index order, like the Python loops they replace.
__global__ __launch_bounds__(64) void truss_front_kernel(const truss_front_args_t A) {
  __shared__ double px[64], py[64], pc1[64], pc2[64];      // input rows
  __shared__ double sx[64], sy[64], sd[64], scr[64];        // front sorted by obj1; distances; crowding
  __shared__ int sidx[64], keep[64];
  __shared__ double ax[64], ay[64];                         // all rows sorted by clipped x (hv_all)
  const int b = blockIdx.x, i = threadIdx.x, P = A.max_points;
  int n = A.n_points[b];
  n = n < 0 ? 0 : (n > P ? P : n);
  const bool have = i < n;
  const double *row = A.points + ((size_t)b * P + (have ? i : 0)) * 4;
  const double x = row[0], y = row[1], c1 = row[2], c2 = row[3];
  px[i] = x; py[i] = y; pc1[i] = c1; pc2[i] = c2;
  __syncthreads();
  const bool feas = have && !(c1 > 1.0 || c2 > 1.0);
  bool dom = false, dup = false;
  for (int j = 0; j < n; ++j) {
    const bool fj = !(pc1[j] > 1.0 || pc2[j] > 1.0);
    dom |= fj && px[j] < x && py[j] < y;
    dup |= fj && j < i && px[j] == x && py[j] == y && pc1[j] == c1 && pc2[j] == c2;
  }
  const bool fr = feas && !dom && !dup;
  // position in the front sorted by (obj1, obj2, input order); the same for all rows by clipped x
  const double cx = fmin(x, 1.0), cy = fmin(y, 1.0);
  keep[i] = fr ? 1 : 0;
  __syncthreads();
  int rank = 0, arank = 0;
  for (int j = 0; j < n; ++j) {
    rank += keep[j] && (px[j] < x || (px[j] == x && (py[j] < y || (py[j] == y && j < i))));
    const double cxj = fmin(px[j], 1.0);
    arank += (cxj < cx || (cxj == cx && j < i));
  }
  __syncthreads();
  const unsigned long long fmask = __ballot(fr);
  int nf = __popcll(fmask);
  if (fr) { sx[rank] = x; sy[rank] = y; sidx[rank] = i; }
  if (have) { ax[arank] = cx; ay[arank] = cy; }
  __syncthreads();
  // crowding distance on the sorted front (utils.py:96-110)
  if (i + 1 < nf) {
    const double dx = sx[i] - sx[i + 1], dy = sy[i] - sy[i + 1];
    sd[i] = sqrt(dx * dx + dy * dy);
  }
  __syncthreads();
  if (i < nf) scr[i] = nf == 1 ? 0.0 : (i == 0 ? sd[0] : (i == nf - 1 ? sd[nf - 2] : sd[i - 1] + sd[i]));
  __syncthreads();
  // truncation to max_front (train copy): both ends + the interior points of largest crowding distance
  bool kp = i < nf;
  if ((A.flags & TRUSS_FRONT_TRUNCATE) && nf > A.max_front) {
    if (i > 0 && i < nf - 1) {
      int cr = 0;
      for (int j = 1; j < nf - 1; ++j) cr += (scr[j] > scr[i] || (scr[j] == scr[i] && j < i));
      kp = cr < A.max_front - 2;
    }
  }
  const unsigned long long kmask = __ballot(kp);
  const int pos = __popcll(kmask & ((1ull << i) - 1ull));
  const int nk = __popcll(kmask);
  const double kx = i < nf ? sx[i] : 0.0, ky = i < nf ? sy[i] : 0.0;
  const int kid = i < nf ? sidx[i] : -1;
  __syncthreads();
  if (kp) { sx[pos] = kx; sy[pos] = ky; keep[pos] = kid; }
  nf = nk;
  __syncthreads();
  if (i + 1 < nf) {
    const double dx = sx[i] - sx[i + 1], dy = sy[i] - sy[i + 1];
    sd[i] = sqrt(dx * dx + dy * dy);
  }
  if (A.front_idx && i < P) A.front_idx[(size_t)b * P + i] = i < nf ? keep[i] : -1;
  __syncthreads();
  if (i != 0) return;
  if (A.n_front) A.n_front[b] = nf;
  const double rx = A.ref_points ? A.ref_points[2 * b] : 1.0, ry = A.ref_points ? A.ref_points[2 * b + 1] : 1.0;
  if (A.metrics) {
    double maxd = 0.0, disd = 1.0, sumd = 0.0, stdcd = 1.0, pn = 0.0;
    if (nf >= 2) {
      maxd = sd[0];
      for (int k = 0; k < nf - 1; ++k) { maxd = sd[k] > maxd ? sd[k] : maxd; sumd += sd[k]; }
      double acc = 0.0;
      const double ctr = maxd / (nf - 1);            // sic: the reference centres on max/len (utils.py:131)
      for (int k = 0; k < nf - 1; ++k) acc += (sd[k] - ctr) * (sd[k] - ctr);
      disd = sqrt(acc / (nf - 1));
    }
    if (nf > 3) {
      double s = 0.0, mx = 0.0;
      for (int k = 1; k < nf - 1; ++k) {
        const double cd = fabs(sx[k - 1] - sx[k + 1]) + fabs(sy[k - 1] - sy[k + 1]);
        scr[k] = cd; s += cd; mx = cd > mx ? cd : mx;
      }
      if (s != 0.0) {
        const int m = nf - 2;
        double mean = 0.0;
        for (int k = 1; k < nf - 1; ++k) { scr[k] = scr[k] / mx; mean += scr[k]; }
        mean /= m;
        double var = 0.0, p10 = 0.0;
        for (int k = 1; k < nf - 1; ++k) {
          const double v = scr[k], d = v - mean;
          var += d * d;
          const double v2 = v * v, v4 = v2 * v2;
          p10 += v4 * v4 * v2;
        }
        stdcd = sqrt(var / m);
        pn = pow(p10, 0.1);
      }
    }
    double *M = A.metrics + (size_t)b * 5;
    M[0] = maxd; M[1] = disd; M[2] = pn; M[3] = sumd; M[4] = stdcd;
  }
  if (A.hv_front) {      // the front is sorted by obj1 and its obj2 decreases: closed form of the union area
    double hv = 0.0;
    if (nf > 0 && !(nf == 1 && sx[0] == 1.0 && sy[0] == 1.0)) {
      double area = 0.0, runmin = 1.0, minx = sx[0], miny = sy[0];
      for (int k = 0; k < nf; ++k) {
        const double cxk = fmin(sx[k], 1.0), cyk = fmin(sy[k], 1.0);
        runmin = cyk < runmin ? cyk : runmin;
        const double nx = k + 1 < nf ? fmin(sx[k + 1], 1.0) : 1.0;
        area += (nx - cxk) * (1.0 - runmin);
        minx = sx[k] < minx ? sx[k] : minx;
        miny = sy[k] < miny ? sy[k] : miny;
      }
      hv = area - ((1.0 - rx) * (1.0 - minx) + (1.0 - ry) * (1.0 - miny) - (1.0 - rx) * (1.0 - ry));
    }
    A.hv_front[b] = hv;
  }
  if (A.hv_all) {
    double hv = 0.0;
    if (n > 0 && !(n == 1 && px[0] == 1.0 && py[0] == 1.0)) {
      double area = 0.0, runmin = 1.0, minx = px[0], miny = py[0];
      for (int k = 0; k < n; ++k) {
        runmin = ay[k] < runmin ? ay[k] : runmin;
        const double nx = k + 1 < n ? ax[k + 1] : 1.0;
        area += (nx - ax[k]) * (1.0 - runmin);
        minx = px[k] < minx ? px[k] : minx;
        miny = py[k] < miny ? py[k] : miny;
      }
      hv = area - ((1.0 - rx) * (1.0 - minx) + (1.0 - ry) * (1.0 - miny) - (1.0 - rx) * (1.0 - ry));
    }
    A.hv_all[b] = hv;
  }
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
static int tb_launch_step(const truss_topo *t, const StepArgsDev &A, bool emit, void *stream);
static int tb_launch_obs(const truss_topo *t, const ObsArgsDev &A, void *stream);
static int tb_launch_rollout(const truss_topo *t, const StepArgsDev &A, int n_steps, int n_sets, void *stream);

#include "truss_host.h"

// Dynamic LDS beyond 64 KiB needs an opt-in per kernel AND per device; one flag per (kernel, device), set once
// (the C ABI promises thread safety per stream: relaxed atomics, a repeated hipFuncSetAttribute is harmless).
static constexpr int TB_MAX_DEVICES = 64;
struct TbLdsOptIn {
  std::atomic<bool> done[TB_MAX_DEVICES];
  int ensure(const void *kern) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TB_MAX_DEVICES) return tb_fail(TRUSS_EHIP, "hipGetDevice failed");
    if (done[dev].load(std::memory_order_acquire)) return TRUSS_OK;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return tb_fail(TRUSS_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    done[dev].store(true, std::memory_order_release);
    return TRUSS_OK;
  }
};

static int tb_launch_obs(const truss_topo *t, const ObsArgsDev &A, void *stream) {
  static TbLdsOptIn optin;
  if (int rc = optin.ensure((const void *)truss_obs_kernel)) return rc;
  hipLaunchKernelGGL(truss_obs_kernel, dim3((unsigned)A.B, (unsigned)(A.n_split > 1 ? A.n_split + 1 : 1)), dim3(64), tb_obs_lds_bytes(t->N), (hipStream_t)stream, t->dev, A);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("obs kernel launch failed: ") + hipGetErrorString(e));
  return TRUSS_OK;
}

template <int G, int WL, int RPL, int EPL, bool EMIT>
static int hip_run(const truss_topo *t, const StepArgsDev &A, hipStream_t st) {
  static TbLdsOptIn optin;
  auto kern = truss_step_kernel<G, WL, RPL, EPL, EMIT>;
  if (int rc = optin.ensure((const void *)kern)) return rc;
  constexpr int EPB = 64 / G;
  const unsigned grid = (unsigned)((A.B + EPB - 1) / EPB);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(EMIT ? 128 : 64), EMIT ? t->lds_bytes_emit : t->lds_bytes, st, EMIT ? t->dev_emit : t->dev, A);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return TRUSS_OK;
}

template <int G, int WL, int RPL, int EPL>
static int hip_run_rollout(const truss_topo *t, const StepArgsDev &A, int n_steps, int n_sets, hipStream_t st) {
  static TbLdsOptIn optin;
  auto kern = truss_rollout_kernel<G, WL, RPL, EPL>;
  if (int rc = optin.ensure((const void *)kern)) return rc;
  RolloutDev Q;
  Q.T = t->dev;
  Q.A = A;
  Q.n_steps = n_steps;
  Q.n_sets = n_sets;
  Q.gstride = (size_t)A.B * t->N * 2;
  Q.tstride = (size_t)A.B * t->N * 3;
  constexpr int EPB = 64 / G;
  const unsigned grid = (unsigned)((A.B + EPB - 1) / EPB);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), t->lds_bytes, st, Q);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("rollout kernel launch failed: ") + hipGetErrorString(e));
  return TRUSS_OK;
}
static int tb_launch_rollout(const truss_topo *t, const StepArgsDev &A, int n_steps, int n_sets, void *stream) {
  const TbVariant &v = kVariants[t->variant];
#define X(g, wl, r, e) \
  if (v.G == g && v.WL == wl && v.RPL == r && v.EPL == e) return hip_run_rollout<g, wl, r, e>(t, A, n_steps, n_sets, (hipStream_t)stream);
  TRUSS_ROLLOUT_VARIANTS(X)
#undef X
  return tb_fail(TRUSS_EUNSUPPORTED, "variant not compiled with the persistent rollout");
}

static int tb_launch_step(const truss_topo *t, const StepArgsDev &A, bool emit, void *stream) {
  const TbVariant &v = kVariants[t->variant];
  hipStream_t st = (hipStream_t)stream;
  if (emit) {
#define X(g, wl, r, e) \
  if (v.G == g && v.WL == wl && v.RPL == r && v.EPL == e) return hip_run<g, wl, r, e, true>(t, A, st);
    TRUSS_EMIT_VARIANTS(X)
#undef X
    return tb_fail(TRUSS_EUNSUPPORTED, "variant not compiled with the observation writer");
  }
#define X(g, wl, r, e) \
  if (v.G == g && v.WL == wl && v.RPL == r && v.EPL == e) return hip_run<g, wl, r, e, false>(t, A, st);
  TRUSS_VARIANTS(X)
#undef X
  return tb_fail(TRUSS_EUNSUPPORTED, "variant not compiled");
}

#ifdef TRUSS_STAMPS
extern "C" int truss_debug_span(unsigned long long *out, int nblocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_truss_span), (size_t)nblocks * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
extern "C" int truss_debug_stamps32(unsigned long long *out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_truss_stamps), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
extern "C" int truss_debug_stamps(unsigned long long *out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_truss_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif

#include "truss_gcn.h"
#include "truss_gcn_level.h"

#include "truss_front.h"
extern "C" int truss_front(const truss_front_args_t *a, void *stream) {
  if (int rc = tb_front_check(a)) return rc;
  if (a->n_envs == 0) return TRUSS_OK;
  hipLaunchKernelGGL(truss_front_kernel, dim3((unsigned)a->n_envs), dim3(64), 0, (hipStream_t)stream, *a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("front kernel launch failed: ") + hipGetErrorString(e));
  return TRUSS_OK;
}

// ---- GCN aggregation (inference): one workgroup per env and 256 channels, thread = channel -------------
// Memory-bound (reads H once, writes out once, 16-64 FMAs per element): every global access is a contiguous
// run of channels across the threads of a wave; the adjacency row block sits in LDS and is read as a
// broadcast.  Bias and activation are fused (no extra passes over the [B, N, C] tensor).
template <int NMAX>
__global__ __launch_bounds__(256) void truss_gcn_aggregate_kernel(const float *__restrict__ adj, long a_stride, const float *h,
                                                                  const float *__restrict__ bias, float *out, int N, int C, int act) {
  __shared__ float sA[NMAX * NMAX];
  const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  const float *A = adj + (size_t)b * a_stride;
  for (int i = threadIdx.x; i < N * N; i += 256) sA[i] = A[i];
  __syncthreads();
  if (c >= C) return;
  const float *H = h + (size_t)b * N * C + c;
  float col[NMAX];
#pragma unroll
  for (int j = 0; j < NMAX; ++j) col[j] = j < N ? H[(size_t)j * C] : 0.0f;
  const float bc = bias ? bias[c] : 0.0f;
  float *O = out + (size_t)b * N * C + c;
  for (int i = 0; i < N; ++i) {
    float acc = bc;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) acc = fmaf(j < N ? sA[i * N + j] : 0.0f, col[j], acc);
    if (act == 1) acc = acc > 0.0f ? acc : 0.0f;
    else if (act == 2) acc = 1.0f / (1.0f + expf(-acc));
    O[(size_t)i * C] = acc;
  }
}

// The same for channel counts that are multiples of 4, sized to the channel count: a thread owns FOUR channels of one
// graph (16-byte loads / stores) and the threads of a block are dealt over (graph, channel quad) pairs without gaps, so
// no lane idles whatever C is (thread = channel in 256-wide blocks left 22 % of the lanes idle at C = 200); a block of
// 256 threads then spans up to 256 / (C / 4) + 2 graphs, whose adjacencies it stages in LDS.
template <int NMAX>
__global__ __launch_bounds__(256) void truss_gcn_aggregate4_kernel(const float *__restrict__ adj, long a_stride, const float *h,
                                                                   const float *__restrict__ bias, float *out, int B, int N, int C4, int act,
                                                                   int gmax) {
  extern __shared__ float sA[];                       // [graphs of this block][N][N]
  const long t0 = (long)blockIdx.x * 256, t = t0 + threadIdx.x;
  const int g0 = (int)(t0 / C4);
  int g1 = (int)((t0 + 255) / C4);
  g1 = g1 < B - 1 ? g1 : B - 1;
  const int ng = a_stride ? g1 - g0 + 1 : 1, nn = N * N;
  for (int i = threadIdx.x; i < ng * nn; i += 256) sA[i] = adj[(a_stride ? (size_t)(g0 + i / nn) * a_stride : 0) + i % nn];
  __syncthreads();
  const int gph = (int)(t / C4), c4 = (int)(t % C4);
  if (gph >= B) return;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 *H = (const f4 *)h + ((size_t)gph * N) * C4 + c4;
  f4 col[NMAX];
#pragma unroll
  for (int j = 0; j < NMAX; ++j) col[j] = j < N ? H[(size_t)j * C4] : (f4){0.0f, 0.0f, 0.0f, 0.0f};
  const f4 bc = bias ? ((const f4 *)bias)[c4] : (f4){0.0f, 0.0f, 0.0f, 0.0f};
  const float *A = sA + (a_stride ? (gph - g0) * nn : 0);
  f4 *O = (f4 *)out + ((size_t)gph * N) * C4 + c4;
  for (int i = 0; i < N; ++i) {
    f4 acc = bc;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
      const float aij = j < N ? A[i * N + j] : 0.0f;
      acc[0] = fmaf(aij, col[j][0], acc[0]);
      acc[1] = fmaf(aij, col[j][1], acc[1]);
      acc[2] = fmaf(aij, col[j][2], acc[2]);
      acc[3] = fmaf(aij, col[j][3], acc[3]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (act == 1) acc[q] = acc[q] > 0.0f ? acc[q] : 0.0f;
      else if (act == 2) acc[q] = 1.0f / (1.0f + expf(-acc[q]));
    }
    O[(size_t)i * C4] = acc;
  }
}

// Slab variant (dense or sparse pattern): a block owns a 128-byte channel slab (8 quads) of GB whole graphs.  It stages the slab of
// H in LDS with full-line loads (8 threads = one 128-byte line), then every (graph, row, quad) item sums its terms from LDS and
// stores 16 bytes -- 8 items = one whole line.  HBM sees H once and `out` once; the neighbours' rows come from LDS, not from L2.
// nbr == nullptr: dense, the K = N columns in order.
__global__ __launch_bounds__(256) void truss_gcn_aggregate_slab_kernel(const float *__restrict__ adj, long a_stride,
                                                                       const int16_t *__restrict__ nbr, int K, const float *__restrict__ h,
                                                                       const float *__restrict__ bias, float *__restrict__ out, int B, int N,
                                                                       int C4, int GB, int act) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ f4 sH[];                              // [GB][N][8]
  const int slab = blockIdx.y, q0 = slab * 8;
  const int nq = C4 - q0 < 8 ? C4 - q0 : 8;               // quads of this slab (the last one may be short)
  const int b0 = blockIdx.x * GB;
  const int gb = B - b0 < GB ? B - b0 : GB;
  const int items = gb * N * 8;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int q = it & 7, r = it >> 3;                    // r = g * N + row
    if (q < nq) sH[it] = ((const f4 *)h)[((size_t)b0 * N + r) * C4 + q0 + q];
  }
  __syncthreads();
  for (int it = threadIdx.x; it < items; it += 256) {
    const int q = it & 7, r = it >> 3;
    if (q >= nq) continue;
    const int g = r / N, i = r - g * N;
    const float *Arow = adj + (size_t)(b0 + g) * a_stride + (size_t)i * N;
    const f4 *Hg = sH + (size_t)g * N * 8 + q;
    f4 acc = bias ? ((const f4 *)bias)[q0 + q] : (f4){0.0f, 0.0f, 0.0f, 0.0f};
    if (nbr) {
      const int16_t *nb = nbr + (size_t)i * K;
      for (int k = 0; k < K; ++k) {     // (an unrolled round of 12 with all loads in flight was slower: 45 against 34 us at 64 nodes)
        const int j = nb[k];
        if (j < 0) continue;
        const float a = Arow[j];
        const f4 hv = Hg[j * 8];
        acc[0] = fmaf(a, hv[0], acc[0]);
        acc[1] = fmaf(a, hv[1], acc[1]);
        acc[2] = fmaf(a, hv[2], acc[2]);
        acc[3] = fmaf(a, hv[3], acc[3]);
      }
    } else {
      for (int j = 0; j < N; ++j) {
        const float a = Arow[j];
        const f4 hv = Hg[j * 8];
        acc[0] = fmaf(a, hv[0], acc[0]);
        acc[1] = fmaf(a, hv[1], acc[1]);
        acc[2] = fmaf(a, hv[2], acc[2]);
        acc[3] = fmaf(a, hv[3], acc[3]);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (act == 1) acc[c] = acc[c] > 0.0f ? acc[c] : 0.0f;
      else if (act == 2) acc[c] = 1.0f / (1.0f + expf(-acc[c]));
    }
    ((f4 *)out)[((size_t)b0 * N + r) * C4 + q0 + q] = acc;
  }
}
// launch helper: false when the shape does not suit the slab kernel (the callers' other kernels take over)
static bool tb_launch_gcn_slab(const float *adj, int64_t a_stride, const int16_t *nbr, int K, const float *h, const float *bias, float *out,
                               int B, int N, int C, int act, hipStream_t st) {
  if ((C & 3) || (((size_t)h | (size_t)out | (size_t)bias) & 15) != 0) return false;   // (out may alias h: a block reads its whole tile first)
  const int GB = std::max(1, tb_env_int("TRUSS_GCN_SLAB_ITEMS", 512) / (8 * N));   // graphs per block: >= two rounds of items for small graphs
  const size_t lds = (size_t)GB * N * 8 * 16;
  if (lds > 48 * 1024) return false;
  const int C4 = C / 4;
  dim3 grid((unsigned)((B + GB - 1) / GB), (unsigned)((C4 + 7) / 8));
  hipLaunchKernelGGL(truss_gcn_aggregate_slab_kernel, grid, dim3(256), lds, st, adj, (long)a_stride, nbr, K, h, bias, out, B, N, C4, GB, act);
  return true;
}

// Sparse-pattern variant: thread = (graph, row, channel quad), flat over the launch; a row's <= 16 listed neighbours instead of
// all N columns.  The H rows a thread reads are 16-byte loads that the threads of a row issue contiguously (C floats); a graph's
// rows are re-read by their neighbours' threads from L2 / L1, so HBM sees H once and `out` once.
template <int KR>   // neighbours per round: all loads of a round are in flight together (KR = 12 covers a truss row in one round)
__global__ __launch_bounds__(256) void truss_gcn_aggregate_sparse_kernel(const float *__restrict__ adj, long a_stride,
                                                                         const int16_t *__restrict__ nbr, int K, const float *__restrict__ h,
                                                                         const float *__restrict__ bias, float *__restrict__ out, long total,
                                                                         int N, int C4, int act) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int c4 = (int)(t % C4);
  const long r = t / C4;
  const int i = (int)(r % N);
  const long b = r / N;
  const float *Arow = adj + b * a_stride + (long)i * N;
  const f4 *Hb = (const f4 *)h + (size_t)b * N * C4 + c4;
  const int16_t *nb = nbr + (long)i * K;
  f4 acc = bias ? ((const f4 *)bias)[c4] : (f4){0.0f, 0.0f, 0.0f, 0.0f};
  for (int k0 = 0; k0 < K; k0 += KR) {
    int j[KR];
    float a[KR];
    f4 hv[KR];
#pragma unroll
    for (int q = 0; q < KR; ++q) {
      j[q] = k0 + q < K ? (int)nb[k0 + q] : -1;
      const int jc = j[q] < 0 ? i : j[q];
      a[q] = Arow[jc];
      hv[q] = Hb[(size_t)jc * C4];
    }
#pragma unroll
    for (int q = 0; q < KR; ++q) {
      if (j[q] < 0) continue;
      acc[0] = fmaf(a[q], hv[q][0], acc[0]);
      acc[1] = fmaf(a[q], hv[q][1], acc[1]);
      acc[2] = fmaf(a[q], hv[q][2], acc[2]);
      acc[3] = fmaf(a[q], hv[q][3], acc[3]);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (act == 1) acc[q] = acc[q] > 0.0f ? acc[q] : 0.0f;
    else if (act == 2) acc[q] = 1.0f / (1.0f + expf(-acc[q]));
  }
  ((f4 *)out)[t] = acc;
}

extern "C" int truss_gcn_aggregate_sparse(const float *adj, int64_t a_batch_stride, const int16_t *nbr, int32_t k_nbr, const float *h,
                                          const float *bias, float *out, int32_t n_batch, int32_t n_nodes, int32_t n_channels,
                                          int32_t act, void *stream) {
  if (!adj || !nbr || !h || !out) return tb_fail(TRUSS_EINVAL, "truss_gcn_aggregate_sparse: NULL argument");
  if (n_batch < 0 || n_nodes < 1 || n_nodes > 32767 || k_nbr < 1 || k_nbr > 16 || n_channels < 4 || (n_channels & 3) || act < 0 || act > 2)
    return tb_fail(TRUSS_EINVAL, "truss_gcn_aggregate_sparse: n_nodes 1..32767, k_nbr 1..16, n_channels a multiple of 4, act 0..2");
  if ((((size_t)h | (size_t)out | (size_t)bias) & 15) != 0 || h == out)
    return tb_fail(TRUSS_EINVAL, "truss_gcn_aggregate_sparse: h / out / bias must be 16-byte aligned, out must not alias h");
  if (n_batch == 0) return TRUSS_OK;
  // up to 128 nodes the slab kernel (rows from LDS: 33-36 us at 64 / 128 nodes against 40-42; at 256 nodes a block walks 8 rounds
  // over its 32 KB tile and loses: 52 against 42 us, tools/agg_probe.py)
  if (n_nodes <= tb_env_int("TRUSS_GCN_SLAB_MAX_N", 128) &&
      tb_launch_gcn_slab(adj, a_batch_stride, nbr, k_nbr, h, bias, out, n_batch, n_nodes, n_channels, act, (hipStream_t)stream)) {
    hipError_t es = hipGetLastError();
    if (es != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("gcn slab aggregate launch failed: ") + hipGetErrorString(es));
    return TRUSS_OK;
  }
  const int C4 = n_channels / 4;
  const long total = (long)n_batch * n_nodes * C4;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (k_nbr <= 4)
    hipLaunchKernelGGL(truss_gcn_aggregate_sparse_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, adj, (long)a_batch_stride, nbr, k_nbr, h,
                       bias, out, total, n_nodes, C4, act);
  else if (k_nbr <= 8)
    hipLaunchKernelGGL(truss_gcn_aggregate_sparse_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, adj, (long)a_batch_stride, nbr, k_nbr, h,
                       bias, out, total, n_nodes, C4, act);
  else
    hipLaunchKernelGGL(truss_gcn_aggregate_sparse_kernel<12>, grid, dim3(256), 0, (hipStream_t)stream, adj, (long)a_batch_stride, nbr, k_nbr, h,
                       bias, out, total, n_nodes, C4, act);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("gcn sparse aggregate launch failed: ") + hipGetErrorString(e));
  return TRUSS_OK;
}

extern "C" int truss_gcn_aggregate(const float *adj, int64_t a_batch_stride, const float *h, const float *bias, float *out,
                                   int32_t n_batch, int32_t n_nodes, int32_t n_channels, int32_t act, void *stream) {
  if (!adj || !h || !out) return tb_fail(TRUSS_EINVAL, "truss_gcn_aggregate: NULL argument");
  if (n_batch < 0 || n_nodes < 1 || n_nodes > 64 || n_channels < 1 || act < 0 || act > 2)
    return tb_fail(TRUSS_EINVAL, "truss_gcn_aggregate: n_nodes must be 1..64, act 0..2");
  if (n_batch == 0) return TRUSS_OK;
  // 17..64 nodes: the slab kernel (32 nodes: 58-61 us against 73-84 for the channel-quad kernel below, 64 nodes: 44 against 182 for
  // the thread-per-channel kernel and 75 for rocBLAS + bias + activation); <= 16 nodes: the channel-quad kernel (56 against 74 us)
  if (n_nodes > tb_env_int("TRUSS_GCN_SLAB_DENSE_ABOVE", 16) && tb_launch_gcn_slab(adj, a_batch_stride, nullptr, n_nodes, h, bias, out, n_batch, n_nodes, n_channels, act, (hipStream_t)stream)) {
    hipError_t es = hipGetLastError();
    if (es != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("gcn slab aggregate launch failed: ") + hipGetErrorString(es));
    return TRUSS_OK;
  }
  if ((n_channels & 3) == 0 && n_nodes <= 32 && (((size_t)h | (size_t)out | (size_t)bias) & 15) == 0) {
    // channel-quad threads, no idle lanes
    const int C4 = n_channels / 4;
    const int gmax = a_batch_stride ? 256 / C4 + 2 : 1;
    const size_t lds = (size_t)gmax * n_nodes * n_nodes * sizeof(float);
    const long total = (long)n_batch * C4;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (lds <= 64 * 1024) {
      if (n_nodes <= 16)
        hipLaunchKernelGGL(truss_gcn_aggregate4_kernel<16>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, adj, (long)a_batch_stride, h, bias, out,
                           n_batch, n_nodes, C4, act, gmax);
      else
        hipLaunchKernelGGL(truss_gcn_aggregate4_kernel<32>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, adj, (long)a_batch_stride, h, bias, out,
                           n_batch, n_nodes, C4, act, gmax);
      hipError_t e4 = hipGetLastError();
      if (e4 != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("gcn aggregate launch failed: ") + hipGetErrorString(e4));
      return TRUSS_OK;
    }
  }
  dim3 grid((unsigned)n_batch, (unsigned)((n_channels + 255) / 256));
  if (n_nodes <= 16)
    hipLaunchKernelGGL(truss_gcn_aggregate_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, adj, (long)a_batch_stride, h, bias, out, n_nodes, n_channels, act);
  else if (n_nodes <= 32)
    hipLaunchKernelGGL(truss_gcn_aggregate_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, adj, (long)a_batch_stride, h, bias, out, n_nodes, n_channels, act);
  else
    hipLaunchKernelGGL(truss_gcn_aggregate_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, adj, (long)a_batch_stride, h, bias, out, n_nodes, n_channels, act);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("gcn aggregate launch failed: ") + hipGetErrorString(e));
  return TRUSS_OK;
}
