// truss_body.h -- the per-lane program of the batched truss environment step.
//
// One env is owned by a group of G lanes of a 64-wide wavefront (64/G envs per wave, one wave per
// workgroup).  The program is written as PHASES separated by workgroup barriers; a phase never
// needs data another lane produces inside the same phase.  The HIP kernel (truss_hip.hip) runs the
// phases with __syncthreads() in between; the build's CPU lane emulator (tests/emu) runs the very
// same phases lane by lane, which is how the indexing below is debugged without a GPU.
//
// Reference semantics implemented here (paths relative to the reference checkout, train/code/):
//   decode      Game_research04._game_modify prologue      truss2D_ENV.py:370-490 (+ test copies' symmetry)
//   move range  gen_model.set_moveRange                    truss2D_GEN.py:118-133
//   elements    Element.gen_length, Model.gen_global_k     FEM_2Dtruss.py:99-105, 284-307
//   assembly    Model.gen_ssm                              FEM_2Dtruss.py:310-324
//   solve       Model.gen_d (np.linalg.solve)              FEM_2Dtruss.py:327-337
//   post        gen_v/gen_u/gen_q/gen_f/gen_r/gen_yield    FEM_2Dtruss.py:341-431
//   objectives  _game_modify epilogue                      truss2D_ENV.py:503-525
//
// Data movement: phase_stage copies (a) the topology tables (one contiguous blob, a few KB, shared by
// all envs of the workgroup) and (b) every per-env input row (heights, x, sections, actions, targets,
// parameters) from HBM into LDS with 16-byte loads issued back to back, so a wave that sits alone on
// its SIMD pays the HBM latency once instead of once per dependent table lookup.  Everything after
// that reads LDS.
//
// Solver: K is SPD and banded once nodes are ordered along the span.  Free DOFs are permuted on the
// host (truss_host.h) to solver positions 0..n-1 with half bandwidth < W = G*RPL.  Each lane keeps
// RPL rows of the sliding W x W window of the partially factorised matrix in registers (row r lives
// in lane r mod G, register set (r mod W)/G, column c in register c mod W -- all compile-time
// indices once the pivot loop is unrolled by W).  Per pivot k: every lane posts its entry of
// column k to LDS row k (which becomes the stored L*D column for the back substitution), the group
// reads the W entries back (broadcast), applies the rank-1 update to its rows, and the lane that
// owned row k adopts row k+W from the assembled band.  L D L^T, no pivoting; a non-positive pivot
// raises the env's status flag.
#pragma once
#include <stdint.h>
#include <math.h>

#ifndef TB_SCHED_FENCE
#define TB_SCHED_FENCE()  // device: __builtin_amdgcn_sched_barrier(0); pins hand-placed prefetches
#endif
#ifndef TRUSS_HD
#error "define TRUSS_HD (e.g. __device__ __forceinline__) before including truss_body.h"
#endif

// node flag bits
#define TF_TOP 1
#define TF_RESX 2
#define TF_RESY 4
#define TF_LOAD_BRIDGE 8
#define TF_LOAD_ROOF 16

// flags (mirror include/truss_mi355.h)
#define TB_NO_DECODE 0x1u
#define TB_CLAMP_INPLACE 0x2u
#define TB_EMIT_OBS 0x4u

// 16-byte unit of the HBM<->LDS copies: a native vector type (kept in VGPRs; a struct here ended up in scratch)
#if defined(__clang__)
typedef uint32_t tb_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t tb_u2 __attribute__((ext_vector_type(2)));
typedef float tb_f4 __attribute__((ext_vector_type(4)));
typedef float tb_f2 __attribute__((ext_vector_type(2)));
typedef double tb_d2 __attribute__((ext_vector_type(2)));
#else
typedef uint32_t tb_u4 __attribute__((vector_size(16)));
typedef uint32_t tb_u2 __attribute__((vector_size(8)));
typedef float tb_f4 __attribute__((vector_size(16)));
typedef float tb_f2 __attribute__((vector_size(8)));
typedef double tb_d2 __attribute__((vector_size(16)));
#endif
// 16-byte LDS row access (ds_read_b128 / ds_write_b128): band rows are 16-byte aligned
TRUSS_HD void tb_ld_row(const double *p, double *dst, int n2) {
  const tb_d2 *q = (const tb_d2 *)__builtin_assume_aligned(p, 16);
  for (int i = 0; i < n2; ++i) {
    tb_d2 v = q[i];
    dst[2 * i] = v[0];
    dst[2 * i + 1] = v[1];
  }
}

// All topology tables live in ONE blob (global memory); f_* are byte offsets into it.  The step
// kernel copies the blob to the start of its LDS and reads the copy; the observation kernel reads
// the global blob directly.
struct TopoDev {
  int32_t N, E, NP, ndof, n_pad, n_rest, n_sym_nodes, n_sym_elems, n_sections, has_pairs;
  // solver geometry.  One team (NT = 1): KA = n_pad pivots top-down, no middle.  Two teams (NT = 2):
  // team A eliminates positions [0, KA), team B the reversed positions [0, KA) (= original n-1 .. n-KA),
  // the `mid` = n - 2 KA <= W rows in between are merged into team A's window and finished there.
  int32_t nteams, KA, mid, rowsA, rowsB, zlen, dlen, zslot;
  const char *blob;
  int32_t blob_bytes;  // multiple of 16
  int32_t f_conn;      // int16 [E][2]
  int32_t f_pairs;     // int16 [NP][2]  (lo, hi) with lo < hi
  int32_t f_nflags;    // uint8 [N]
  int32_t f_dofpos;    // int16 [N][2]   solver position of (node, comp) or -1 when restrained
  int32_t f_restslot;  // int16 [N][2]   reaction slot (reference order nsc-ndof-1) or -1
  int32_t f_asm;       // int16 [E][4]   band offsets of the element's four off-diagonal entries
                       //                (ax,bx) -cc, (ay,by) -ss, (ax,by) -cs, (ay,bx) -cs; -1 = restrained
  int32_t f_diagoff;   // int16 [N][3]   band offsets of (x,x), (y,y), (x,y) of the node block, or -1
  int32_t f_zcode;     // uint8 [nteams*zlen] load code of every z/P slot: bit0 comp (0 x, 1 y), bit1 loaded (bridge), bit2 loaded (roof)
  int32_t f_posnode;   // int16 [n_pad]  node*2+comp at solver position, -1 = padding row
  int32_t f_symn;      // int16 [n][2]   (dst, src)
  int32_t f_syme;      // int16 [n][2]
  int32_t f_area;      // double [n_sections] area in m^2
  int32_t f_isr;       // double [n_sections] 1 / (area * long_stress)
  int32_t f_adj8;      // int16 [N][8]   elements incident to each node, padded with E (a zero slot)
  int32_t f_exs;       // int16 [E][4]   slots of the element's four end displacements (n0x n0y n1x n1y) in the solution vector;
                       //                restrained DOFs point at the zero slot (one 8-byte read per element in the post phase)
  int32_t f_areaf;     // float [n_sections] (float)area                          (nN_x_e column 1, ENV:150)
  int32_t f_vsf;       // float [n_sections] (float)(area / largest area)         (A_s, ENV:86-87)
  double e_mod, long_stress;
  // LDS layout of the step kernel: [blob copy][env 0][env 1]...; o_* are byte offsets inside one env
  int32_t o_env0, env_stride;
  int32_t o_kb, o_zs, o_xsol, o_red, o_rbuf, o_par, o_y, o_x, o_tg, o_geo, o_tac, o_sec, o_ev, o_zring, o_mrg;
  // output staging rows inside the (dead) band region, each 16-byte aligned
  int32_t so_q0, so_sr, so_disp, so_mu, so_md, so_comp;
  // ---- fused observation emission (TB_EMIT_OBS, see "observation emission" in StepLane) ----
  // emit_ok: this topology/variant can write the observation tensors from the step kernel itself.
  // chunk = 16 output bytes; lane g of an env handles chunks g, g + G, ... of every tensor (= "iterations").  Tables
  // are padded by the host to the variant's compile-time iteration counts, entries past the end of a tensor repeat
  // its last chunk (the store address is clamped the same way: a duplicate rewrites the same bytes).
  //   etab (global memory -> registers):   entries of iterations (2p, 2p+1) of a lane are adjacent (one 16-byte load)
  //     et_xn / et_nxn : [pair][G][2] x 4 uint16  float offsets (from the env's LDS base) of the chunk's 4 values
  //     et_mat         : [pair][G][2] x 4 uint16  N x N matrices, row-major chunks of 4 cells: the element joining
  //                      (row, col), or E = none
  //   f_tnxe (LDS, behind the tables every kernel stages: `blob_bytes` of an EMIT launch includes it):
  //     [iter][G] x 4 uint16 float offsets, as et_xn -- when E % 4 == 0.  Otherwise the rows of nN_x_e (21 E floats per env)
  //     are not 16-byte aligned per env: the table is a plain [21 E] uint16 list and the tensor is written in chunks of
  //     nxe_cw = 2 (E even) or 1 floats
  int32_t emit_ok;
  const char *etab;
  int32_t o_flag;   // LDS byte offset of the workgroup's progress word (compute wave -> streaming wave)
  int32_t et_xn, et_nxn, et_mat, f_tnxe;
  int32_t nc_xn, nc_nxn, nc_nxe, nc_mat;   // chunks per env
  int32_t nxe_cw;                          // floats per chunk of nN_x_e: 4, 2 or 1
  // feature bank: float offsets (from the env's LDS base) of the arrays the tables point into; they live in
  // bytes of the env that are dead by the time they are written (band, solver scratch, action rows)
  int32_t b_const;   // [0] = 0, [1] = 1, [2] = 1 / (1 + 1e-6f)
  // records (LDS stores are the expensive instruction here: two / four 16-byte stores per element / node)
  int32_t b_erec;    // per element, 4 floats: sec, area, length, tension (nN_x_e columns 0 1 2 3)
  int32_t b_erec2;   // per element, 2 floats: compression, violated (nN_x_e columns 4 6)
  int32_t b_tc;      // per element, 2 floats: A_n_ts value, A_n_cs value (ENV:92-100); + one all-zero pair E ("no edge").  Written with
                     //   the member results themselves, so that the first tensors that need the solve start to stream 4 k cycles
                     //   earlier than the records (progress 2)
  int32_t b_nraw;    // per node, 4 floats: loaded, target / y, |dy|, |dy| / max_def >= 1 (features the step holds in no row)
  int32_t b_nna, b_nnb, b_nn8;   // per node: normalised x_n columns (0 1 4 7), (8 9 10 11) as 4-float records, column 12 as floats
};

#define TB_TAB(type, base, off) ((const type *)((base) + (off)))

struct StepArgsDev {
  int32_t B;
  uint32_t flags;
  const float *x, *y_in;
  const int32_t *sec_in;
  const float *mu_in, *md_in;
  float *a_geo, *a_topo;
  const uint8_t *coin;
  const float *target;
  const double *env_params;
  float *y_out;
  int32_t *sec_out;
  float *mu_out, *md_out;
  float *disp, *q0, *sr;
  uint8_t *comp;
  float *point, *obj;
  double *disp64, *q064, *energy, *react;
  int32_t *status;
  float *x_n, *A_s, *A_ts, *A_cs, *nxn, *nxe;   // observation tensors of the new design (TB_EMIT_OBS), each may be NULL
};

#define TRUSS_NRED 6  // vol, dt, con1, con2, energy, (spare)

TRUSS_HD float tb_clamp01(float v) { return v > 1.0f ? 1.0f : (v < 0.0f ? 0.0f : v); }

// Python round(np.float32, 2): rint(v*100)/100 in float32 (truss2D_ENV.py:413)
TRUSS_HD float tb_round2(float v) { return rintf(v * 100.0f) / 100.0f; }

// G lanes own one env; the solver window is WL lanes x RPL rows (W = WL*RPL columns).  With G >= 2 WL and
// one row per lane the second group of WL lanes is a second TEAM that eliminates the reversed system from
// the other end (two-sided scheme, see the solver section); any further lanes mirror lane g % WL through the
// solver (same registers, same LDS addresses) and only contribute in the per-node / per-element phases.
// EMIT instantiations additionally write the observation tensors of the new design (state_data +
// state_data_not_norm, truss2D_ENV.py:40-193, 497-500) from the same launch: "observation emission" below.
template <int G, int WL, int RPL, int EPL, bool EMIT = false>
struct StepLane {
  static constexpr int W = WL * RPL;
  static constexpr int WL_ = WL;
  static constexpr int G_ = G;
  static constexpr bool EMIT_ = EMIT;
  static constexpr int NPL = (2 * EPL + 4) / 5;   // nodes per lane unrolled (two-row grid trusses: N ~ 0.4 E)
  static constexpr int EPB = 64 / G;  // envs per wave
  static constexpr int NDEG = 8;      // unrolled node-degree bound of the diagonal gather
  static constexpr int NT = G / WL;   // solver teams per env (1, or 2 = two-sided elimination)

  int lane, g, gs, team, env, envc;
  double *Kt, *Zt, *ZRt;  // this lane's team: band rows, z vector (P_r until row r is eliminated), trash slots
  bool active;
  const char *TB;  // LDS copy of the topology blob
  char *L;         // this env's LDS region
  // per-env scalars
  double max_def, load_x, load_y;
  float ymax32, dmin32, ymd32, int1, int2;
  int is_roof, heads;
  // element stash
  double ek[EPL], ec[EPL], es[EPL], ei[EPL];
  // solver window
  double R[RPL][W];
  double rhs[RPL];
  double xs[W];
  int bad;
  // partial reductions
  double p_vol, p_dt, p_en;
  float p_c1, p_c2;

  TRUSS_HD double *kb(const TopoDev &T) { return (double *)(L + T.o_kb); }
  TRUSS_HD double *zs(const TopoDev &T) { return (double *)(L + T.o_zs); }
  TRUSS_HD double *xsol(const TopoDev &T) { return (double *)(L + T.o_xsol); }
  TRUSS_HD double *red(const TopoDev &T) { return (double *)(L + T.o_red); }
  TRUSS_HD double *rbuf(const TopoDev &T) { return (double *)(L + T.o_rbuf); }
  TRUSS_HD double *par(const TopoDev &T) { return (double *)(L + T.o_par); }
  TRUSS_HD float *ysh(const TopoDev &T) { return (float *)(L + T.o_y); }
  TRUSS_HD float *xsh(const TopoDev &T) { return (float *)(L + T.o_x); }
  TRUSS_HD float *tgsh(const TopoDev &T) { return (float *)(L + T.o_tg); }
  TRUSS_HD float *geosh(const TopoDev &T) { return (float *)(L + T.o_geo); }
  TRUSS_HD float *tac(const TopoDev &T) { return (float *)(L + T.o_tac); }
  TRUSS_HD int32_t *secsh(const TopoDev &T) { return (int32_t *)(L + T.o_sec); }
  TRUSS_HD double *evsh(const TopoDev &T) { return (double *)(L + T.o_ev); }
  TRUSS_HD const int16_t *t_conn(const TopoDev &T) const { return TB_TAB(int16_t, TB, T.f_conn); }
  TRUSS_HD const int16_t *t_pairs(const TopoDev &T) const { return TB_TAB(int16_t, TB, T.f_pairs); }
  TRUSS_HD const uint8_t *t_nflags(const TopoDev &T) const { return TB_TAB(uint8_t, TB, T.f_nflags); }
  TRUSS_HD const int16_t *t_dofpos(const TopoDev &T) const { return TB_TAB(int16_t, TB, T.f_dofpos); }
  TRUSS_HD const int16_t *t_restslot(const TopoDev &T) const { return TB_TAB(int16_t, TB, T.f_restslot); }
  TRUSS_HD const int16_t *t_asm(const TopoDev &T) const { return TB_TAB(int16_t, TB, T.f_asm); }
  TRUSS_HD const int16_t *t_diagoff(const TopoDev &T) const { return TB_TAB(int16_t, TB, T.f_diagoff); }
  TRUSS_HD const uint8_t *t_zcode(const TopoDev &T) const { return TB_TAB(uint8_t, TB, T.f_zcode); }
  TRUSS_HD const int16_t *t_posnode(const TopoDev &T) const { return TB_TAB(int16_t, TB, T.f_posnode); }
  TRUSS_HD const int16_t *t_symn(const TopoDev &T) const { return TB_TAB(int16_t, TB, T.f_symn); }
  TRUSS_HD const int16_t *t_syme(const TopoDev &T) const { return TB_TAB(int16_t, TB, T.f_syme); }
  TRUSS_HD const double *t_area(const TopoDev &T) const { return TB_TAB(double, TB, T.f_area); }
  TRUSS_HD const double *t_isr(const TopoDev &T) const { return TB_TAB(double, TB, T.f_isr); }
  TRUSS_HD const int16_t *t_adj8(const TopoDev &T) const { return TB_TAB(int16_t, TB, T.f_adj8); }
  TRUSS_HD const int16_t *t_exs(const TopoDev &T) const { return TB_TAB(int16_t, TB, T.f_exs); }

  // ------------------------------------------------------------------------------------------
  TRUSS_HD void init(int lane_, int block, const TopoDev &T, const StepArgsDev &A, char *lds) {
    lane = lane_;
    g = lane % G;
    gs = g % WL;
    team = T.nteams == 2 ? (g / WL) & 1 : 0;  // lanes beyond the team(s) mirror lanes 0..nteams*WL-1
    int grp = lane / G;
    env = block * EPB + grp;
    active = env < A.B;
    envc = active ? env : A.B - 1;
    TB = lds;
    L = lds + T.o_env0 + (size_t)grp * T.env_stride;
    Kt = (double *)(L + T.o_kb) + team * T.rowsA * W;
    Zt = (double *)(L + T.o_zs) + team * T.zlen;     // zlen is even: 16-byte aligned teams
    ZRt = (double *)(L + T.o_zring) + team * W;    // adjacent banks for the two teams
    bad = 0;
    p_vol = p_dt = p_en = 0.0;
    p_c1 = p_c2 = 0.0f;
    rs_first_step = 0;
    rs_y_out = A.y_out;
    rs_sec_out = A.sec_out;
    rs_next_geo = rs_next_topo = nullptr;
  }

  // ---- phase 0: HBM -> LDS (tables + this env's inputs), clear the band ------------------------
  // A wave sits alone on its SIMD, so the HBM latency is paid once only if every load is in flight
  // before the first dependent LDS store: all loads go to registers first (clamped indices, no
  // control flow), then all stores.  Rows whose length is not a multiple of 4 words, or topologies
  // beyond the unrolled bounds, take the plain loop (stage_row_slow).
  static constexpr int BIT = 8;                                // blob units (16 B) per lane, unrolled (8 KB of tables)
  static constexpr int BIT_BIG = 20;                           // ... for the large-truss variants (20 KB)
  // float4 units of an [N] / [E] row per lane, from what the variant is built for: E <= G * EPL, and the
  // grid families' N ~ 0.4 E (G * NPL nodes); at least the 64 nodes / 128 elements of the small variants
  static constexpr int NCAP = (G * ((2 * EPL + 4) / 5) > 64) ? G * ((2 * EPL + 4) / 5) : 64;
  static constexpr int ECAP = (G * EPL > 128) ? G * EPL : 128;
  static constexpr int NIT = (NCAP / 4 + G - 1) / G;
  static constexpr int EIT = (ECAP / 4 + G - 1) / G;
  // persistent rollout: the next step's actions, loaded while this step assembles and solves (rollout_prefetch) and parked
  // in the action rows once the merge scratch that shares their bytes is dead (rollout_stash)
  tb_u4 pg[2 * NIT], pa[3 * NIT];
  // what changes from step to step of a persistent rollout (init() takes them from the arguments: a plain step)
  int rs_first_step;
  float *rs_y_out;
  int32_t *rs_sec_out;
  const float *rs_next_geo, *rs_next_topo;
  TRUSS_HD void rollout_prefetch(const TopoDev &T, const StepArgsDev &A) {
    if (rs_next_geo) {     // wave-uniform
      const size_t bn = (size_t)envc * T.N;
      row_load<2 * NIT>(rs_next_geo + bn * 2, 2 * (T.N >> 2), pg);
      row_load<3 * NIT>(rs_next_topo + bn * 3, 3 * (T.N >> 2), pa);
    }
  }
  TRUSS_HD void rollout_stash(const TopoDev &T, const StepArgsDev &A) {
    if (rs_next_geo) {
      row_store<2 * NIT>(geosh(T), 2 * (T.N >> 2), pg);
      row_store<3 * NIT>(tac(T), 3 * (T.N >> 2), pa);
    }
  }

  TRUSS_HD void stage_row_slow(const void *src, void *dst, int nwords) const {
    const uint32_t *s1 = (const uint32_t *)src;
    uint32_t *d1 = (uint32_t *)dst;
    for (int q = g; q < nwords; q += G) d1[q] = s1[q];
  }

  template <int IT>
  TRUSS_HD void row_load(const void *src, int nq, tb_u4 (&v)[IT]) const {
    const tb_u4 *s4 = (const tb_u4 *)src;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int q = g + G * i;
      v[i] = s4[q < nq ? q : nq - 1];
    }
  }
  template <int IT>
  TRUSS_HD void row_store(void *dst, int nq, const tb_u4 (&v)[IT]) const {
    tb_u4 *d4 = (tb_u4 *)dst;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int q = g + G * i;
      d4[q < nq ? q : nq - 1] = v[i];  // clamped like the load: a duplicate writes the same value; no
    }                                  // branch, so the compiler cannot sink the load under a guard
  }

  // clear the band with 16-byte stores (padding-row identity is written after the next barrier).  The
  // band geometry comes with the kernel arguments, not with the blob being staged.
  TRUSS_HD void band_clear(const TopoDev &T) {
    tb_u4 *K4 = (tb_u4 *)kb(T);
    const tb_u4 z = {0u, 0u, 0u, 0u};
    const int tot4 = ((T.rowsA + T.rowsB) * W) >> 1;
    for (int i = g; i < tot4; i += G) K4[i] = z;
  }

  // BI = 16-byte blob units per lane (unrolled): every iteration is a global load and an LDS store for the
  // whole wave whether or not the blob reaches that far, so the common small blob gets its own instance
  template <int BI>
  TRUSS_HD void stage_fast(const TopoDev &T, const StepArgsDev &A, bool decode) {
    const size_t bn = (size_t)envc * T.N, be = (size_t)envc * T.E;
    {
      const int nb = T.blob_bytes >> 4, nn = T.N >> 2, ne = T.E >> 2;
      tb_u4 vb[BI], vy[NIT], vx[NIT], vt[NIT], vs[EIT], vp[1], vg[2 * NIT], va[3 * NIT];
      {
        const tb_u4 *s4 = (const tb_u4 *)T.blob;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
          int q = lane + 64 * i;
          vb[i] = s4[q < nb ? q : nb - 1];
        }
      }
      row_load<NIT>(A.y_in + bn, nn, vy);
      row_load<NIT>(A.x + bn, nn, vx);
      row_load<NIT>(A.target + bn, nn, vt);
      row_load<EIT>(A.sec_in + be, ne, vs);
      row_load<1>(A.env_params + (size_t)envc * 8, 4, vp);
      if (decode) {
        row_load<2 * NIT>(A.a_geo + bn * 2, 2 * nn, vg);
        row_load<3 * NIT>(A.a_topo + bn * 3, 3 * nn, va);
      }
      band_clear(T);   // independent of the loads in flight: its LDS stores overlap the HBM latency
      {
        tb_u4 *d4 = (tb_u4 *)TB;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
          int q = lane + 64 * i;
          d4[q < nb ? q : nb - 1] = vb[i];
        }
      }
      row_store<NIT>(ysh(T), nn, vy);
      row_store<NIT>(xsh(T), nn, vx);
      row_store<NIT>(tgsh(T), nn, vt);
      row_store<EIT>(secsh(T), ne, vs);
      row_store<1>(par(T), 4, vp);
      if (decode) {
        row_store<2 * NIT>(geosh(T), 2 * nn, vg);
        row_store<3 * NIT>(tac(T), 3 * nn, va);
      }
    }
  }

  TRUSS_HD void phase_stage(const TopoDev &T, const StepArgsDev &A) {
    const size_t bn = (size_t)envc * T.N, be = (size_t)envc * T.E;
    const bool decode = !(A.flags & TB_NO_DECODE);
    const bool fast = (T.N & 3) == 0 && (T.E & 3) == 0 && T.N <= NCAP && T.E <= ECAP && T.blob_bytes <= BIT_BIG * 64 * 16;
    heads = A.coin ? (A.coin[envc] != 0) : 0;
    if (rs_first_step != 0) {
      // persistent rollout, a step after the first: topology tables, constants and the design (the previous step's result)
      // are in LDS, this step's actions were parked there by rollout_stash -- nothing to fetch
      band_clear(T);
      return;
    }
    if (fast) {
      if (T.blob_bytes <= 3 * 64 * 16) stage_fast<3>(T, A, decode);
      else if (T.blob_bytes <= BIT * 64 * 16) stage_fast<BIT>(T, A, decode);
      else stage_fast<BIT_BIG>(T, A, decode);
    } else {
      {
        const tb_u4 *s4 = (const tb_u4 *)T.blob;
        tb_u4 *d4 = (tb_u4 *)TB;
        for (int q = lane; q < (T.blob_bytes >> 4); q += 64) d4[q] = s4[q];
      }
      stage_row_slow(A.y_in + bn, ysh(T), T.N);
      stage_row_slow(A.x + bn, xsh(T), T.N);
      stage_row_slow(A.target + bn, tgsh(T), T.N);
      stage_row_slow(A.sec_in + be, secsh(T), T.E);
      stage_row_slow(A.env_params + (size_t)envc * 8, par(T), 16);
      if (decode) {
        stage_row_slow(A.a_geo + bn * 2, geosh(T), 2 * T.N);
        stage_row_slow(A.a_topo + bn * 3, tac(T), 3 * T.N);
      }
      band_clear(T);
    }
  }

  TRUSS_HD void load_params(const TopoDev &T) {
    const double *P = par(T);
    double y_max = P[0], d_min = P[1];
    max_def = P[2];
    load_x = P[3];
    load_y = P[4];
    int1 = (float)P[5];
    int2 = (float)P[6];
    is_roof = P[7] != 0.0;
    ymax32 = (float)y_max;
    dmin32 = (float)d_min;
    ymd32 = (float)(y_max - d_min);  // python float arithmetic, stored to float32 (ENV:482)
  }

  // set_moveRange for one node (truss2D_GEN.py:118-133), float32
  TRUSS_HD void move_range(bool top, float y, float yp, float &up, float &dn) const {
    float upt = fabsf(ymax32 - y), dnt = fabsf((y - yp) - dmin32);
    float upb = is_roof ? fabsf((yp - y) - dmin32) : 0.0f, dnb = is_roof ? fabsf(y) : 0.0f;
    up = top ? upt : upb;
    dn = top ? dnt : dnb;
  }

  // geometry move + support + round for one node (truss2D_ENV.py:398-413)
  TRUSS_HD float move_node(float y, float g0, float g1, float mu, float md, bool resy) const {
    bool down = g1 > g0;  // np.argmax: first maximum wins
    float a = down ? g1 : g0;
    float rng = down ? md : mu;
    float step = (a * rng) * 0.25f;
    y = down ? (y - step) : (y + step);
    y = tb_round2(y);
    return resy ? 0.0f : y;
  }

  // ---- phase 1: action decode for the lane's node pairs (all operands in LDS) ----
  // One vertical pair = one unit of work (the repairs of truss2D_ENV.py:469-490 couple the two nodes).  The first PPL pairs of
  // a lane are straight-line code: table reads, operand reads, float32 arithmetic with selects (the reference's nested ifs, in
  // their order), stores -- a pair index past the last pair redoes the arithmetic of pair NP-1 and stores nothing but the
  // (idempotent) clamped actions.
  struct PairIn {
    int n0, n1, f0, f1;
    float y0, y1, ga0, ga1, gb0, gb1, ta[3], tb[3];
  };
  TRUSS_HD void pair_load(const TopoDev &T, int p, PairIn &q) {
    const float *Y = ysh(T), *GE = geosh(T), *TA = tac(T);
    const int16_t *PR = t_pairs(T);
    const uint8_t *NF = t_nflags(T);
    q.n0 = PR[2 * p];
    q.n1 = PR[2 * p + 1];
    q.y0 = Y[q.n0];
    q.y1 = Y[q.n1];
    q.f0 = NF[q.n0];
    q.f1 = NF[q.n1];
    q.ga0 = GE[2 * q.n0];
    q.ga1 = GE[2 * q.n0 + 1];
    q.gb0 = GE[2 * q.n1];
    q.gb1 = GE[2 * q.n1 + 1];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      q.ta[j] = TA[3 * q.n0 + j];
      q.tb[j] = TA[3 * q.n1 + j];
    }
  }
  TRUSS_HD void pair_decode(const TopoDev &T, const StepArgsDev &A, PairIn &q, bool real) {
    const size_t bn = (size_t)envc * T.N;
    float *Y = ysh(T), *TA = tac(T);
    const int n0 = q.n0, n1 = q.n1;
    const bool top0 = q.f0 & TF_TOP, top1 = q.f1 & TF_TOP;
    const float ga0 = tb_clamp01(q.ga0), ga1 = tb_clamp01(q.ga1), gb0 = tb_clamp01(q.gb0), gb1 = tb_clamp01(q.gb1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      q.ta[j] = tb_clamp01(q.ta[j]);
      q.tb[j] = tb_clamp01(q.tb[j]);
      TA[3 * n0 + j] = q.ta[j];
      TA[3 * n1 + j] = q.tb[j];
    }
    if ((A.flags & TB_CLAMP_INPLACE) && active && real) {  // truss2D_ENV.py:376-388 mutates the caller's arrays (wave-uniform flag)
      A.a_geo[(bn + n0) * 2 + 0] = ga0;
      A.a_geo[(bn + n0) * 2 + 1] = ga1;
      A.a_geo[(bn + n1) * 2 + 0] = gb0;
      A.a_geo[(bn + n1) * 2 + 1] = gb1;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        A.a_topo[(bn + n0) * 3 + j] = q.ta[j];
        A.a_topo[(bn + n1) * 3 + j] = q.tb[j];
      }
    }
    float mu0, md0, mu1, md1;
    if (A.mu_in) {   // wave-uniform pointer test
      mu0 = A.mu_in[bn + n0];
      md0 = A.md_in[bn + n0];
      mu1 = A.mu_in[bn + n1];
      md1 = A.md_in[bn + n1];
    } else {
      move_range(top0, q.y0, q.y1, mu0, md0);
      move_range(top1, q.y1, q.y0, mu1, md1);
    }
    float a = move_node(q.y0, ga0, ga1, mu0, md0, q.f0 & TF_RESY);
    float b = move_node(q.y1, gb0, gb1, mu1, md1, q.f1 & TF_RESY);
    // sequential repairs (truss2D_ENV.py:469-490): visit lo then hi in each of the three loops; every statement sees the values the
    // previous one left (selects, no branches)
    // loop 1: below y_min (= 0):   top node -> d_min and its partner to 0; bottom node -> 0
    const bool c1 = a < 0.0f;
    a = c1 ? (top0 ? dmin32 : 0.0f) : a;
    b = (c1 && top0) ? 0.0f : b;
    const bool c2 = b < 0.0f;
    b = c2 ? (top1 ? dmin32 : 0.0f) : b;
    a = (c2 && top1) ? 0.0f : a;
    // loop 2: above y_max:   top node -> y_max; bottom node -> y_max - d_min and its partner to y_max
    const bool c3 = a > ymax32;
    a = c3 ? (top0 ? ymax32 : ymd32) : a;
    b = (c3 && !top0) ? ymax32 : b;
    const bool c4 = b > ymax32;
    b = c4 ? (top1 ? ymax32 : ymd32) : b;
    a = (c4 && !top1) ? ymax32 : a;
    // loop 3: pair closer than d_min -> lift the top node
    a = (top0 && fabsf(a - b) < dmin32) ? b + dmin32 : a;
    b = (top1 && fabsf(b - a) < dmin32) ? a + dmin32 : b;
    if (real) {     // in-place update of the heights: a duplicate must not store (clamping the actions twice is idempotent)
      Y[n0] = a;
      Y[n1] = b;
    }
  }
  TRUSS_HD void phase_decode(const TopoDev &T, const StepArgsDev &A) {
    load_params(T);
    if (A.flags & TB_NO_DECODE) return;
    constexpr int PPL = (NPL + 1) / 2;
    PairIn q[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      const int p = g + G * i;
      pair_load(T, p < T.NP ? p : T.NP - 1, q[i]);
    }
#pragma unroll
    for (int i = 0; i < PPL; ++i) pair_decode(T, A, q[i], g + G * i < T.NP);
    for (int p = g + G * PPL; p < T.NP; p += G) {     // topologies with more pairs per lane than the unrolled part covers
      PairIn r;
      pair_load(T, p, r);
      pair_decode(T, A, r, true);
    }
  }

  // ---- phase 2 (only when the topology has mirror tables): node symmetry ----
  TRUSS_HD void phase_sym_nodes(const TopoDev &T) {
    float *Y = ysh(T);
    const int16_t *SN = t_symn(T);
    for (int i = g; i < T.n_sym_nodes; i += G) {
      int dst = SN[2 * i], src = SN[2 * i + 1];
      float v = heads ? Y[src] : Y[dst];
      Y[dst] = v;
      Y[src] = v;
    }
  }

  // ---- phase 3: section update (truss2D_ENV.py:421-466), in place in LDS ----
  // Straight-line: connectivity + current sections of the lane's EPL elements, then the six action values of each, then the
  // argmax / +-1 / clamp, then the (guarded) stores: the loads of all elements are in flight together.
  TRUSS_HD void phase_sizing(const TopoDev &T, const StepArgsDev &A) {
    if (A.flags & TB_NO_DECODE) return;
    const float *TA = tac(T);
    int32_t *S = secsh(T);
    const int16_t *CN = t_conn(T);
    int n0[EPL], n1[EPL], sc[EPL];
    float a0[EPL][3], a1[EPL][3];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const int e = g + G * i;
      const int ee = e < T.E ? e : T.E - 1;
      const uint32_t c = *(const uint32_t *)(CN + 2 * ee);    // both end nodes: one 4-byte read
      n0[i] = (int)(int16_t)(c & 0xffffu);
      n1[i] = (int)(int16_t)(c >> 16);
      sc[i] = S[ee];
    }
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        a0[i][j] = TA[n0[i] * 3 + j];
        a1[i][j] = TA[n1[i] * 3 + j];
      }
    }
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const float p0 = a0[i][0] + a1[i][0], p1 = a0[i][1] + a1[i][1], p2 = a0[i][2] + a1[i][2];
      int am = 0;
      float best = p0;
      if (p1 > best) { am = 1; best = p1; }
      if (p2 > best) { am = 2; }
      const int s = sc[i];
      const int sdn = s > 0 ? s - 1 : 0, sup = s < T.n_sections - 1 ? s + 1 : T.n_sections - 1;
      sc[i] = am == 0 ? sdn : (am == 1 ? sup : s);
    }
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const int e = g + G * i;
      if (e < T.E) S[e] = sc[i];     // in-place update: a duplicate must not store (it would be harmless in lock-step only)
    }
  }

  TRUSS_HD void phase_sym_elems(const TopoDev &T) {
    int32_t *S = secsh(T);
    const int16_t *SE = t_syme(T);
    for (int i = g; i < T.n_sym_elems; i += G) {
      int a = SE[2 * i], b = SE[2 * i + 1];
      int m = S[a] < S[b] ? S[a] : S[b];
      S[a] = m;
      S[b] = m;
    }
  }

  // load on the DOF at ORIGINAL solver position r (0 for padding / out of range)
  TRUSS_HD double load_at(const TopoDev &T, int r) const {
    if (r < 0 || r >= T.ndof) return 0.0;
    int nd = t_posnode(T)[r];
    if (nd < 0) return 0.0;
    int fl = t_nflags(T)[nd >> 1] & (is_roof ? TF_LOAD_ROOF : TF_LOAD_BRIDGE);
    if (!fl) return 0.0;
    return (nd & 1) ? load_y : load_x;
  }
  // original solver position of team-frame position p (team B works on the reversed system)
  TRUSS_HD int orig_pos(const TopoDev &T, int tm, int p) const { return tm ? T.ndof - 1 - p : p; }
  // right-hand side a team holds for its frame position p: team A owns its part and the middle rows,
  // team B only its own part (its middle rows start from zero and only collect Schur updates)
  TRUSS_HD double team_load(const TopoDev &T, int tm, int p) const {
    if (T.nteams == 1) return load_at(T, p);
    if (tm == 0) return p < T.ndof - T.KA ? load_at(T, p) : 0.0;
    return p < T.KA ? load_at(T, T.ndof - 1 - p) : 0.0;
  }

  // ---- phase 4: element stiffness + scatter-add into the LDS band; load vector ----
  TRUSS_HD void phase_elements(const TopoDev &T, const StepArgsDev &A) {
    const float *Y = ysh(T), *X = xsh(T);
    const int32_t *S = secsh(T);
    const int16_t *CN = t_conn(T);
    const double *AR = t_area(T), *ISR = t_isr(T);
    double *K = kb(T);
    double kcc[EPL], kcs[EPL], kss[EPL];
    // pass 1: gather + arithmetic for all of the lane's elements (no control flow: the square roots
    // and divisions of different elements overlap)
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const int e = g + G * i;
      const int ee = e < T.E ? e : T.E - 1;
      const int n0 = CN[2 * ee], n1 = CN[2 * ee + 1];
      const int sc = S[ee];
      const double dx = (double)X[n1] - (double)X[n0];
      const double dy = (double)Y[n1] - (double)Y[n0];
      const double l2 = dx * dx + dy * dy;
      const double rl = tb_rsqrt(l2);  // 1/L: v_rsq_f64 seed + Newton (no f64 sqrt/div sequences)
      const double len = l2 * rl;
      const double c = dx * rl, s = dy * rl;
      const double Aa = AR[sc];
      const double k = (T.e_mod * Aa) * rl;
      ek[i] = k;
      ec[i] = c;
      es[i] = s;
      ei[i] = ISR[sc];
      if constexpr (EMIT) el[i] = (float)len;   // nN_x_e column 2 (ENV:151)
      if (e < T.E) p_vol += (double)(float)(Aa * len);  // all_v[i] is a float32 store (ENV:508)
      const double kc = k * c, ks = k * s;
      kcc[i] = kc * c;
      kcs[i] = kc * s;
      kss[i] = ks * s;
    }
    // pass 2 (FEM_2Dtruss.py:320-324 restricted to the lower band, without atomics): the four
    // off-diagonal entries of an element belong to that element alone -> plain stores; the element's
    // (k cc, k cs, k ss) go to LDS for the node-diagonal gather of phase_assemble_nodes.
    const int16_t *AC = t_asm(T);
    double *EV = evsh(T);
    if (g == 0) EV[3 * T.E] = EV[3 * T.E + 1] = EV[3 * T.E + 2] = 0.0;  // slot the padded adjacency points at
    // branch-free: a lane index past the last element redoes element E-1 (same values to the same
    // places), entries on restrained DOFs go to a trash slot behind the band
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const int e = g + G * i;
      const int ee = e < T.E ? e : T.E - 1;
      EV[3 * ee + 0] = kcc[i];
      EV[3 * ee + 1] = kcs[i];
      EV[3 * ee + 2] = kss[i];
      const int16_t *code = AC + ee * 4;
      K[code[0]] = -kcc[i];
      K[code[1]] = -kss[i];
      K[code[2]] = -kcs[i];
      K[code[3]] = -kcs[i];
    }
  }

  // ---- phase 5: node-diagonal 2x2 blocks = sum over incident elements in ascending element order
  // (deterministic).  The adjacency is padded to NDEG entries per node so the gather is straight-line
  // code and its LDS reads overlap.
  TRUSS_HD void phase_assemble_nodes(const TopoDev &T) {
    const double *EV = evsh(T);
    const int16_t *AD = t_adj8(T), *DO = t_diagoff(T);
    double *K = kb(T);
    // the first NPL nodes of a lane, straight-line: the adjacency rows (one 16-byte read per node) and the band offsets of all of
    // them, then the incident elements' values in two rounds of four per node -- the loads of a round are in flight together, the
    // sums keep the ascending element order --, then the stores.  (With one node after the other the loads of the second node
    // could not pass the band stores of the first: K and the element values are both LDS doubles to the compiler.)  A node index
    // past the last node redoes node N-1 (same values to the same places).
    static_assert(NDEG == 8, "adjacency rows are read as one 16-byte unit");
    tb_u4 adj[NPL];
    int dof[NPL][3];
    double cc[NPL], cs[NPL], ss[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int n0 = g + G * i, n = n0 < T.N ? n0 : T.N - 1;
      adj[i] = *(const tb_u4 *)(AD + n * NDEG);
      dof[i][0] = DO[3 * n];
      dof[i][1] = DO[3 * n + 1];
      dof[i][2] = DO[3 * n + 2];
      cc[i] = cs[i] = ss[i] = 0.0;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double ev[NPL][4][3];
#pragma unroll
      for (int i = 0; i < NPL; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const uint32_t w = adj[i][2 * h + (a >> 1)];
          const int e = (a & 1) ? (int)(w >> 16) : (int)(w & 0xffffu);
          ev[i][a][0] = EV[3 * e + 0];
          ev[i][a][1] = EV[3 * e + 1];
          ev[i][a][2] = EV[3 * e + 2];
        }
#pragma unroll
      for (int i = 0; i < NPL; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          cc[i] += ev[i][a][0];
          cs[i] += ev[i][a][1];
          ss[i] += ev[i][a][2];
        }
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      K[dof[i][0]] = cc[i];      // restrained DOFs point at the trash slot
      K[dof[i][1]] = ss[i];
      K[dof[i][2]] = cs[i];
    }
    for (int n = g + G * NPL; n < T.N; n += G) {     // topologies with more nodes per lane than the unrolled part covers
      double c2 = 0.0, s2 = 0.0, x2 = 0.0;
#pragma unroll
      for (int a = 0; a < NDEG; ++a) {
        const int e = AD[n * NDEG + a];
        c2 += EV[3 * e + 0];
        x2 += EV[3 * e + 1];
        s2 += EV[3 * e + 2];
      }
      K[DO[3 * n]] = c2;
      K[DO[3 * n + 1]] = s2;
      K[DO[3 * n + 2]] = x2;
    }
    // identity rows: team A beyond its part + middle; team B beyond the middle (its middle rows keep a
    // zero diagonal: they are never pivots of team B)
    const int idA = T.nteams == 1 ? T.ndof : T.ndof - T.KA;
    for (int r = idA + g; r < T.rowsA; r += G) K[r * W + r % W] = 1.0;
    for (int r = T.KA + T.mid + g; r < T.rowsB; r += G) K[(T.rowsA + r) * W + r % W] = 1.0;
  }

  // ---- phase 5b: the element-value buffer is dead; its bytes become the solver scratch ----
  TRUSS_HD void solver_scratch_init(const TopoDev &T) {
    double *Z = zs(T);
    const uint8_t *ZC = t_zcode(T);
    const int lbit = is_roof ? 4 : 2;
    {   // load vector P in each team's frame (host table); unrolled with clamped indices like the node loops
      const int zt = T.zlen * T.nteams;
      constexpr int ZIT = 6;
#pragma unroll
      for (int i = 0; i < ZIT; ++i) {
        const int r0 = g + G * i, r = r0 < zt ? r0 : zt - 1;
        const int c = ZC[r];
        Z[r] = (c & lbit) ? ((c & 1) ? load_y : load_x) : 0.0;
      }
      for (int r = g + G * ZIT; r < zt; r += G) {
        const int c = ZC[r];
        Z[r] = (c & lbit) ? ((c & 1) ? load_y : load_x) : 0.0;
      }
    }
    tb_d2 *X2 = (tb_d2 *)__builtin_assume_aligned(xsol(T), 16);
    const tb_d2 z2 = {0.0, 0.0};
    for (int r = g; r < (T.zslot + 2) / 2; r += G) X2[r] = z2;  // incl. the zero slot restrained DOFs read
    double *RB = rbuf(T);
    for (int i = g; i < T.n_rest; i += G) RB[i] = 0.0;
  }

  // ---- solver -------------------------------------------------------------------------------
  // K x = P with K SPD and banded (half-bandwidth < W), as L D L^T without square roots:
  //   factorisation + forward substitution: pivot_write / pivot_update (one LDS exchange per pivot: every
  //     lane posts its entry of column k, all read the column back; rows enter the window from
  //     registers fetched once per block of W pivots, factor_rows_fetch);
  //   two-sided scheme (nteams == 2): team A eliminates rows 0..KA-1 of the system as ordered by the host,
  //     team B the last KA rows of the reversed system, merge_post / merge_take fold B's Schur complement
  //     into A's window, A factorises the middle block;
  //   back substitution, distributed: backsub_rows_fetch / backsub_step / backsub_share (owner lanes, DPP
  //     broadcast of x_k), outwards from the middle in both teams (backsub_flush_mid / backsub_reload hand
  //     the middle solutions to team B through the LDS).
  // Every routine below works in the lane's TEAM FRAME (Kt / Zt / ZRt): the code is identical for both
  // teams; only the LDS bases differ.  LDS stores are the expensive instruction here (the store path is
  // shared by the four waves of a CU, ~40 cycles per wave-instruction under load),
  // which is why nothing is stored that can stay in registers or be recomputed.
  TRUSS_HD void solver_init(const TopoDev &T) {
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
      int r = gs + WL * s;
#pragma unroll
      for (int c = 0; c < W; ++c) R[s][c] = (c <= r) ? Kt[r * W + c] : Kt[c * W + r];  // band rows are stored by column residue
      rhs[s] = Zt[r];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) xs[j] = 0.0;
  }

  // post this lane's entries of pivot column k (kk = k mod W, compile-time after unrolling).  The pivot
  // lane's right-hand side z_k does not go through the LDS: it reaches the team by a DPP row broadcast in
  // pivot_update (an LDS store costs ~40 cycles per wave-instruction when the four waves of the CU store
  // together), and the lane keeps it for the back substitution: one store of the kept values per block of W
  // pivots (factor_z_store) instead of one per pivot.
  double zkeep[RPL];
  TRUSS_HD void pivot_write(const TopoDev &T, int k, int kk) {
#pragma unroll
    for (int s = 0; s < RPL; ++s) Kt[k * W + gs + WL * s] = R[s][kk];
    bx = rhs[kk / WL];
  }
  // rows kb + (gs + WL s - c) mod W for a block that starts at window slot c (c = 0 for the clean blocks)
  TRUSS_HD void factor_z_store(const TopoDev &T, int kb, int c, int cnt) {
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
      const int off = (gs + WL * s - c + W) % W;
      double *zp = off < cnt ? Zt + kb + off : ZRt + gs;     // lanes without a pivot in this (partial) block: trash slot
      *zp = zkeep[s];
    }
  }

  TRUSS_HD void pivot_update(const TopoDev &T, int k, int kk) {
    double col[W];
    {  // this lane's entries of the entering column k+W: static data, requested before the posted column
      const double *Kn = Kt + (k + W) * W;
#pragma unroll
      for (int s = 0; s < RPL; ++s) ecol[s] = Kn[gs + WL * s];
    }
#pragma unroll
    for (int i = 0; i < W / 2; ++i) {
      tb_d2 v = ((const tb_d2 *)__builtin_assume_aligned(Kt + k * W, 16))[i];
      col[2 * i] = v[0];
      col[2 * i + 1] = v[1];
    }
    const double zk = tb_team_bcast(*this, kk % WL);     // the pivot lane's bx = its right-hand side z_k
    if (gs == kk % WL) zkeep[kk / WL] = bx;
    const double d = col[kk];
    const double inv = tb_rcp(d);  // d <= 0 / NaN is detected after the loop (pivot_check), off the chain
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
      double l = R[s][kk] * inv;
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (j != kk) R[s][j] = fma(-l, col[j], R[s][j]);
      rhs[s] = fma(-l, zk, rhs[s]);
    }
    // the window slides: column k leaves, column k+W enters; the pivot row's lane adopts row k+W, which
    // it requested at the start of this block of W pivots (factor_rows_fetch).  Band rows are stored by
    // column residue (entry (r, c) at r*W + c%W): a row maps straight onto the window registers and every
    // other lane finds its entry of the new column at its own residue.
#pragma unroll
    for (int s = 0; s < RPL; ++s) R[s][kk] = ecol[s];
    if (gs == kk % WL) {
      const int sp = kk / WL;
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (j != kk) R[sp][j] = fr[sp][j];   // slot kk already holds the new row's diagonal: it came in as ecol
      rhs[sp] = frhs[sp];
    }
  }
  // Rows kb+W .. kb+2W-1 enter the window during the W pivots of block kb, one per pivot, each into the
  // lane that owned the eliminated row.  They are static in the LDS until then, so every lane fetches its
  // successor rows once per block (4 reads per block and lane instead of 4 per pivot and wave, and their
  // latency is off the pivot chain).
  double fr[RPL][W], frhs[RPL], ecol[RPL];
  TRUSS_HD void factor_rows_fetch(const TopoDev &T, int kb) {
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
      const int rn = kb + W + gs + WL * s;
      const tb_d2 *K2 = (const tb_d2 *)__builtin_assume_aligned(Kt + rn * W, 16);
#pragma unroll
      for (int i = 0; i < W / 2; ++i) {
        tb_d2 v = K2[i];
        fr[s][2 * i] = v[0];
        fr[s][2 * i + 1] = v[1];
      }
      frhs[s] = Zt[rn];
    }
  }

  // K not SPD <=> some pivot d <= 0 (or NaN / inf).  Each lane scans a share of its team's pivots once,
  // after the factorisation: d_k sits in the posted column k at residue k mod W.  (Team B's idle pivots
  // k >= KA are not looked at.)
  TRUSS_HD void pivot_check(const TopoDev &T) {
    const int lim = (T.nteams == 2 && team == 1) ? T.KA : (T.nteams == 2 ? T.KA + W : T.KA);
    for (int k = gs; k < lim; k += WL) {
      const double d = Kt[k * W + k % W];
      if (!(d > 0.0) || !(d < 1.7e308)) bad = 1;
    }
  }

  // ---- two-sided elimination: fold team B's window (Schur updates of the middle rows) into team A's
  // frame row held by this lane's window slot (RPL == 1): KA + ((gs - KA) mod W)
  TRUSS_HD int win_row(const TopoDev &T, int slot) const { return T.KA + ((slot - T.KA % W + W) % W); }
  TRUSS_HD void merge_post(const TopoDev &T) {
    if (team == 1) {
      double *M = (double *)(L + T.o_mrg);
#pragma unroll
      for (int j = 0; j < W; ++j) M[gs * W + j] = R[0][j];
      M[W * W + gs] = rhs[0];
    }
  }
  TRUSS_HD void merge_take(const TopoDev &T) {
    if (team == 0) {
      const double *M = (const double *)(L + T.o_mrg);
      const int lim = T.ndof - T.KA;           // frame-A rows/cols >= lim belong to team B's part
      const int p = win_row(T, gs);
      const int pb = T.ndof - 1 - p;           // the same DOF in team B's frame
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const int c = win_row(T, j);
        const int cb = T.ndof - 1 - c;
        const bool ok = p < lim && c < lim;
        const double add = M[ok ? (pb % W) * W + (cb % W) : 0];
        R[0][j] += ok ? add : 0.0;
      }
      const double az = M[W * W + (p < lim ? pb % W : 0)];
      rhs[0] += p < lim ? az : 0.0;
    }
  }

  // back substitution, distributed over the lanes of the team:
  //   x_k = (z_k - sum_m A[k+m,k] x_{k+m}) / d_k
  // Row k of the factor (the posted pivot column k with the pivot d_k in it, and z_k) is needed by ONE dot product.  Each
  // lane therefore loads only the rows it owns -- row kb + gs (+ WL s) of a block of W steps, one set of
  // loads per block instead of one per step and lane -- evaluates every step's dot product with its own
  // row (only the owner's result means anything), and the owner's x_k reaches the other lanes of the
  // team through a DPP row broadcast: registers only, no LDS round trip in the chain of dependent
  // solutions.  (All lanes reading every row was 5 ds_read_b128 per step and wave: the LDS, shared by the
  // four waves of a CU, was the bottleneck of this phase.)
  //  * the terms are accumulated oldest-x first in two partial sums: only one fma, one multiply and the
  //    broadcast depend on the x_{k+1} the previous step has just produced;
  //  * the rows of the next block are requested while the current block is being solved.
  double bc[RPL][W], bz[RPL], bd[RPL];   // rows owned in the current block
  double nc[RPL][W], nz[RPL], nd[RPL];   // ... in the next block (in flight); nd = the pivot d itself
  double bx;                             // this lane's candidate for x_k (valid in the owner lane)
  double myx[RPL] = {};                  // x of the rows this lane owns in the current block
  TRUSS_HD void backsub_rows_fetch(const TopoDev &T, int kb) {
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
      const int p = kb + gs + WL * s;
      const int pc = p < 0 ? 0 : p;
      const tb_d2 *K2 = (const tb_d2 *)__builtin_assume_aligned(Kt + pc * W, 16);
#pragma unroll
      for (int i = 0; i < W / 2; ++i) {
        tb_d2 v = K2[i];
        nc[s][2 * i] = v[0];
        nc[s][2 * i + 1] = v[1];
      }
      nz[s] = Zt[pc];
      nd[s] = Kt[pc * W + pc % W];   // the pivot d_p (a register array must not be indexed by the lane id)
    }
  }
  TRUSS_HD void backsub_rows_adopt() {
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
#pragma unroll
      for (int j = 0; j < W; ++j) bc[s][j] = nc[s][j];
      bz[s] = nz[s];
      // 1/d_p of the owned row: the same operations as in the factorisation, so the same bits; once per block and lane
      bd[s] = tb_rcp(nd[s]);
    }
  }
  TRUSS_HD void backsub_step(const TopoDev &T, int k, int kk) {
    const int s = kk / WL;
    double acc0 = bz[s], acc1 = 0.0;
#pragma unroll
    for (int m = W - 1; m >= 2; --m) {  // rows k+m, m = W-1 .. 2 (older solutions)
      const int j = (kk + m) % W;
      if (m & 1) acc1 = fma(-bc[s][j], xs[j], acc1);
      else acc0 = fma(-bc[s][j], xs[j], acc0);
    }
    const int jn = (kk + 1) % W;        // row k+1: the newest solution
    double acc = fma(-bc[s][jn], xs[jn], acc0 + acc1);
    bx = acc * bd[s];
  }
  TRUSS_HD void backsub_share(int kk) {
    xs[kk] = tb_team_bcast(*this, kk % WL);
    if (gs == kk % WL) myx[kk / WL] = bx;   // the owner keeps x_k for the block's store
  }
  // after a block of W steps the owners store their solutions (team-frame rows kb + gs + WL s).  Team B's
  // rows of the middle block (>= KA) were solved by team A and are not written by team B.
  TRUSS_HD bool owns_row(const TopoDev &T, int p) const {
    if (T.nteams == 1) return true;
    return team == 0 ? p < T.ndof - T.KA : p < T.KA;   // A: its part + the middle; B: its part
  }
  TRUSS_HD void backsub_flush(const TopoDev &T, int kb) {
    double *XS = xsol(T);
#pragma unroll
    for (int s = 0; s < RPL; ++s) {   // every lane stores the rows it owns: one store per block, not W
      const int p = kb + gs + WL * s;
      XS[owns_row(T, p) ? orig_pos(T, team, p) : T.zslot + 1] = myx[s];  // rows of the other team: dummy slot
    }
  }
  // mid-block variant used right after the middle block: lane gs owns the window row win_row(gs)
  TRUSS_HD void backsub_flush_mid(const TopoDev &T) {
    double *XS = xsol(T);
    const int p = win_row(T, gs);
    XS[(team == 0 && owns_row(T, p)) ? p : T.zslot + 1] = myx[0];
  }
  // team B picks up the middle solutions (in its own frame) before it continues outwards
  TRUSS_HD void backsub_reload(const TopoDev &T) {
    const double *XS = xsol(T);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int o = orig_pos(T, team, win_row(T, j));
      xs[j] = XS[(o >= 0 && o < T.ndof) ? o : T.zslot];
    }
  }

  // Output staging: after the back substitution the band region is dead; the per-env result rows are
  // built there and streamed to HBM as whole rows with 16-byte stores (phase_store).
  //   [q0 E f32][sr E f32][disp 2N f32][mu N f32][md N f32][comp E u8]
  TRUSS_HD float *oq0(const TopoDev &T) { return (float *)(L + T.o_kb + T.so_q0); }
  TRUSS_HD float *osr(const TopoDev &T) { return (float *)(L + T.o_kb + T.so_sr); }
  TRUSS_HD float *odisp(const TopoDev &T) { return (float *)(L + T.o_kb + T.so_disp); }
  TRUSS_HD float *omu(const TopoDev &T) { return (float *)(L + T.o_kb + T.so_mu); }
  TRUSS_HD float *omd(const TopoDev &T) { return (float *)(L + T.o_kb + T.so_md); }
  TRUSS_HD uint8_t *ocomp(const TopoDev &T) { return (uint8_t *)(L + T.o_kb + T.so_comp); }

  // ---- phase 6: member forces, stress ratios, reactions (FEM_2Dtruss.py:341-431) ----
  // Straight-line over the lane's EPL elements: every table read first, then every solution read, then the arithmetic, then the
  // stores -- the three levels of dependent LDS reads of different elements overlap (a loop body with guards ran them one
  // element after the other: 7 k of a workgroup's 43 k cycles).  An index past the last element redoes element E-1 (same
  // values to the same places; max is idempotent).  The optional float64 / reaction outputs run behind, under wave-uniform
  // guards, from the values kept in registers.
  TRUSS_HD void phase_post_elements(const TopoDev &T, const StepArgsDev &A) {
    const double *XS = xsol(T);
    const int16_t *EX = t_exs(T);
    float *Q = oq0(T), *SR = osr(T);
    uint8_t *CP = ocomp(T);
    tb_u2 ex[EPL];
    double qv[EPL];
    float srv[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const int e = g + G * i;
      const int ee = e < T.E ? e : T.E - 1;
      ex[i] = *(const tb_u2 *)(EX + 4 * ee);
    }
    double v[EPL][4];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      v[i][0] = XS[ex[i][0] & 0xffffu];
      v[i][1] = XS[ex[i][0] >> 16];
      v[i][2] = XS[ex[i][1] & 0xffffu];
      v[i][3] = XS[ex[i][1] >> 16];
    }
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const double c = ec[i], s = es[i], k = ek[i];
      const double u0 = c * v[i][0] + s * v[i][1];
      const double u2 = c * v[i][2] + s * v[i][3];
      const double q = k * u0 + (-k) * u2;
      qv[i] = q;
      srv[i] = (float)(fabs(q) * ei[i]);  // |q/A| / long_stress (FEM:419,429)
      p_c1 = fmaxf(p_c1, fabsf(srv[i]));
    }
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const int e = g + G * i;
      const int ee = e < T.E ? e : T.E - 1;
      Q[ee] = (float)qv[i];
      SR[ee] = srv[i];
      CP[ee] = qv[i] > 0.0 ? 1 : 0;
      if constexpr (EMIT) {   // the two edge values of A_n_ts / A_n_cs (ENV:92-100): all that the first post-solve tensors need
        const bool cmp = qv[i] > 0.0;
        const float val = fminf(srv[i], 1.0f) * (srv[i] > 1.0f ? 1.0f : 0.5f);
        const tb_f2 tc = {cmp ? 0.0f : val, cmp ? val : 0.0f};
        ((tb_f2 *)((float *)L + T.b_tc))[ee] = tc;
        pe_sr[i] = srv[i];
        pe_cmp = i == 0 ? (cmp ? 1u : 0u) : (pe_cmp | (cmp ? 1u << i : 0u));
      }
    }
    if constexpr (EMIT) {
      if (g == 0) {
        const tb_f2 z = {0.0f, 0.0f};
        ((tb_f2 *)((float *)L + T.b_tc))[T.E] = z;                      // "no edge"
      }
    }
    if (A.q064 && active) {      // wave-uniform pointer test
      const size_t be = (size_t)envc * T.E;
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        const int e = g + G * i;
        if (e < T.E) A.q064[be + e] = qv[i];
      }
    }
    if (A.react) {
      // f = T^T q = q0 * [c, s, -c, -s] summed into the restrained DOFs (FEM:389-411)
      const int16_t *CN = t_conn(T), *RS = t_restslot(T);
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        const int e = g + G * i;
        if (e < T.E) {
          const int n0 = CN[2 * e], n1 = CN[2 * e + 1];
          const double q = qv[i], c = ec[i], s = es[i];
          const int s0 = RS[2 * n0], s1 = RS[2 * n0 + 1];
          const int s2 = RS[2 * n1], s3 = RS[2 * n1 + 1];
          if (s0 >= 0) tb_lds_add(&rbuf(T)[s0], q * c);
          if (s1 >= 0) tb_lds_add(&rbuf(T)[s1], q * s);
          if (s2 >= 0) tb_lds_add(&rbuf(T)[s2], q * (-c));
          if (s3 >= 0) tb_lds_add(&rbuf(T)[s3], q * (-s));
        }
      }
    }
  }

  // ---- phase 7: nodal results, move ranges of the new design, objective partials ----
  // Straight-line like phase 6: table reads, solution reads, arithmetic with selects instead of branches, stores.
  TRUSS_HD void phase_post_nodes(const TopoDev &T, const StepArgsDev &A) {
    const double *XS = xsol(T);
    const float *Y = ysh(T), *TG = tgsh(T);
    const int16_t *DP = t_dofpos(T), *PR = t_pairs(T);
    const uint8_t *NF = t_nflags(T);
    const size_t bn = (size_t)envc * T.N;
    const int zslot = T.zslot;
    float *DS = odisp(T), *MU = omu(T), *MD = omd(T);
    int px[NPL], py[NPL], fl[NPL];
    float yv[NPL], tg[NPL];
    double dxv[NPL], dyv[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {       // a clamped duplicate of node N-1 recomputes and rewrites the same values
      const int n0 = g + G * i, n = n0 < T.N ? n0 : T.N - 1;
      px[i] = DP[2 * n];
      py[i] = DP[2 * n + 1];
      fl[i] = NF[n];
      yv[i] = Y[n];
      tg[i] = TG[n];
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      dxv[i] = XS[px[i] < 0 ? zslot : px[i]];
      dyv[i] = XS[py[i] < 0 ? zslot : py[i]];
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int n0 = g + G * i, n = n0 < T.N ? n0 : T.N - 1;
      const bool real = n0 < T.N, top = fl[i] & TF_TOP;
      const double dt = (double)fabsf(tg[i] - yv[i]);                 // all_dt (ENV:514)
      const float dr = fabsf((float)(dyv[i] / max_def));             // all_d (ENV:516)
      p_dt += (real && top) ? dt : 0.0;
      p_c2 = top ? p_c2 : fmaxf(p_c2, dr);                            // (a duplicate repeats node N-1's value: max is idempotent)
      DS[2 * n + 0] = (float)dxv[i];
      DS[2 * n + 1] = (float)dyv[i];
    }
    if (A.disp64 && active) {      // wave-uniform pointer test
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int n = g + G * i;
        if (n < T.N) {
          A.disp64[(bn + n) * 2 + 0] = dxv[i];
          A.disp64[(bn + n) * 2 + 1] = dyv[i];
        }
      }
    }
    for (int n = g + G * NPL; n < T.N; n += G) {     // topologies with more nodes per lane than the unrolled part covers
      const int qx = DP[2 * n], qy = DP[2 * n + 1];
      const double dx = XS[qx < 0 ? zslot : qx], dy = XS[qy < 0 ? zslot : qy];
      const float y = Y[n];
      if (NF[n] & TF_TOP) p_dt += (double)fabsf(TG[n] - y);
      else p_c2 = fmaxf(p_c2, fabsf((float)(dy / max_def)));
      DS[2 * n + 0] = (float)dx;
      DS[2 * n + 1] = (float)dy;
      if (A.disp64 && active) {
        A.disp64[(bn + n) * 2 + 0] = dx;
        A.disp64[(bn + n) * 2 + 1] = dy;
      }
    }
    if (A.mu_out && T.has_pairs) {
      constexpr int PPL = (NPL + 1) / 2;
#pragma unroll
      for (int i = 0; i < PPL; ++i) {
        const int p0 = g + G * i, p = p0 < T.NP ? p0 : T.NP - 1;
        const int lo = PR[2 * p], hi = PR[2 * p + 1];
        const float ylo = Y[lo], yhi = Y[hi];
        float u0, d0, u1, d1;
        move_range(NF[lo] & TF_TOP, ylo, yhi, u0, d0);
        move_range(NF[hi] & TF_TOP, yhi, ylo, u1, d1);
        MU[lo] = u0;
        MD[lo] = d0;
        MU[hi] = u1;
        MD[hi] = d1;
      }
      for (int p = g + G * PPL; p < T.NP; p += G) {
        int lo = PR[2 * p], hi = PR[2 * p + 1];
        move_range(NF[lo] & TF_TOP, Y[lo], Y[hi], MU[lo], MD[lo]);
        move_range(NF[hi] & TF_TOP, Y[hi], Y[lo], MU[hi], MD[hi]);
      }
    }
    if (A.energy)
      for (int r = g; r < T.ndof; r += G) p_en += XS[r] * load_at(T, r);
  }

  // copy one result row LDS -> HBM with the env's own G lanes (generic fallback)
  TRUSS_HD void store_row(void *dst, const void *src, int nbytes) const {
    if ((nbytes & 15) == 0 && (((size_t)dst) & 15) == 0) {
      const tb_u4 *s4 = (const tb_u4 *)src;
      tb_u4 *d4 = (tb_u4 *)dst;
      for (int q = g; q < (nbytes >> 4); q += G) d4[q] = s4[q];
    } else if ((nbytes & 3) == 0) {
      const uint32_t *s1 = (const uint32_t *)src;
      uint32_t *d1 = (uint32_t *)dst;
      for (int q = g; q < (nbytes >> 2); q += G) d1[q] = s1[q];
    } else {
      const uint8_t *s1 = (const uint8_t *)src;
      uint8_t *d1 = (uint8_t *)dst;
      for (int q = g; q < nbytes; q += G) d1[q] = s1[q];
    }
  }
  template <int IT>
  TRUSS_HD void out_load(const void *src, int nq, tb_u4 (&v)[IT]) const {
    const tb_u4 *s4 = (const tb_u4 *)src;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int q = g + G * i;
      v[i] = s4[q < nq ? q : nq - 1];
    }
  }
  template <int IT>
  TRUSS_HD void out_store(void *dst, int nq, const tb_u4 (&v)[IT]) const {
    tb_u4 *d4 = (tb_u4 *)dst;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int q = g + G * i;
      TB_STREAM_STORE(&d4[q < nq ? q : nq - 1], v[i]);  // clamped duplicates rewrite the same 16 bytes
    }
  }

  // ---- phase 8a: stream the result rows out: every LDS read first, then every global store ----
  TRUSS_HD void phase_store(const TopoDev &T, const StepArgsDev &A) {
    if (!active) return;
    const size_t bn = (size_t)env * T.N, be = (size_t)env * T.E;
    const bool mr = A.mu_out && T.has_pairs;
    const bool fast = (T.N & 3) == 0 && (T.E & 15) == 0 && T.N <= NCAP && T.E <= ECAP;
    if (fast) {
      const int nn = T.N >> 2, ne = T.E >> 2, nc = T.E >> 4;
      tb_u4 vy[NIT], vs[EIT], vq[EIT], vr[EIT], vc[(EIT + 3) / 4], vd[2 * NIT], vu[NIT], vw[NIT];
      out_load<NIT>(ysh(T), nn, vy);
      out_load<EIT>(secsh(T), ne, vs);
      out_load<EIT>(oq0(T), ne, vq);
      out_load<EIT>(osr(T), ne, vr);
      out_load<(EIT + 3) / 4>(ocomp(T), nc, vc);
      out_load<2 * NIT>(odisp(T), 2 * nn, vd);
      if (mr) {
        out_load<NIT>(omu(T), nn, vu);
        out_load<NIT>(omd(T), nn, vw);
      }
      out_store<NIT>(rs_y_out + bn, nn, vy);
      if (rs_sec_out) out_store<EIT>(rs_sec_out + be, ne, vs);
      out_store<EIT>(A.q0 + be, ne, vq);
      out_store<EIT>(A.sr + be, ne, vr);
      out_store<(EIT + 3) / 4>(A.comp + be, nc, vc);
      out_store<2 * NIT>(A.disp + bn * 2, 2 * nn, vd);
      if (mr) {
        out_store<NIT>(A.mu_out + bn, nn, vu);
        out_store<NIT>(A.md_out + bn, nn, vw);
      }
      return;
    }
    store_row(rs_y_out + bn, ysh(T), 4 * T.N);
    if (rs_sec_out) store_row(rs_sec_out + be, secsh(T), 4 * T.E);
    store_row(A.q0 + be, oq0(T), 4 * T.E);
    store_row(A.sr + be, osr(T), 4 * T.E);
    store_row(A.comp + be, ocomp(T), T.E);
    store_row(A.disp + bn * 2, odisp(T), 8 * T.N);
    if (mr) {
      store_row(A.mu_out + bn, omu(T), 4 * T.N);
      store_row(A.md_out + bn, omd(T), 4 * T.N);
    }
  }

  // ---- phase 8: the env's objective partials are folded across its G lanes in registers (DPP butterfly: tb_group_sum_d /
  // tb_group_max_f; the float32 contributions are summed in float64, which is exact, so the order does not matter) and a few
  // lanes write the results:  lane 0: point (one 16-byte store), obj   lane 1: energy, reactions   lane 2: status
  // (Through LDS -- every lane's partials stored, six lanes folding 16 values each -- this phase took 2.2 k of 43 k cycles.)
  TRUSS_HD void phase_finish(const TopoDev &T, const StepArgsDev &A) {
    const double s_vol = tb_group_sum_d(*this, 0), s_dt = tb_group_sum_d(*this, 1);
    const float m_c1 = tb_group_max_f(*this, 0), m_c2 = tb_group_max_f(*this, 1), m_bad = tb_group_max_f(*this, 2);
    double s_en = 0.0;
    if (A.energy) s_en = tb_group_sum_d(*this, 2);     // wave-uniform
    if (!active) return;
    if (g == 0) {
      const float obj1 = (float)s_vol, obj2 = (float)s_dt;
      const tb_f4 pt = {obj1 / int1, obj2 / int2, m_c1, m_c2};
      *(tb_f4 *)(A.point + (size_t)env * 4) = pt;
      if (A.obj) {
        const tb_f2 o = {obj1, obj2};
        *(tb_f2 *)(A.obj + (size_t)env * 2) = o;
      }
    }
    if (g == 1 % G) {
      if (A.energy) A.energy[env] = 0.5 * s_en;
      if (A.react) {
        const double *RB = rbuf(T);
        for (int i = 0; i < T.n_rest; ++i) A.react[(size_t)env * T.n_rest + i] = RB[i];
      }
    }
    // bit 0: non-positive pivot; bit 1: the streaming wave of an EMIT workgroup gave up waiting (tb_obs_timed_out: the workgroup's
    // timeout word; the streaming wave also raises the bit itself, in case this wave is the one that never gets here)
    if (g == 2 % G && A.status) A.status[env] = (m_bad != 0.0f ? 1 : 0) | (tb_obs_timed_out(*this, T) ? 2 : 0);
  }

  // ==== observation emission (EMIT) ==============================================================
  // state_data / state_data_not_norm (truss2D_ENV.py:40-193) of the design this step analyses, written by the
  // step's own wave: every input of the observation is in LDS or registers at some point of the step, so the
  // second launch, its re-read of the result rows and the host hop between the two launches go away.
  //
  // All index arithmetic is done once per topology on the host (truss_host.h, tb_build_emit): an output tensor
  // is a flat run of 16-byte chunks per env, chunk q is written by lane q % G, and a table gives, for each of
  // the chunk's four floats, the offset of its source inside the env's LDS region -- either a row the step
  // holds anyway (x, y, staged q0 / max_up / max_down) or an entry of the FEATURE BANK (derived per-node /
  // per-element values, placed in bytes that are dead by the time they are written).  The kernel side is then:
  // fill the bank, gather, stream.  Every store instruction writes whole aligned 256-byte runs per env (a first
  // version stored the all-zero chunks of the N x N matrices separately from the chunks holding an edge: partial
  // writes of the same lines at different times, 71.7 us per step instead of 54.7 us for two launches).
  //
  // WHO streams.  The launch is one compute wave per SIMD; a store burst issued by that wave blocks it (in-order
  // issue, a short store queue) for as long as HBM takes to drain the burst, so stores issued early do not overlap
  // with its later compute: step time = compute + bytes / bandwidth (35.0 us with A_s streamed before the solver
  // and A_n_ts / A_n_cs before the nodal phases; 40.9 us with everything at the end).  EMIT workgroups therefore
  // have a SECOND wavefront, the streaming wave: it loads the tables, sleeps on a progress word in LDS and does
  // all the gather + store work of the tensors whose inputs the compute wave has announced:
  //   1  new sections final (after sizing)        -> A_s   (area ratios: needs the sections only; ENV:86-87)
  //   2  element part of the bank written         -> A_n_ts, A_n_cs
  //   3  node part of the bank written (the end)  -> nN_x_e, nN_x_n, x_n
  // (publishing the raw node features separately, so that nN_x_e / nN_x_n start before the normalisation: +0.8 us)
  // The compute wave never waits for the streaming wave and never stores an observation byte; of the observation work
  // it keeps the bank: derived element / node values it has in registers or one LDS read away (the streaming wave
  // shares its SIMD with another workgroup's compute wave and gets the issue slots that one leaves: arithmetic is
  // slow there, waiting for HBM is free), with the column normalisation of x_n as its very last act.  Every table
  // lookup, gather and store is the streaming wave's.  The two communicate through the
  // result rows the step stages in LDS anyway (q0, stress ratios, flags, displacements, move ranges), the design rows
  // and the bank; everything the streaming wave reads stays untouched until the workgroup ends, and the bank lives
  // in bytes nobody else touches any more when it is written (dead band, action rows, solver scratch except the
  // reactions).  Measured at 4096 envs: bank by the compute wave 32.4-33.4 us, bank by the streaming wave 35.8 us.  Its table loads precede all its stores: gfx9 has ONE vmcnt for
  // loads and stores, in issue order, and a load issued behind a store can only be waited for together with it.
  static constexpr int NF = G * NPL, EF = G * EPL;   // nodes / elements covered by the unrolled per-lane loops
  static constexpr int IX = EMIT ? (13 * NF / 4 + G - 1) / G : 1;     // chunks per lane: x_n
  static constexpr int IN_ = EMIT ? (12 * NF / 4 + G - 1) / G : 1;    //                  nN_x_n
  static constexpr int IE = EMIT ? (21 * EF / 4 + G - 1) / G : 1;     //                  nN_x_e
  static constexpr int IM = EMIT ? (NF * NF / 4 + G - 1) / G : 1;     //                  each N x N matrix
  static constexpr int NDYN = 9;                                      // x_n columns 0 1 4 7 8 9 10 11 12 vary per env
  static constexpr int KB = 8;                                        // chunks gathered back to back before their stores
  tb_u2 etx[IX], etn[IN_], etm[IM], ete[IE];
  float el[EMIT ? EPL : 1];           // element lengths (phase_elements)
  float nfe[EMIT ? NPL : 1][NDYN];    // raw dynamic columns of the lane's nodes
  float pmn[NDYN], pmx[NDYN];         // partial column min / max over the lane's nodes

  template <int IT>
  TRUSS_HD void tab_load(const char *base, tb_u2 (&v)[IT]) const {
    const tb_u4 *t = (const tb_u4 *)base;
#pragma unroll
    for (int p = 0; p < (IT + 1) / 2; ++p) {
      const tb_u4 w = t[p * G + g];
      const tb_u2 lo = {w[0], w[1]}, hi = {w[2], w[3]};
      v[2 * p] = lo;
      if (2 * p + 1 < IT) v[2 * p + 1] = hi;
    }
  }
  TRUSS_HD void emit_tables_load(const TopoDev &T) {   // streaming wave, before its first store
    if constexpr (EMIT) {
      tab_load<IM>(T.etab + T.et_mat, etm);
      tab_load<IX>(T.etab + T.et_xn, etx);
      tab_load<IN_>(T.etab + T.et_nxn, etn);
    }
  }
  // store address of iteration i (a lane past the end of the tensor redoes its last chunk)
  TRUSS_HD int chunk_of(int i, int nc) const {
    const int q = g + G * i;
    return q < nc ? q : nc - 1;
  }

  // ---- A_s (ENV:86-87): area / largest area on both cells of every element; needs the new sections only:
  // cell -> element (table) -> section (the env's section row) -> ratio (topology table)
  TRUSS_HD void obs_emit_as(const TopoDev &T, const StepArgsDev &A) {
    if constexpr (EMIT) {
      if (!active || !A.A_s) return;
      const int32_t *S = secsh(T);
      const float *VF = TB_TAB(float, TB, T.f_vsf);
      tb_f4 *o4 = (tb_f4 *)A.A_s + (size_t)env * ((size_t)T.N * T.N / 4);
      const int E = T.E;
#pragma unroll
      for (int b0 = 0; b0 < IM; b0 += KB) {
        if (b0 * G < T.nc_mat) {             // wave-uniform, once per batch
          int sc[KB][4];
#pragma unroll
          for (int k = 0; k < KB; ++k)
            if (b0 + k < IM) {
              const tb_u2 t = etm[b0 + k];
              const int e0 = t[0] & 0xffffu, e1 = t[0] >> 16, e2 = t[1] & 0xffffu, e3 = t[1] >> 16;
              sc[k][0] = e0 < E ? S[e0] : -1;
              sc[k][1] = e1 < E ? S[e1] : -1;
              sc[k][2] = e2 < E ? S[e2] : -1;
              sc[k][3] = e3 < E ? S[e3] : -1;
            }
#pragma unroll
          for (int k = 0; k < KB; ++k)
            if (b0 + k < IM) {
              const tb_f4 v = {sc[k][0] >= 0 ? VF[sc[k][0]] : 0.0f, sc[k][1] >= 0 ? VF[sc[k][1]] : 0.0f,
                               sc[k][2] >= 0 ? VF[sc[k][2]] : 0.0f, sc[k][3] >= 0 ? VF[sc[k][3]] : 0.0f};
              TB_OBS_STORE(&o4[chunk_of(b0 + k, T.nc_mat)], v);
            }
        }
      }
    }
  }
  // ---- A_n_ts / A_n_cs (ENV:89-100): one (tension, compression) record per cell, two chunks stored
  TRUSS_HD void obs_emit_tc(const TopoDev &T, const StepArgsDev &A) {
    if constexpr (EMIT) {
      if (!active || (!A.A_ts && !A.A_cs)) return;
      const tb_f2 *R = (const tb_f2 *)((const float *)L + T.b_tc);
      const size_t nn4 = (size_t)T.N * T.N / 4;
      tb_f4 *pt = (tb_f4 *)A.A_ts + env * nn4, *pc = (tb_f4 *)A.A_cs + env * nn4;
#pragma unroll
      for (int b0 = 0; b0 < IM; b0 += KB) {
        if (b0 * G < T.nc_mat) {
          tb_f2 r[KB][4];
#pragma unroll
          for (int k = 0; k < KB; ++k)
            if (b0 + k < IM) {
              const tb_u2 t = etm[b0 + k];
              r[k][0] = R[t[0] & 0xffffu];
              r[k][1] = R[t[0] >> 16];
              r[k][2] = R[t[1] & 0xffffu];
              r[k][3] = R[t[1] >> 16];
            }
#pragma unroll
          for (int k = 0; k < KB; ++k)
            if (b0 + k < IM) {
              const int q = chunk_of(b0 + k, T.nc_mat);
              const tb_f4 vt = {r[k][0][0], r[k][1][0], r[k][2][0], r[k][3][0]};
              const tb_f4 vc = {r[k][0][1], r[k][1][1], r[k][2][1], r[k][3][1]};
              if (A.A_ts) TB_OBS_STORE(&pt[q], vt);
              if (A.A_cs) TB_OBS_STORE(&pc[q], vc);
            }
        }
      }
    }
  }

  // ---- bank, elements (compute wave, behind progress 2): nN_x_e columns 0-4, 6 (ENV:148-156) from the registers of post_elements
  float pe_sr[EMIT ? EPL : 1];
  uint32_t pe_cmp;               // bit i: element i of the lane is in compression
  TRUSS_HD void obs_elements_bank(const TopoDev &T, const StepArgsDev &A) {
    if constexpr (EMIT) {
      float *Lf = (float *)L;
      const int32_t *S = secsh(T);
      const float *AF = TB_TAB(float, TB, T.f_areaf);
      int sc[EPL];
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        const int e = g + G * i;
        sc[i] = S[e < T.E ? e : T.E - 1];      // a clamped duplicate rewrites element E-1's values
      }
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        const int e = g + G * i;
        const int ee = e < T.E ? e : T.E - 1;
        const bool cmp = (pe_cmp >> i) & 1u;
        const tb_f4 r0 = {(float)sc[i], AF[sc[i]], el[i], cmp ? 0.0f : 1.0f};
        const tb_f2 r1 = {cmp ? 1.0f : 0.0f, pe_sr[i] > 1.0f ? 1.0f : 0.0f};
        ((tb_f4 *)(Lf + T.b_erec))[ee] = r0;
        ((tb_f2 *)(Lf + T.b_erec2))[ee] = r1;
      }
      if (g == 0) {
        Lf[T.b_const + 0] = 0.0f;
        Lf[T.b_const + 1] = 1.0f;
        Lf[T.b_const + 2] = (1.0f - 0.0f) / (1.0f - 0.0f + 1e-6f);   // a 0/1 column with both values present
      }
    }
  }

  // ---- bank, nodes A (compute wave, behind the step's own stores): raw features of the lane's nodes (ENV:50-100, 134-146) ----
  TRUSS_HD void obs_nodes_raw(const TopoDev &T, const StepArgsDev &A) {
    if constexpr (EMIT) {
      load_params(T);
      float *Lf = (float *)L;
      const float *Y = ysh(T), *X = xsh(T), *TG = tgsh(T), *DS = odisp(T), *MU = omu(T), *MD = omd(T);
      const uint8_t *NFL = t_nflags(T);
      const float maxdef32 = (float)max_def;
      const int lbit = is_roof ? TF_LOAD_ROOF : TF_LOAD_BRIDGE;
#pragma unroll
      for (int c = 0; c < NDYN; ++c) {
        pmn[c] = INFINITY;
        pmx[c] = -INFINITY;
      }
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int n = g + G * i;
        const int nc = n < T.N ? n : T.N - 1;   // a clamped duplicate rewrites node N-1's values
        const int fl = NFL[nc];
        const float y = Y[nc];
        float *f = nfe[i];
        f[0] = X[nc];
        f[1] = y;
        f[2] = (fl & lbit) ? 1.0f : 0.0f;
        f[3] = MU[nc];
        f[4] = MD[nc];
        f[5] = (fl & TF_TOP) ? TG[nc] / (y + 1e-6f) : 0.0f;
        f[6] = fabsf(DS[2 * nc + 1]);
        const float ratio = f[6] / maxdef32;
        f[7] = fminf(ratio, 1.0f) * (ratio > 1.0f ? 1.0f : 0.5f);
        f[8] = ratio > 1.0f ? 1.0f : 0.0f;
        const tb_f4 raw = {f[2], f[5], f[6], ratio >= 1.0f ? 1.0f : 0.0f};
        ((tb_f4 *)(Lf + T.b_nraw))[nc] = raw;
        if (n < T.N) {
#pragma unroll
          for (int c = 0; c < NDYN; ++c) {
            pmn[c] = fminf(pmn[c], f[c]);
            pmx[c] = fmaxf(pmx[c], f[c]);
          }
        }
      }
    }
  }

  // ---- bank, nodes B (compute wave, last): column min / max over the env's lanes, normalised columns (ENV:102) ----
  TRUSS_HD void obs_bank_fill(const TopoDev &T, const StepArgsDev &A) {
    if constexpr (EMIT) {
      float *Lf = (float *)L;
      float nv[NPL][NDYN];
#pragma unroll
      for (int c = 0; c < NDYN; ++c) {
        const float lo = tb_group_min(*this, c), hi = tb_group_max(*this, c);
        // (v - lo) / (hi - lo + 1e-6f) (ENV:102) as a product with the reciprocal of the column's denominator (one
        // reciprocal per column instead of one IEEE division per value): <= 1 ulp from the quotient
        const float inv = tb_rcpf(hi - lo + 1e-6f);
#pragma unroll
        for (int i = 0; i < NPL; ++i) nv[i][c] = (nfe[i][c] - lo) * inv;
      }
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int n = g + G * i;
        const int nc = n < T.N ? n : T.N - 1;
        const tb_f4 a = {nv[i][0], nv[i][1], nv[i][2], nv[i][3]}, b = {nv[i][4], nv[i][5], nv[i][6], nv[i][7]};
        ((tb_f4 *)(Lf + T.b_nna))[nc] = a;
        ((tb_f4 *)(Lf + T.b_nnb))[nc] = b;
        Lf[T.b_nn8 + nc] = nv[i][8];
      }
    }
  }

  TRUSS_HD tb_f4 gather4(const float *Lf, const tb_u2 t) const {
    const tb_f4 v = {Lf[t[0] & 0xffffu], Lf[t[0] >> 16], Lf[t[1] & 0xffffu], Lf[t[1] >> 16]};
    return v;
  }
  // rows of one tensor: KB chunks' gathers are issued back to back, then their stores (a chunk at a time the
  // wave paid one LDS round trip per 16 output bytes)
  template <int IT>
  TRUSS_HD void emit_rows(float *out, size_t row_floats, int nc, const tb_u2 (&tab)[IT]) const {
    if (!out) return;
    const float *Lf = (const float *)L;
    tb_f4 *o4 = (tb_f4 *)(out + (size_t)env * row_floats);
#pragma unroll
    for (int b0 = 0; b0 < IT; b0 += KB) {
      if (b0 * G < nc) {
        tb_f4 v[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k)
          if (b0 + k < IT) v[k] = gather4(Lf, tab[b0 + k]);
#pragma unroll
        for (int k = 0; k < KB; ++k)
          if (b0 + k < IT) TB_OBS_STORE(&o4[chunk_of(b0 + k, nc)], v[k]);
      }
    }
  }
  // the nN_x_e table sits in LDS (staged with the topology tables); the streaming wave copies its entries to registers
  // while it waits for the node bank: one LDS round trip less per batch of the last, exposed segment
  TRUSS_HD void obs_nxe_table(const TopoDev &T) {
    if constexpr (EMIT) {
      if (T.nxe_cw != 4) return;
      const tb_u2 *tab = TB_TAB(tb_u2, TB, T.f_tnxe);
#pragma unroll
      for (int i = 0; i < IE; ++i) ete[i] = tab[i * G + g];
    }
  }
  // nN_x_e of a topology whose element count is not a multiple of 4: 8- or 4-byte chunks, four per lane and round
  TRUSS_HD void emit_nxe_narrow(const TopoDev &T, const StepArgsDev &A) const {
    if (!A.nxe) return;
    const float *Lf = (const float *)L;
    const uint16_t *tab = TB_TAB(uint16_t, TB, T.f_tnxe);
    float *o = A.nxe + (size_t)env * 21 * T.E;
    const int nc = T.nc_nxe;
    if (T.nxe_cw == 2) {
      for (int q0 = g; q0 < nc; q0 += 4 * G) {
        tb_f2 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = q0 + k * G < nc ? q0 + k * G : nc - 1;
          const tb_f2 w = {Lf[tab[2 * q]], Lf[tab[2 * q + 1]]};
          v[k] = w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = q0 + k * G < nc ? q0 + k * G : nc - 1;
          TB_OBS_STORE((tb_f2 *)o + q, v[k]);
        }
      }
    } else {
      for (int q0 = g; q0 < nc; q0 += 4 * G) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = Lf[tab[q0 + k * G < nc ? q0 + k * G : nc - 1]];
#pragma unroll
        for (int k = 0; k < 4; ++k) TB_OBS_STORE(o + (q0 + k * G < nc ? q0 + k * G : nc - 1), v[k]);
      }
    }
  }
  TRUSS_HD void obs_emit_raw_rows(const TopoDev &T, const StepArgsDev &A) {
    if constexpr (EMIT) {
      if (!active) return;
      if (T.nxe_cw == 4) emit_rows<IE>(A.nxe, (size_t)21 * T.E, T.nc_nxe, ete);
      else emit_nxe_narrow(T, A);
      emit_rows<IN_>(A.nxn, (size_t)12 * T.N, T.nc_nxn, etn);
    }
  }
  TRUSS_HD void obs_emit_xn(const TopoDev &T, const StepArgsDev &A) {
    if constexpr (EMIT) {
      if (!active) return;
      emit_rows<IX>(A.x_n, (size_t)13 * T.N, T.nc_xn, etx);
    }
  }
};

// The phase schedule, shared by the HIP kernel and the emulator.
//   EMIT_POINT(k)  (EMIT kernels) everything segment k of the streaming wave reads is final.  HIP: the compute wave
//               publishes progress k and goes on; emulator: runs TRUSS_STREAM_SEGk here
//   PH(call)    run `ln.call` for every lane, then a workgroup barrier
//   PH_NS(call) run it without a trailing barrier (no lane reads what another lane writes in it
//               before the next barrier)
//   BAR()       explicit barrier
// W_ and EMIT_ must be constexprs in scope; TRUSS_UNROLL expands to the unroll pragma on the GPU so that the
// register-window indices (kk_) are compile-time constants.
#ifndef TRUSS_ST
#define TRUSS_ST(i)  // phase time stamp hook (diagnostic build only)
#endif
// The streaming wave's work per progress value (PH as above).  HIP: the second wavefront of an EMIT workgroup runs
// segment k once the compute wave has published k; emulator: EMIT_POINT(k, ...) runs segment k in place.
#define TRUSS_STREAM_SEG1(PH, T, A) PH(obs_emit_as(T, A));
#define TRUSS_STREAM_SEG2(PH, T, A) \
  PH(obs_emit_tc(T, A));            \
  PH(obs_nxe_table(T));
#define TRUSS_STREAM_SEG3(PH, T, A) \
  PH(obs_emit_raw_rows(T, A));      \
  PH(obs_emit_xn(T, A));
#define TRUSS_STEP_SCHEDULE(PH, PH_NS, BAR, T, A)                                   \
  TRUSS_ST(0);                                                                      \
  PH(phase_stage(T, A));                                                            \
  TRUSS_ST(1);                                                                      \
  PH(phase_decode(T, A));                                                           \
  if ((T).n_sym_nodes > 0 && !((A).flags & TB_NO_DECODE)) { PH(phase_sym_nodes(T)); } \
  TRUSS_ST(2);                                                                      \
  PH(phase_sizing(T, A));                                                           \
  if ((T).n_sym_elems > 0 && !((A).flags & TB_NO_DECODE)) { PH(phase_sym_elems(T)); } \
  if (EMIT_) { EMIT_POINT(1); TRUSS_ST(19); } /* compile-time; the sections are final */ \
  PH_NS(rollout_prefetch(T, A)); /* persistent rollout only: the action rows are consumed, fetch the next step's */ \
  TRUSS_ST(3);                                                                      \
  PH(phase_elements(T, A));                                                         \
  TRUSS_ST(10);                                                                     \
  PH(phase_assemble_nodes(T));                                                      \
  TRUSS_ST(11);                                                                     \
  PH(solver_scratch_init(T));                                                       \
  TRUSS_ST(4);                                                                      \
  PH(solver_init(T));                                                               \
  {                                                                                 \
    /* Factorisation.  Blocks of W pivots that lie entirely inside a team's own part run the clean   \
       block loop. */                                                                                \
    const int kend_ = (T).nteams == 2 ? (T).KA + W_ : (T).KA;                       \
    int kb_ = 0;                                                                    \
    for (; kb_ + W_ <= (T).KA; kb_ += W_) {                                         \
      PH_NS(factor_rows_fetch(T, kb_));                                             \
      TRUSS_UNROLL                                                                  \
      for (int kk_ = 0; kk_ < W_; ++kk_) {                                          \
        PH(pivot_write(T, kb_ + kk_, kk_));                                         \
        PH_NS(pivot_update(T, kb_ + kk_, kk_));                                     \
      }                                                                             \
      PH_NS(factor_z_store(T, kb_, 0, W_));                                         \
    }                                                                               \
    TRUSS_ST(13);                                                                   \
    /* Two-sided scheme: the rest of team A's/B's own pivots of this block, the merge at k = KA, and the \
       W pivots of the middle block.  Where the merge falls inside a block of W (c = KA mod W) is a      \
       property of the topology; the region is instantiated for every c and one instance runs, so that   \
       all window indices are compile-time and no pivot carries a guard. */                              \
    if ((T).nteams == 2 && W_ <= 8) {                                               \
      const int c0_ = (T).KA % W_;                                                  \
      TRUSS_UNROLL                                                                  \
      for (int c_ = 0; c_ < W_; ++c_) {                                             \
        if (c_ == c0_) {                                                            \
          if (c_ > 0) { PH_NS(factor_rows_fetch(T, kb_)); }                         \
          TRUSS_UNROLL                                                              \
          for (int kk_ = 0; kk_ < c_; ++kk_) {                                      \
            PH(pivot_write(T, kb_ + kk_, kk_));                                     \
            PH_NS(pivot_update(T, kb_ + kk_, kk_));                                 \
          }                                                                         \
          if (c_ > 0) { PH_NS(factor_z_store(T, kb_, 0, c_)); }                     \
          PH(merge_post(T));                                                        \
          PH(merge_take(T));                                                        \
          TRUSS_UNROLL                                                              \
          for (int i_ = 0; i_ < W_; ++i_) {                                         \
            if ((c_ + i_) % W_ == 0) { PH_NS(factor_rows_fetch(T, (T).KA + i_)); }  \
            PH(pivot_write(T, (T).KA + i_, (c_ + i_) % W_));                        \
            PH_NS(pivot_update(T, (T).KA + i_, (c_ + i_) % W_));                    \
          }                                                                         \
          PH_NS(factor_z_store(T, (T).KA, c_, W_));                                 \
        }                                                                           \
      }                                                                             \
    } else if ((T).nteams == 2) { /* wide windows: W instances would not fit; guarded blocks instead */ \
      for (; kb_ < kend_; kb_ += W_) {                                              \
        PH_NS(factor_rows_fetch(T, kb_));                                           \
        TRUSS_UNROLL                                                                \
        for (int kk_ = 0; kk_ < W_; ++kk_) {                                        \
          if (kb_ + kk_ < kend_) {                                                  \
            if (kb_ + kk_ == (T).KA) {                                              \
              PH(merge_post(T));                                                    \
              PH(merge_take(T));                                                    \
            }                                                                       \
            PH(pivot_write(T, kb_ + kk_, kk_));                                     \
            PH_NS(pivot_update(T, kb_ + kk_, kk_));                                 \
          }                                                                         \
        }                                                                           \
        PH_NS(factor_z_store(T, kb_, 0, kend_ - kb_ < W_ ? kend_ - kb_ : W_));      \
      }                                                                             \
    }                                                                               \
    BAR();                                                                          \
    PH_NS(pivot_check(T));                                                          \
    TRUSS_ST(5);                                                                    \
    /* Back substitution, top row first.  Two-sided scheme: the W steps of the middle block, the hand-over \
       of the middle solutions to team B at k = KA - 1 and the rest of that block, instantiated per        \
       c = KA mod W like the merge region above; everything below runs the clean block loop. */            \
    kb_ = ((kend_ - 1) / W_) * W_;                                                  \
    PH_NS(backsub_rows_fetch(T, kb_));                                              \
    if ((T).nteams == 2 && W_ <= 8) {                                               \
      const int c0_ = (T).KA % W_;                                                  \
      TRUSS_UNROLL                                                                  \
      for (int c_ = 0; c_ < W_; ++c_) {                                             \
        if (c_ == c0_) {                                                            \
          PH_NS(backsub_rows_adopt());                                              \
          PH_NS(backsub_rows_fetch(T, kb_ - W_));                                   \
          TRUSS_UNROLL                                                              \
          for (int i_ = W_ - 1; i_ >= 0; --i_) {                                    \
            PH_NS(backsub_step(T, (T).KA + i_, (c_ + i_) % W_));                    \
            PH_NS(backsub_share((c_ + i_) % W_));                                   \
            if ((c_ + i_) % W_ == 0 && i_ > 0) { /* into the block below */        \
              kb_ -= W_;                                                            \
              PH_NS(backsub_rows_adopt());                                          \
              PH_NS(backsub_rows_fetch(T, kb_ - W_));                               \
            }                                                                       \
          }                                                                         \
          PH_NS(backsub_flush_mid(T));                                              \
          BAR();                                                                    \
          PH_NS(backsub_reload(T));                                                 \
          TRUSS_UNROLL                                                              \
          for (int kk_ = c_ - 1; kk_ >= 0; --kk_) {                                 \
            PH_NS(backsub_step(T, kb_ + kk_, kk_));                                 \
            PH_NS(backsub_share(kk_));                                              \
          }                                                                         \
          if (c_ > 0) { PH_NS(backsub_flush(T, kb_)); }                             \
          kb_ -= W_;                                                                \
        }                                                                           \
      }                                                                             \
    } else if ((T).nteams == 2) {                                                   \
      for (; kb_ >= 0 && kb_ + W_ - 1 >= (T).KA - 1; kb_ -= W_) {                   \
        PH_NS(backsub_rows_adopt());                                                \
        PH_NS(backsub_rows_fetch(T, kb_ - W_));                                     \
        TRUSS_UNROLL                                                                \
        for (int kk_ = W_ - 1; kk_ >= 0; --kk_) {                                   \
          if (kb_ + kk_ < kend_) {                                                  \
            if (kb_ + kk_ == (T).KA - 1) {                                          \
              PH_NS(backsub_flush_mid(T));                                          \
              BAR();                                                                \
              PH_NS(backsub_reload(T));                                             \
            }                                                                       \
            PH_NS(backsub_step(T, kb_ + kk_, kk_));                                 \
            PH_NS(backsub_share(kk_));                                              \
          }                                                                         \
        }                                                                           \
        PH_NS(backsub_flush(T, kb_));                                               \
      }                                                                             \
    }                                                                               \
    TRUSS_ST(14);                                                                   \
    for (; kb_ >= 0; kb_ -= W_) {                                                   \
      PH_NS(backsub_rows_adopt());                                                  \
      PH_NS(backsub_rows_fetch(T, kb_ - W_));                                       \
      TRUSS_UNROLL                                                                  \
      for (int kk_ = W_ - 1; kk_ >= 0; --kk_) {                                     \
        PH_NS(backsub_step(T, kb_ + kk_, kk_));                                     \
        PH_NS(backsub_share(kk_));                                                  \
      }                                                                             \
      PH_NS(backsub_flush(T, kb_));                                                 \
    }                                                                               \
  }                                                                                 \
  BAR();                                                                            \
  TRUSS_ST(6);                                                                      \
  PH_NS(rollout_stash(T, A));                                                       \
  PH(phase_post_elements(T, A));                                                    \
  if (EMIT_) {                                                                      \
    EMIT_POINT(2);                                                                  \
    TRUSS_ST(18);                                                                   \
    PH(obs_elements_bank(T, A));                                                    \
  }                                                                                 \
  TRUSS_ST(7);                                                                      \
  PH(phase_post_nodes(T, A));                                                       \
  TRUSS_ST(8);                                                                      \
  if (EMIT_) { /* compile-time: the node part of the bank first, so that the streaming wave starts on the rows while   \
                  this wave issues the step's own result stores (they queue behind the observation stream anyway) */  \
    TRUSS_ST(15);                                                                   \
    PH(obs_nodes_raw(T, A));                                                        \
    TRUSS_ST(16);                                                                   \
    PH(obs_bank_fill(T, A));                                                        \
    EMIT_POINT(3);                                                                  \
    TRUSS_ST(17);                                                                   \
  }                                                                                 \
  PH_NS(phase_finish(T, A));                                                        \
  TRUSS_ST(12);                                                                     \
  PH_NS(phase_store(T, A));                                                         \
  TRUSS_ST(9);

// ================================================================================================
// Observation tensors: state_data + state_data_not_norm (truss2D_ENV.py:40-193).
// One env per 64-lane workgroup.  The three dynamic N x N matrices are built in LDS and streamed
// out as whole rows (float4), so HBM sees each output byte exactly once.
// ================================================================================================
struct ObsArgsDev {
  int32_t B;
  uint32_t flags;
  const float *x, *y;
  const int32_t *sec;
  const float *mu, *md, *target, *disp, *q0, *sr;
  const uint8_t *comp;
  const double *env_params;
  float *x_n, *A_s, *A_ts, *A_cs, *nxn, *nxe;
  int32_t tile_rows;   // rows of the N x N matrices built per pass (tb_obs_tile_rows)
  int32_t n_split;     // 1: one workgroup per env walks over the row tiles; > 1: one workgroup per (env, tile) -- large trusses, where a
                       // batch has few envs and each has hundreds of KB to write (tb_obs_split)
};

struct ObsLane {
  int lane, env;
  char *L;
  float maxdef32;
  int is_roof;
  // split launches (ObsArgsDev::n_split > 1): a workgroup either builds ONE tile of matrix rows (role 1) or the per-node /
  // per-element rows (role 2); role 0 = everything, tile after tile
  int role;
  int minmax_done;

  // LDS: raw[N][13] | mn[13] | mx[13] | part[128] | (pad to 16) | As[TR][N] | Ats[TR][N] | Acs[TR][N]
  TRUSS_HD float *raw() { return (float *)L; }
  TRUSS_HD float *mn(const TopoDev &T) { return raw() + T.N * 13; }
  TRUSS_HD float *mx(const TopoDev &T) { return mn(T) + 13; }
  TRUSS_HD float *part(const TopoDev &T) { return mx(T) + 13; }   // [2][4][16] partial column minima / maxima
  TRUSS_HD float *mats(const TopoDev &T) { return (float *)(L + (((size_t)(T.N * 13 + 26 + 128) * 4 + 15) & ~(size_t)15)); }

  TRUSS_HD void init(int lane_, int block, int tile, const TopoDev &, const ObsArgsDev &A, char *lds) {
    lane = lane_;
    env = block;
    L = lds;
    role = A.n_split <= 1 ? 0 : tile == 0 ? 2 : 1;     // the rows' workgroups (the longest) are dispatched first
    minmax_done = 0;
    const double *P = A.env_params + (size_t)env * 8;
    maxdef32 = (float)P[2];
    is_roof = P[7] != 0.0;
  }

  TRUSS_HD void phase_nodes(const TopoDev &T, const ObsArgsDev &A) {
    if (role == 1) return;
    const size_t bn = (size_t)env * T.N;
    float *R = raw();
    for (int n = lane; n < T.N; n += 64) {
      const int fl = TB_TAB(uint8_t, T.blob, T.f_nflags)[n];
      const float top = (fl & TF_TOP) ? 1.0f : 0.0f;
      const float y = A.y[bn + n];
      float f[13];
      f[0] = A.x[bn + n];
      f[1] = y;
      f[2] = (fl & TF_RESX) ? 1.0f : 0.0f;
      f[3] = (fl & TF_RESY) ? 1.0f : 0.0f;
      f[4] = (fl & (is_roof ? TF_LOAD_ROOF : TF_LOAD_BRIDGE)) ? 1.0f : 0.0f;
      f[5] = top;
      f[6] = 1.0f - top;
      f[7] = A.mu[bn + n];
      f[8] = A.md[bn + n];
      f[9] = (fl & TF_TOP) ? A.target[bn + n] / (y + 1e-6f) : 0.0f;
      f[10] = fabsf(A.disp[(bn + n) * 2 + 1]);
      const float ratio = f[10] / maxdef32;
      f[11] = fminf(ratio, 1.0f) * (ratio > 1.0f ? 1.0f : 0.5f);
      f[12] = ratio > 1.0f ? 1.0f : 0.0f;
#pragma unroll
      for (int c = 0; c < 13; ++c) R[n * 13 + c] = f[c];
      if (A.nxn) {
        float *o = A.nxn + (bn + n) * 12;
#pragma unroll
        for (int c = 0; c < 11; ++c) o[c] = f[c];
        o[11] = ratio >= 1.0f ? 1.0f : 0.0f;
      }
    }
  }

  // rows [r0, r0 + TR) of the three matrices: clear the tile
  TRUSS_HD void phase_tile_clear(const TopoDev &T, const ObsArgsDev &A) {
    if (role != 1 && !minmax_done) {
      // column min / max of the raw node features (ENV:102), first half: lane = (column, quarter of the nodes)
      const int c = lane & 15, q = lane >> 4;
      if (c < 13) {
        const float *R = raw();
        float lo = R[c], hi = R[c];                     // node 0: neutral for both
        for (int n = q; n < T.N; n += 4) {
          const float v = R[n * 13 + c];
          lo = fminf(lo, v);
          hi = fmaxf(hi, v);
        }
        part(T)[q * 16 + c] = lo;
        part(T)[64 + q * 16 + c] = hi;
      }
    }
    if (role == 2) return;
    float *M = mats(T);
    const int tot = 3 * A.tile_rows * T.N;
    if ((tot & 3) == 0) {                       // 16-byte stores (mats() is 16-byte aligned)
      tb_u4 *M4 = (tb_u4 *)M;
      const tb_u4 z = {0u, 0u, 0u, 0u};
      for (int i = lane; i < tot / 4; i += 64) M4[i] = z;
    } else {
      for (int i = lane; i < tot; i += 64) M[i] = 0.0f;
    }
  }

  TRUSS_HD void phase_edges(const TopoDev &T, const ObsArgsDev &A, int r0) {
    const float *R = raw();
    const int TR = A.tile_rows;
    if (r0 == 0 && role != 1) {               // second half of the column min / max: the four partial results per column
      if (lane < 13) {
        const float *P = part(T);
        mn(T)[lane] = fminf(fminf(P[lane], P[16 + lane]), fminf(P[32 + lane], P[48 + lane]));
        mx(T)[lane] = fmaxf(fmaxf(P[64 + lane], P[80 + lane]), fmaxf(P[96 + lane], P[112 + lane]));
      }
      minmax_done = 1;
    }
    const size_t be = (size_t)env * T.E;
    float *As = mats(T), *Ats = As + TR * T.N, *Acs = Ats + TR * T.N;
    const double *AR = TB_TAB(double, T.blob, T.f_area);
    const int16_t *CN = TB_TAB(int16_t, T.blob, T.f_conn);
    const double amax = AR[T.n_sections - 1];
    if (role == 1) {
      // one tile of a split launch: its entries come from the rows' incident elements (padded adjacency, 8 slots per node), one
      // (row, slot) per lane -- not from a scan over all elements by every tile's workgroup
      const int16_t *ADJ = TB_TAB(int16_t, T.blob, T.f_adj8);
      for (int i = lane; i < TR * 8; i += 64) {
        const int n = r0 + (i >> 3);
        const int e = n < T.N ? ADJ[n * 8 + (i & 7)] : T.E;
        if (e >= T.E) continue;
        const int a = CN[2 * e], b = CN[2 * e + 1];
        const int other = a == n ? b : a;
        const double area = AR[A.sec[be + e]];
        const float srv = A.sr[be + e];
        const float vs = (float)(area / amax);
        const float val = fminf(srv, 1.0f) * (srv > 1.0f ? 1.0f : 0.5f);
        As[(n - r0) * T.N + other] = vs;
        (A.comp[be + e] == 0 ? Ats : Acs)[(n - r0) * T.N + other] = val;
      }
      return;
    }
    for (int e = lane; e < T.E; e += 64) {
      const int a = CN[2 * e], b = CN[2 * e + 1];
      const int s = A.sec[be + e];
      const double area = AR[s];
      const float srv = A.sr[be + e];
      const int cmp = A.comp[be + e];
      const float vs = (float)(area / amax);
      const float val = fminf(srv, 1.0f) * (srv > 1.0f ? 1.0f : 0.5f);
      float *Aq = cmp == 0 ? Ats : Acs;
      const bool tiles = role != 2;          // the rows' workgroup of a split launch holds no tile
      if (tiles && a >= r0 && a < r0 + TR) { // the element's two entries, each in the tile that holds its row
        As[(a - r0) * T.N + b] = vs;
        Aq[(a - r0) * T.N + b] = val;
      }
      if (tiles && b >= r0 && b < r0 + TR) {
        As[(b - r0) * T.N + a] = vs;
        Aq[(b - r0) * T.N + a] = val;
      }
      if (A.nxe && r0 == 0) {
        float *o = A.nxe + (be + e) * 21;
        const double dx = (double)R[b * 13 + 0] - (double)R[a * 13 + 0];
        const double dy = (double)R[b * 13 + 1] - (double)R[a * 13 + 1];
        o[0] = (float)s;
        o[1] = (float)area;
        o[2] = (float)sqrt(dx * dx + dy * dy);
        o[3] = cmp ? 0.0f : 1.0f;
        o[4] = cmp ? 1.0f : 0.0f;
        o[5] = A.q0[be + e];
        o[6] = srv > 1.0f ? 1.0f : 0.0f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float *r = R + (q ? b : a) * 13;
          float *oo = o + 7 + 7 * q;
          oo[0] = r[0];
          oo[1] = r[1];
          oo[2] = r[2];
          oo[3] = r[3];
          oo[4] = r[4];
          oo[5] = r[10];
          oo[6] = (r[10] / maxdef32) >= 1.0f ? 1.0f : 0.0f;
        }
      }
    }
  }

  TRUSS_HD void phase_store(const TopoDev &T, const ObsArgsDev &A, int r0) {
    const float *R = raw();
    if (A.x_n && r0 == 0 && role != 1) {
      float *o = A.x_n + (size_t)env * T.N * 13;
      const float *lo = mn(T), *hi = mx(T);
      for (int i = lane; i < T.N * 13; i += 64) {
        int c = i % 13;
        TB_STREAM_STORE(&o[i], (R[i] - lo[c]) / (hi[c] - lo[c] + 1e-6f));
      }
    }
    if (role == 2) return;
    const int rows = (r0 + A.tile_rows <= T.N ? A.tile_rows : T.N - r0);
    const int nn = rows * T.N;                       // floats of this tile
    const float *M = mats(T);
    float *outs[3] = {A.A_s, A.A_ts, A.A_cs};
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      if (!outs[m]) continue;
      float *o = outs[m] + (size_t)env * T.N * T.N + (size_t)r0 * T.N;
      const float *src = M + m * A.tile_rows * T.N;
      if ((nn & 3) == 0) {
        const tb_u4 *s4 = (const tb_u4 *)src;
        tb_u4 *o4 = (tb_u4 *)o;
        for (int i = lane; i < nn / 4; i += 64) TB_STREAM_STORE(&o4[i], s4[i]);
      } else {
        for (int i = lane; i < nn; i += 64) o[i] = src[i];
      }
    }
  }
};

// split launches: the phases a workgroup's role does not have return at once (ObsLane::role; R0 = its tile's first row, 0 for the
// rows' workgroup)
#define TRUSS_OBS_SCHEDULE(PH, PH_NS, T, A, R0)                          \
  if ((A).n_split > 1) {                                                 \
    PH(phase_nodes(T, A));                                               \
    PH(phase_tile_clear(T, A));                                          \
    PH(phase_edges(T, A, R0));                                           \
    PH(phase_store(T, A, R0));                                           \
  } else {                                                               \
    PH(phase_nodes(T, A));                                               \
    for (int r0_ = 0; r0_ < (T).N; r0_ += (A).tile_rows) {               \
      PH(phase_tile_clear(T, A));                                        \
      PH(phase_edges(T, A, r0_));                                        \
      PH(phase_store(T, A, r0_));                                        \
    }                                                                    \
  }
