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

struct TopoDev {
  int32_t N, E, NP, ndof, n_pad, n_rest, n_sym_nodes, n_sym_elems, n_sections, has_pairs;
  const int16_t *conn;       // [E][2]
  const int16_t *pairs;      // [NP][2]  (lo, hi) with lo < hi
  const uint8_t *nflags;     // [N]
  const int16_t *dofpos;     // [N][2]   solver position of (node, comp) or -1 when restrained
  const int16_t *restslot;   // [N][2]   reaction slot (reference order nsc-ndof-1) or -1
  const int32_t *asm_code;   // [E][10]  (band offset << 3) | (type << 1) | negate, or -1
  const int16_t *posnode;    // [n_pad]  node*2+comp at solver position, -1 = padding row
  const int16_t *sym_nodes;  // [n][2]   (dst, src)
  const int16_t *sym_elems;  // [n][2]
  const double *sec_area;   // [n_sections] area in m^2
  double e_mod, long_stress;
  // byte offsets inside one env's LDS region
  int32_t o_kb, o_zs, o_dinv, o_xsol, o_red, o_rbuf, o_ysh, o_xsh, o_tac, o_sec, env_stride;
};

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
};

#define TRUSS_NRED 6  // vol, dt, con1, con2, energy, (spare)

TRUSS_HD float tb_clamp01(float v) { return v > 1.0f ? 1.0f : (v < 0.0f ? 0.0f : v); }

// Python round(np.float32, 2): rint(v*100)/100 in float32 (truss2D_ENV.py:413)
TRUSS_HD float tb_round2(float v) { return rintf(v * 100.0f) / 100.0f; }

template <int G, int RPL, int EPL>
struct StepLane {
  static constexpr int W = G * RPL;
  static constexpr int EPB = 64 / G;  // envs per wave

  int lane, g, env, envc;
  bool active;
  char *L;  // this env's LDS region
  // per-env scalars
  double max_def, load_x, load_y;
  float ymax32, dmin32, ymd32, int1, int2;
  int is_roof, heads;
  // element stash
  double ek[EPL], ec[EPL], es[EPL], eA[EPL];
  int esec[EPL];
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
  TRUSS_HD double *dinv(const TopoDev &T) { return (double *)(L + T.o_dinv); }
  TRUSS_HD double *xsol(const TopoDev &T) { return (double *)(L + T.o_xsol); }
  TRUSS_HD double *red(const TopoDev &T) { return (double *)(L + T.o_red); }
  TRUSS_HD double *rbuf(const TopoDev &T) { return (double *)(L + T.o_rbuf); }
  TRUSS_HD float *ysh(const TopoDev &T) { return (float *)(L + T.o_ysh); }
  TRUSS_HD float *xsh(const TopoDev &T) { return (float *)(L + T.o_xsh); }
  TRUSS_HD float *tac(const TopoDev &T) { return (float *)(L + T.o_tac); }
  TRUSS_HD int32_t *secsh(const TopoDev &T) { return (int32_t *)(L + T.o_sec); }

  // ------------------------------------------------------------------------------------------
  TRUSS_HD void init(int lane_, int block, const TopoDev &T, const StepArgsDev &A, char *lds) {
    lane = lane_;
    g = lane % G;
    int grp = lane / G;
    env = block * EPB + grp;
    active = env < A.B;
    envc = active ? env : A.B - 1;
    L = lds + (size_t)grp * T.env_stride;
    const double *P = A.env_params + (size_t)envc * 8;
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
    heads = A.coin ? (A.coin[envc] != 0) : 0;
    bad = 0;
    p_vol = p_dt = p_en = 0.0;
    p_c1 = p_c2 = 0.0f;
  }

  // set_moveRange for one node (truss2D_GEN.py:118-133), float32
  TRUSS_HD void move_range(bool top, float y, float yp, float &up, float &dn) const {
    if (top) {
      up = fabsf(ymax32 - y);
      dn = fabsf((y - yp) - dmin32);
    } else if (is_roof) {
      up = fabsf((yp - y) - dmin32);
      dn = fabsf(y);
    } else {
      up = 0.0f;
      dn = 0.0f;
    }
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

  // ---- phase 1: action decode for the lane's node pairs, stage y/x/actions in LDS, clear K ----
  TRUSS_HD void phase_decode(const TopoDev &T, const StepArgsDev &A) {
    const size_t bn = (size_t)envc * T.N;
    float *Y = ysh(T), *X = xsh(T), *TA = tac(T);
    const bool decode = !(A.flags & TB_NO_DECODE);
    if (decode) {
      for (int p = g; p < T.NP; p += G) {
        int nd[2] = {T.pairs[2 * p], T.pairs[2 * p + 1]};
        float yv[2], mu[2], md[2], g0[2], g1[2];
        bool top[2], resy[2];
        for (int q = 0; q < 2; ++q) {
          int n = nd[q];
          yv[q] = A.y_in[bn + n];
          top[q] = T.nflags[n] & TF_TOP;
          resy[q] = T.nflags[n] & TF_RESY;
          g0[q] = tb_clamp01(A.a_geo[(bn + n) * 2 + 0]);
          g1[q] = tb_clamp01(A.a_geo[(bn + n) * 2 + 1]);
          float t0 = tb_clamp01(A.a_topo[(bn + n) * 3 + 0]);
          float t1 = tb_clamp01(A.a_topo[(bn + n) * 3 + 1]);
          float t2 = tb_clamp01(A.a_topo[(bn + n) * 3 + 2]);
          TA[n * 3 + 0] = t0;
          TA[n * 3 + 1] = t1;
          TA[n * 3 + 2] = t2;
          if ((A.flags & TB_CLAMP_INPLACE) && active) {
            A.a_geo[(bn + n) * 2 + 0] = g0[q];
            A.a_geo[(bn + n) * 2 + 1] = g1[q];
            A.a_topo[(bn + n) * 3 + 0] = t0;
            A.a_topo[(bn + n) * 3 + 1] = t1;
            A.a_topo[(bn + n) * 3 + 2] = t2;
          }
          X[n] = A.x[bn + n];
        }
        if (A.mu_in) {
          for (int q = 0; q < 2; ++q) {
            mu[q] = A.mu_in[bn + nd[q]];
            md[q] = A.md_in[bn + nd[q]];
          }
        } else {
          move_range(top[0], yv[0], yv[1], mu[0], md[0]);
          move_range(top[1], yv[1], yv[0], mu[1], md[1]);
        }
        float a = move_node(yv[0], g0[0], g1[0], mu[0], md[0], resy[0]);
        float b = move_node(yv[1], g0[1], g1[1], mu[1], md[1], resy[1]);
        // sequential repairs (truss2D_ENV.py:469-490): visit lo then hi in each of the three loops
        // loop 1: below y_min (= 0)
        if (a < 0.0f) {
          if (top[0]) { a = dmin32; b = 0.0f; } else { a = 0.0f; }
        }
        if (b < 0.0f) {
          if (top[1]) { b = dmin32; a = 0.0f; } else { b = 0.0f; }
        }
        // loop 2: above y_max
        if (a > ymax32) {
          if (top[0]) { a = ymax32; } else { a = ymd32; b = ymax32; }
        }
        if (b > ymax32) {
          if (top[1]) { b = ymax32; } else { b = ymd32; a = ymax32; }
        }
        // loop 3: pair closer than d_min -> lift the top node
        if (top[0] && fabsf(a - b) < dmin32) a = b + dmin32;
        if (top[1] && fabsf(b - a) < dmin32) b = a + dmin32;
        Y[nd[0]] = a;
        Y[nd[1]] = b;
      }
    } else {
      for (int n = g; n < T.N; n += G) {
        Y[n] = A.y_in[bn + n];
        X[n] = A.x[bn + n];
      }
    }
    // clear the band (identity on padding rows), the reaction buffer and the zero slot
    double *K = kb(T);
    const int tot = (T.n_pad + W) * W;
    for (int i = g; i < tot; i += G) K[i] = ((i % W) == 0 && (i / W) >= T.ndof) ? 1.0 : 0.0;
    double *RB = rbuf(T);
    for (int i = g; i < T.n_rest; i += G) RB[i] = 0.0;
    if (g == 0) xsol(T)[T.n_pad] = 0.0;
  }

  // ---- phase 2 (only when the topology has mirror tables): node symmetry ----
  TRUSS_HD void phase_sym_nodes(const TopoDev &T) {
    float *Y = ysh(T);
    for (int i = g; i < T.n_sym_nodes; i += G) {
      int dst = T.sym_nodes[2 * i], src = T.sym_nodes[2 * i + 1];
      float v = heads ? Y[src] : Y[dst];
      Y[dst] = v;
      Y[src] = v;
    }
  }

  // ---- phase 3: section update (truss2D_ENV.py:421-466) ----
  TRUSS_HD void phase_sizing(const TopoDev &T, const StepArgsDev &A) {
    const size_t be = (size_t)envc * T.E;
    const float *TA = tac(T);
    const bool decode = !(A.flags & TB_NO_DECODE);
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      int e = g + G * i;
      int s = 0;
      if (e < T.E) {
        s = A.sec_in[be + e];
        if (decode) {
          int n0 = T.conn[2 * e], n1 = T.conn[2 * e + 1];
          float p0 = TA[n0 * 3 + 0] + TA[n1 * 3 + 0];
          float p1 = TA[n0 * 3 + 1] + TA[n1 * 3 + 1];
          float p2 = TA[n0 * 3 + 2] + TA[n1 * 3 + 2];
          int am = 0;
          float best = p0;
          if (p1 > best) { am = 1; best = p1; }
          if (p2 > best) { am = 2; }
          if (am == 0) s = s > 0 ? s - 1 : 0;
          else if (am == 1) s = s < T.n_sections - 1 ? s + 1 : T.n_sections - 1;
        }
        if (T.n_sym_elems > 0) secsh(T)[e] = s;
      }
      esec[i] = s;
    }
  }

  TRUSS_HD void phase_sym_elems(const TopoDev &T) {
    int32_t *S = secsh(T);
    for (int i = g; i < T.n_sym_elems; i += G) {
      int a = T.sym_elems[2 * i], b = T.sym_elems[2 * i + 1];
      int m = S[a] < S[b] ? S[a] : S[b];
      S[a] = m;
      S[b] = m;
    }
  }

  TRUSS_HD void phase_sym_reload(const TopoDev &T) {
    const int32_t *S = secsh(T);
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      int e = g + G * i;
      if (e < T.E) esec[i] = S[e];
    }
  }

  TRUSS_HD double load_at(const TopoDev &T, int r) const {
    if (r >= T.n_pad) return 0.0;
    int nd = T.posnode[r];
    if (nd < 0) return 0.0;
    int fl = T.nflags[nd >> 1] & (is_roof ? TF_LOAD_ROOF : TF_LOAD_BRIDGE);
    if (!fl) return 0.0;
    return (nd & 1) ? load_y : load_x;
  }

  // ---- phase 4: element stiffness + scatter-add into the LDS band; load vector ----
  TRUSS_HD void phase_elements(const TopoDev &T, const StepArgsDev &A) {
    const float *Y = ysh(T), *X = xsh(T);
    double *K = kb(T);
    const size_t be = (size_t)envc * T.E;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      int e = g + G * i;
      ek[i] = ec[i] = es[i] = 0.0;
      eA[i] = 1.0;
      if (e < T.E) {
        int n0 = T.conn[2 * e], n1 = T.conn[2 * e + 1];
        double dx = (double)X[n1] - (double)X[n0];
        double dy = (double)Y[n1] - (double)Y[n0];
        double len = sqrt(dx * dx + dy * dy);
        double c = dx / len, s = dy / len;
        double Aa = T.sec_area[esec[i]];
        double k = T.e_mod * Aa / len;
        ek[i] = k;
        ec[i] = c;
        es[i] = s;
        eA[i] = Aa;
        p_vol += (double)(float)(Aa * len);  // all_v[i] is a float32 store (ENV:508)
        double kc = k * c, ks = k * s;
        const double kcc = kc * c, kcs = kc * s, kss = ks * s;
        const int32_t *code = T.asm_code + e * 10;
        for (int t = 0; t < 10; ++t) {
          int cd = code[t];
          if (cd >= 0) {
            int ty = (cd >> 1) & 3;
            double v = ty == 0 ? kcc : (ty == 1 ? kcs : kss);
            tb_lds_add(&K[cd >> 3], (cd & 1) ? -v : v);
          }
        }
        if (A.sec_out && active) A.sec_out[be + e] = esec[i];
      }
    }
    double *Z = zs(T);
    for (int r = g; r < T.n_pad + W; r += G) Z[r] = load_at(T, r);
  }

  // ---- solver -------------------------------------------------------------------------------
  TRUSS_HD void solver_init(const TopoDev &T) {
    const double *K = kb(T);
    const double *Z = zs(T);
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
      int r = g + G * s;
#pragma unroll
      for (int c = 0; c < W; ++c) R[s][c] = (c <= r) ? K[r * W + (r - c)] : K[c * W + (c - r)];
      rhs[s] = Z[r];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) xs[j] = 0.0;
  }

  // post this lane's entries of pivot column k (kk = k mod W, compile-time after unrolling)
  TRUSS_HD void pivot_write(const TopoDev &T, int k, int kk) {
    double *K = kb(T);
#pragma unroll
    for (int s = 0; s < RPL; ++s) K[k * W + g + G * s] = R[s][kk];
    if (g == kk % G) zs(T)[k] = rhs[kk / G];
  }

  TRUSS_HD void pivot_update(const TopoDev &T, int k, int kk) {
    const double *K = kb(T);
    double col[W];
#pragma unroll
    for (int j = 0; j < W; ++j) col[j] = K[k * W + j];
    const double zk = zs(T)[k];
    const double d = col[kk];
    if (!(d > 0.0)) bad = 1;
    const double inv = tb_rcp(d);
    if (g == kk % G) dinv(T)[k] = inv;
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
      double l = R[s][kk] * inv;
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (j != kk) R[s][j] = fma(-l, col[j], R[s][j]);
      rhs[s] = fma(-l, zk, rhs[s]);
    }
    // the window slides: column k leaves, column k+W enters; the pivot row's lane adopts row k+W
    const int rn = k + W;
    const double *Kn = K + rn * W;
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
      int o = (g + G * s - kk + W) % W;  // this row is k+o; it needs K[k+W][k+o] = band entry W-o
      R[s][kk] = Kn[(W - o) % W];
    }
    if (g == kk % G) {
      const int sp = kk / G;
#pragma unroll
      for (int j = 1; j < W; ++j) R[sp][(kk + j) % W] = Kn[W - j];
      rhs[sp] = zs(T)[rn];
    }
  }

  // x_k = (z_k - sum_m A[k+m,k] x_{k+m}) / d_k, every lane of the group redundantly
  TRUSS_HD void backsub_step(const TopoDev &T, int k, int kk) {
    const double *K = kb(T);
    double acc = zs(T)[k];
#pragma unroll
    for (int j = 0; j < W; ++j)
      if (j != kk) acc = fma(-K[k * W + j], xs[j], acc);
    double xk = acc * dinv(T)[k];
    xs[kk] = xk;
    if (g == 0) xsol(T)[k] = xk;
  }

  // ---- phase 6: member forces, stress ratios, reactions (FEM_2Dtruss.py:341-431) ----
  TRUSS_HD void phase_post_elements(const TopoDev &T, const StepArgsDev &A) {
    const double *XS = xsol(T);
    const size_t be = (size_t)envc * T.E;
    const int zslot = T.n_pad;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      int e = g + G * i;
      if (e < T.E) {
        int n0 = T.conn[2 * e], n1 = T.conn[2 * e + 1];
        int p0 = T.dofpos[2 * n0], p1 = T.dofpos[2 * n0 + 1], p2 = T.dofpos[2 * n1], p3 = T.dofpos[2 * n1 + 1];
        double v0 = XS[p0 < 0 ? zslot : p0], v1 = XS[p1 < 0 ? zslot : p1];
        double v2 = XS[p2 < 0 ? zslot : p2], v3 = XS[p3 < 0 ? zslot : p3];
        double c = ec[i], s = es[i], k = ek[i];
        double u0 = c * v0 + s * v1;
        double u2 = c * v2 + s * v3;
        double q = k * u0 + (-k) * u2;
        double srr = fabs(q / eA[i]) / T.long_stress;
        float srf = (float)srr;
        p_c1 = fmaxf(p_c1, fabsf(srf));
        if (active) {
          A.q0[be + e] = (float)q;
          A.sr[be + e] = srf;
          A.comp[be + e] = q > 0.0 ? 1 : 0;
          if (A.q064) A.q064[be + e] = q;
        }
        if (A.react) {
          // f = T^T q = q0 * [c, s, -c, -s] summed into the restrained DOFs (FEM:389-411)
          int s0 = T.restslot[2 * n0], s1 = T.restslot[2 * n0 + 1];
          int s2 = T.restslot[2 * n1], s3 = T.restslot[2 * n1 + 1];
          if (s0 >= 0) tb_lds_add(&rbuf(T)[s0], q * c);
          if (s1 >= 0) tb_lds_add(&rbuf(T)[s1], q * s);
          if (s2 >= 0) tb_lds_add(&rbuf(T)[s2], q * (-c));
          if (s3 >= 0) tb_lds_add(&rbuf(T)[s3], q * (-s));
        }
      }
    }
  }

  // ---- phase 7: nodal outputs, move ranges of the new design, objective partials ----
  TRUSS_HD void phase_post_nodes(const TopoDev &T, const StepArgsDev &A) {
    const double *XS = xsol(T);
    const float *Y = ysh(T);
    const size_t bn = (size_t)envc * T.N;
    const int zslot = T.n_pad;
    for (int n = g; n < T.N; n += G) {
      int px = T.dofpos[2 * n], py = T.dofpos[2 * n + 1];
      double dx = XS[px < 0 ? zslot : px], dy = XS[py < 0 ? zslot : py];
      bool top = T.nflags[n] & TF_TOP;
      float y = Y[n];
      if (top) {
        float tg = A.target[bn + n];
        p_dt += (double)fabsf(tg - y);  // all_dt (ENV:514)
      } else {
        p_c2 = fmaxf(p_c2, fabsf((float)(dy / max_def)));  // all_d (ENV:516)
      }
      if (active) {
        A.y_out[bn + n] = y;
        A.disp[(bn + n) * 2 + 0] = (float)dx;
        A.disp[(bn + n) * 2 + 1] = (float)dy;
        if (A.disp64) {
          A.disp64[(bn + n) * 2 + 0] = dx;
          A.disp64[(bn + n) * 2 + 1] = dy;
        }
      }
    }
    if (A.mu_out && T.has_pairs) {
      for (int p = g; p < T.NP; p += G) {
        int lo = T.pairs[2 * p], hi = T.pairs[2 * p + 1];
        float ul, dl, uh, dh;
        move_range(T.nflags[lo] & TF_TOP, Y[lo], Y[hi], ul, dl);
        move_range(T.nflags[hi] & TF_TOP, Y[hi], Y[lo], uh, dh);
        if (active) {
          A.mu_out[bn + lo] = ul;
          A.md_out[bn + lo] = dl;
          A.mu_out[bn + hi] = uh;
          A.md_out[bn + hi] = dh;
        }
      }
    }
    if (A.energy)
      for (int r = g; r < T.n_pad; r += G) p_en += XS[r] * load_at(T, r);
    double *RD = red(T);
    RD[0 * G + g] = p_vol;
    RD[1 * G + g] = p_dt;
    RD[2 * G + g] = (double)p_c1;
    RD[3 * G + g] = (double)p_c2;
    RD[4 * G + g] = p_en;
  }

  // ---- phase 8: one lane per env folds the partials (fixed order) and writes point ----
  TRUSS_HD void phase_finish(const TopoDev &T, const StepArgsDev &A) {
    if (g != 0 || !active) return;
    const double *RD = red(T);
    double vol = 0.0, dt = 0.0, en = 0.0;
    float c1 = 0.0f, c2 = 0.0f;
    for (int j = 0; j < G; ++j) {
      vol += RD[0 * G + j];
      dt += RD[1 * G + j];
      c1 = fmaxf(c1, (float)RD[2 * G + j]);
      c2 = fmaxf(c2, (float)RD[3 * G + j]);
      en += RD[4 * G + j];
    }
    float obj1 = (float)vol, obj2 = (float)dt;
    float *pt = A.point + (size_t)env * 4;
    pt[0] = obj1 / int1;
    pt[1] = obj2 / int2;
    pt[2] = c1;
    pt[3] = c2;
    if (A.obj) {
      A.obj[(size_t)env * 2 + 0] = obj1;
      A.obj[(size_t)env * 2 + 1] = obj2;
    }
    if (A.energy) A.energy[env] = 0.5 * en;
    if (A.react) {
      const double *RB = rbuf(T);
      for (int i = 0; i < T.n_rest; ++i) A.react[(size_t)env * T.n_rest + i] = RB[i];
    }
    if (A.status) A.status[env] = bad;
  }
};

// The phase schedule, shared by the HIP kernel and the emulator.
//   PH(call)    run `ln.call` for every lane, then a workgroup barrier
//   PH_NS(call) run it without a trailing barrier (no lane reads what another lane writes in it
//               before the next barrier)
//   BAR()       explicit barrier
// W_ must be a constexpr in scope; TRUSS_UNROLL expands to the unroll pragma on the GPU so that the
// register-window indices (kk_) are compile-time constants.
#define TRUSS_STEP_SCHEDULE(PH, PH_NS, BAR, T, A)                                   \
  PH(phase_decode(T, A));                                                           \
  if ((T).n_sym_nodes > 0 && !((A).flags & TB_NO_DECODE)) { PH(phase_sym_nodes(T)); } \
  PH(phase_sizing(T, A));                                                           \
  if ((T).n_sym_elems > 0) {                                                        \
    if (!((A).flags & TB_NO_DECODE)) { PH(phase_sym_elems(T)); }                    \
    PH(phase_sym_reload(T));                                                        \
  }                                                                                 \
  PH(phase_elements(T, A));                                                         \
  PH(solver_init(T));                                                               \
  for (int kb_ = 0; kb_ < (T).n_pad; kb_ += W_) {                                   \
    TRUSS_UNROLL                                                                    \
    for (int kk_ = 0; kk_ < W_; ++kk_) {                                            \
      PH(pivot_write(T, kb_ + kk_, kk_));                                           \
      PH_NS(pivot_update(T, kb_ + kk_, kk_));                                       \
    }                                                                               \
  }                                                                                 \
  BAR();                                                                            \
  for (int kb_ = (T).n_pad - W_; kb_ >= 0; kb_ -= W_) {                             \
    TRUSS_UNROLL                                                                    \
    for (int kk_ = W_ - 1; kk_ >= 0; --kk_) { PH_NS(backsub_step(T, kb_ + kk_, kk_)); } \
  }                                                                                 \
  BAR();                                                                            \
  PH(phase_post_elements(T, A));                                                    \
  PH(phase_post_nodes(T, A));                                                       \
  PH_NS(phase_finish(T, A));

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
};

struct ObsLane {
  int lane, env;
  char *L;
  float maxdef32;
  int is_roof;

  // LDS: raw[N][13] | mn[13] | mx[13] | (pad to 16) | As[N][N] | Ats[N][N] | Acs[N][N]
  TRUSS_HD float *raw() { return (float *)L; }
  TRUSS_HD float *mn(const TopoDev &T) { return raw() + T.N * 13; }
  TRUSS_HD float *mx(const TopoDev &T) { return mn(T) + 13; }
  TRUSS_HD float *mats(const TopoDev &T) { return (float *)(L + (((size_t)(T.N * 13 + 26) * 4 + 15) & ~(size_t)15)); }

  TRUSS_HD void init(int lane_, int block, const TopoDev &, const ObsArgsDev &A, char *lds) {
    lane = lane_;
    env = block;
    L = lds;
    const double *P = A.env_params + (size_t)env * 8;
    maxdef32 = (float)P[2];
    is_roof = P[7] != 0.0;
  }

  TRUSS_HD void phase_nodes(const TopoDev &T, const ObsArgsDev &A) {
    const size_t bn = (size_t)env * T.N;
    float *R = raw();
    for (int n = lane; n < T.N; n += 64) {
      const int fl = T.nflags[n];
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
    float *M = mats(T);
    const int tot = 3 * T.N * T.N;
    for (int i = lane; i < tot; i += 64) M[i] = 0.0f;
  }

  TRUSS_HD void phase_edges(const TopoDev &T, const ObsArgsDev &A) {
    const float *R = raw();
    if (lane < 13) {  // column min / max for the normalisation (ENV:102)
      float lo = R[lane], hi = R[lane];
      for (int n = 1; n < T.N; ++n) {
        float v = R[n * 13 + lane];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
      }
      mn(T)[lane] = lo;
      mx(T)[lane] = hi;
    }
    const size_t be = (size_t)env * T.E;
    float *As = mats(T), *Ats = As + T.N * T.N, *Acs = Ats + T.N * T.N;
    const double amax = T.sec_area[T.n_sections - 1];
    for (int e = lane; e < T.E; e += 64) {
      const int a = T.conn[2 * e], b = T.conn[2 * e + 1];
      const int s = A.sec[be + e];
      const double area = T.sec_area[s];
      const float srv = A.sr[be + e];
      const int cmp = A.comp[be + e];
      const float vs = (float)(area / amax);
      const float val = fminf(srv, 1.0f) * (srv > 1.0f ? 1.0f : 0.5f);
      As[a * T.N + b] = vs;
      As[b * T.N + a] = vs;
      if (cmp == 0) {
        Ats[a * T.N + b] = val;
        Ats[b * T.N + a] = val;
      } else {
        Acs[a * T.N + b] = val;
        Acs[b * T.N + a] = val;
      }
      if (A.nxe) {
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

  TRUSS_HD void phase_store(const TopoDev &T, const ObsArgsDev &A) {
    const float *R = raw();
    if (A.x_n) {
      float *o = A.x_n + (size_t)env * T.N * 13;
      const float *lo = mn(T), *hi = mx(T);
      for (int i = lane; i < T.N * 13; i += 64) {
        int c = i % 13;
        o[i] = (R[i] - lo[c]) / (hi[c] - lo[c] + 1e-6f);
      }
    }
    const int nn = T.N * T.N;
    const float *M = mats(T);
    float *outs[3] = {A.A_s, A.A_ts, A.A_cs};
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      if (!outs[m]) continue;
      float *o = outs[m] + (size_t)env * nn;
      const float *src = M + m * nn;
      if ((nn & 3) == 0) {
        const float4 *s4 = (const float4 *)src;
        float4 *o4 = (float4 *)o;
        for (int i = lane; i < nn / 4; i += 64) o4[i] = s4[i];
      } else {
        for (int i = lane; i < nn; i += 64) o[i] = src[i];
      }
    }
  }
};

#define TRUSS_OBS_SCHEDULE(PH, PH_NS, T, A) \
  PH(phase_nodes(T, A));                    \
  PH(phase_edges(T, A));                    \
  PH_NS(phase_store(T, A));
