// truss_host.h -- host side of the C ABI (include/truss_mi355.h): topology tables, DOF numbering,
// band ordering, LDS layout, argument checking and kernel selection.  Header-only and backend
// agnostic: the includer provides
//     TRUSS_BACKEND_NAME                           "hip" | "emu"
//     void *tb_dev_alloc(size_t);  void tb_dev_free(void *);
//     bool tb_dev_upload(void *dst, const void *src, size_t bytes);
//     int  tb_launch_step(const truss_topo *, const StepArgsDev &, bool emit, void *stream);
//          (emit: the instantiation that also writes the observation tensors; only asked for when dev.emit_ok)
//     int  tb_launch_obs(const truss_topo *, const ObsArgsDev &, void *stream);
//     int  tb_launch_rollout(const truss_topo *, const StepArgsDev &, int n_steps, int n_sets, void *stream);
//          (all chained steps in one launch; only asked for when tb_rollout_one_launch() says the topology allows it)
// truss_hip.hip implements them with the HIP runtime; tests/emu/truss_emu.cpp with malloc and the
// CPU lane emulator.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <vector>

#include "../../include/truss_mi355.h"

static thread_local std::string g_truss_err;
static int tb_fail(int code, const std::string &msg) {
  g_truss_err = msg;
  return code;
}

// compiled (G, RPL, EPL) instantiations of the step kernel; keep in sync with the dispatch tables
struct TbVariant {
  int G, WL, RPL, EPL;  // lanes per env, window lanes, rows per lane, elements per lane
};
#ifdef TRUSS_ONLY_DEFAULT_VARIANT   // diagnostic builds (tools/ablate.sh, `make diag`): compile one kernel only
#ifndef TRUSS_DIAG_VARIANT           // -D'TRUSS_DIAG_VARIANT(X)=X(64,8,1,10)' -DTRUSS_DIAG_NO_EMIT for another one
#define TRUSS_DIAG_VARIANT(X) X(16, 8, 1, 5)
#endif
#define TRUSS_VARIANTS(X) TRUSS_DIAG_VARIANT(X)
#else
#define TRUSS_VARIANTS(X) \
  X(8, 8, 1, 5) X(8, 8, 1, 10) X(16, 8, 1, 3) X(16, 8, 1, 5) X(32, 8, 1, 3) X(16, 16, 1, 5) X(16, 8, 2, 5) X(4, 4, 2, 20) \
  X(32, 8, 1, 10) X(64, 8, 1, 10) /* large trusses: up to 320 / 640 elements, 128 / 256 nodes (BASELINE config 5) */
#endif
// variants that are also compiled with the fused observation emission (StepLane<..., EMIT = true>)
#if defined(TRUSS_ONLY_DEFAULT_VARIANT) && defined(TRUSS_DIAG_NO_EMIT)
#define TRUSS_EMIT_VARIANTS(X)
#elif defined(TRUSS_ONLY_DEFAULT_VARIANT)
#define TRUSS_EMIT_VARIANTS(X) TRUSS_DIAG_VARIANT(X)
#else
#define TRUSS_EMIT_VARIANTS(X) X(8, 8, 1, 5) X(16, 8, 1, 3) X(16, 8, 1, 5)
#endif
// variants with a persistent rollout kernel (truss_rollout as one launch)
#ifdef TRUSS_ONLY_DEFAULT_VARIANT
#define TRUSS_ROLLOUT_VARIANTS(X) TRUSS_DIAG_VARIANT(X)
#else
#define TRUSS_ROLLOUT_VARIANTS(X) X(8, 8, 1, 5) X(8, 8, 1, 10) X(16, 8, 1, 3) X(16, 8, 1, 5) X(32, 8, 1, 3) X(32, 8, 1, 10) X(64, 8, 1, 10)
#endif
static bool tb_variant_rolls(int G, int WL, int RPL, int EPL) {
#define X(g, wl, r, e) \
  if (G == g && WL == wl && RPL == r && EPL == e) return true;
  TRUSS_ROLLOUT_VARIANTS(X)
#undef X
  return false;
}
static bool tb_variant_emits(int G, int WL, int RPL, int EPL) {
#define X(g, wl, r, e) \
  if (G == g && WL == wl && RPL == r && EPL == e) return true;
  TRUSS_EMIT_VARIANTS(X)
#undef X
  return false;
}
static const TbVariant kVariants[] = {
#define X(g, wl, r, e) {g, wl, r, e},
    TRUSS_VARIANTS(X)
#undef X
};
static const int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));

// observation kernel LDS: raw[N][13] | mn[13] | mx[13] | part[128] | (pad to 16) | three [TR][N] matrix tiles.
// TR = rows per tile: all N rows when the three N x N matrices fit (one pass, 32-node trusses: 12 KB); large
// trusses are written in row tiles (256 nodes: 3 x 256 KB would not fit any CU).
static int tb_env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// Trusses of 64 nodes and more: a batch has few envs and each writes 70-870 KB -- one workgroup per (env, tile of rows) with small
// tiles (64 nodes: 8 rows = 6 KB, 128 nodes: 8 rows, 256 nodes: 4 rows = 12 KB) plus one workgroup per env for the per-node / per-element
// rows, dispatched first (it is the longest), instead of one workgroup per env walking over 96 KB tiles: 256 envs of 256 nodes were 256
// waves on 1024 SIMDs, 2.1 TB/s; now 5.8 TB/s (128 nodes: 2.1 -> 5.1, 64 nodes: 3.3 -> 4.0; tools/obs_probe.py).
// TRUSS_OBS_SPLIT_MIN_N / TRUSS_OBS_TILE_KB: diagnostic overrides for sweeps.
static inline bool tb_obs_splits(int N) { return N >= tb_env_int("TRUSS_OBS_SPLIT_MIN_N", 64); }
static inline int tb_obs_tile_rows(int N) {
  if (tb_obs_splits(N)) {
    const int tr = (int)(((size_t)tb_env_int("TRUSS_OBS_TILE_KB", N >= 128 ? 12 : 6) * 1024) / ((size_t)3 * N * 4)) & ~1;
    return tr < 2 ? 2 : tr;
  }
  const size_t budget = 96 * 1024;
  size_t tr = budget / ((size_t)3 * N * 4);
  if (tr >= (size_t)N) {
    // everything would fit in one pass -- but two passes of half the rows halve the LDS of a workgroup, and
    // the launch then fits the CUs in one round (4096 envs x 13.8 KB did not: 26.8 -> 24.9 us at 32 nodes)
    tr = N >= 16 ? (size_t)(((N + 1) / 2 + 3) & ~3) : (size_t)N;
  }
  if (tr > (size_t)N) tr = (size_t)N;
  return tr < 1 ? 1 : (int)tr;
}
static inline size_t tb_obs_lds_bytes(int N) {
  return ((((size_t)(N * 13 + 26 + 128) * 4 + 15) & ~(size_t)15) + (size_t)3 * tb_obs_tile_rows(N) * N * 4 + 15) & ~(size_t)15;
}


template <typename T>
static size_t tb_push(std::vector<char> &blob, const std::vector<T> &v) {
  size_t off = (blob.size() + 15) & ~size_t(15);
  blob.resize(off + std::max<size_t>(v.size(), 1) * sizeof(T));
  if (!v.empty()) memcpy(blob.data() + off, v.data(), v.size() * sizeof(T));
  return off;
}

struct truss_topo {
  int N = 0, E = 0, NP = 0, ndof = 0, n_pad = 0, n_rest = 0, bw = 0;
  int G = 0, WL = 0, RPL = 0, EPL = 0, W = 0, variant = -1;
  std::vector<int32_t> nsc, ttnsc, perm;
  TopoDev dev{};        // what a plain step launch gets
  TopoDev dev_emit{};   // ... an EMIT launch: the staged tables include the nN_x_e gather table (larger blob_bytes / o_env0)
  size_t lds_bytes_emit = 0;
  void *blob = nullptr;
  void *etab = nullptr;   // emission tables of the fused observation writer (TopoDev::etab)
  size_t lds_bytes = 0;
  int n_sections = 0;
};

// ---- fused observation emission: feature bank placement + gather tables (TopoDev "emit" fields) -------------
// Dead byte ranges of one env's LDS region, by the time from which the streaming wave may overwrite them:
//   early = once the member results are staged (behind phase_post_elements): band behind the staging rows and the
//           objective partials; action rows
//   late  = behind phase_post_nodes: solver scratch (z vectors, solution vector, trash slots -- not the reactions
//           and the objective partials, so that the placement does not depend on where phase_finish runs)
// Builds the register tables (t->etab, uploaded here) and appends the LDS-resident nN_x_e table to `blob`.
// Returns false when the topology is outside what the fused writer covers; truss_step then runs the observation
// kernel as a second launch.
struct TbFrag {
  size_t lo, hi;
};
static bool tb_build_emit(truss_topo *t, const int32_t *conn, const uint8_t *res, const uint8_t *top,
                          std::vector<TbFrag> early, std::vector<TbFrag> late, std::vector<char> &blob) {
  TopoDev &D = t->dev;
  const int N = t->N, E = t->E, G = t->G;
  D.emit_ok = 0;
  if (!tb_variant_emits(t->G, t->WL, t->RPL, t->EPL) || tb_env_int("TRUSS_NO_FUSED_OBS", 0)) return false;
  const int NPL = (2 * t->EPL + 4) / 5, NF = G * NPL, EF = G * t->EPL;
  if ((N & 3) || N > NF || E > EF || !D.has_pairs) return false;
  // the kernel's compile-time iteration counts (StepLane::IX ...)
  const int IX = (13 * NF / 4 + G - 1) / G, IN_ = (12 * NF / 4 + G - 1) / G, IE = (21 * EF / 4 + G - 1) / G,
            IM = (NF * NF / 4 + G - 1) / G;
  // first-fit placement of the bank arrays (sizes in floats; all multiples of 4 but the constants, placed last)
  struct Arr {
    int32_t *dst;
    int n;
  };
  auto place = [](std::vector<Arr> arrs, std::vector<TbFrag> &frags) {
    for (auto &f : frags) f.lo = (f.lo + 15) & ~size_t(15);
    for (const Arr &a : arrs) {
      bool placed = false;
      for (auto &f : frags)
        if (f.lo + (size_t)a.n * 4 <= f.hi) {
          *a.dst = (int32_t)(f.lo / 4);
          f.lo += ((size_t)a.n * 4 + 15) & ~size_t(15);
          placed = true;
          break;
        }
      if (!placed) return false;
    }
    return true;
  };
  if (!place({{&D.b_tc, 2 * (E + 1)}, {&D.b_erec, 4 * E}, {&D.b_erec2, 2 * E}, {&D.b_const, 3}}, early)) return false;   // written with / right behind the member results
  for (const TbFrag &f : early) late.push_back(f);                                   // what the element records left over
  if (!place({{&D.b_nna, 4 * N}, {&D.b_nnb, 4 * N}, {&D.b_nraw, 4 * N}, {&D.b_nn8, N}}, late)) return false;
  if (D.env_stride / 4 > 65535 || E + 1 > 65535) return false;
  const int ZERO = D.b_const, ONE = D.b_const + 1, C1 = D.b_const + 2;
  const int fX = D.o_x / 4, fY = D.o_y / 4, fMU = (D.o_kb + D.so_mu) / 4, fMD = (D.o_kb + D.so_md) / 4,
            fQ0 = (D.o_kb + D.so_q0) / 4;
  // 0/1 columns of x_n (res_x, res_y, top, 1 - top): min-max normalisation of a constant column gives 0
  auto flag = [&](int n, int c) -> int { return c == 2 ? res[2 * n] != 0 : c == 3 ? res[2 * n + 1] != 0 : c == 5 ? top[n] != 0 : top[n] == 0; };
  bool any0[13] = {}, any1[13] = {};
  for (int n = 0; n < N; ++n)
    for (int c : {2, 3, 5, 6}) (flag(n, c) ? any1[c] : any0[c]) = true;
  const int dyn_of[13] = {0, 1, -1, -1, 2, -1, -1, 3, 4, 5, 6, 7, 8};
  std::vector<uint16_t> txn, tnxn, tnxe, tmat;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < 13; ++c)
      txn.push_back((uint16_t)(dyn_of[c] < 0   ? (flag(n, c) && any0[c] ? C1 : ZERO)
                               : dyn_of[c] < 4 ? D.b_nna + 4 * n + dyn_of[c]
                               : dyn_of[c] < 8 ? D.b_nnb + 4 * n + dyn_of[c] - 4
                                               : D.b_nn8 + n));
  for (int n = 0; n < N; ++n) {
    const int nr = D.b_nraw + 4 * n;   // loaded, target / y, |dy|, violated
    const int src[12] = {fX + n, fY + n, res[2 * n] ? ONE : ZERO, res[2 * n + 1] ? ONE : ZERO, nr + 0, top[n] ? ONE : ZERO,
                         top[n] ? ZERO : ONE, fMU + n, fMD + n, nr + 1, nr + 2, nr + 3};
    for (int c = 0; c < 12; ++c) tnxn.push_back((uint16_t)src[c]);
  }
  for (int e = 0; e < E; ++e) {
    const int er = D.b_erec + 4 * e, er2 = D.b_erec2 + 2 * e;   // (sec, area, length, tension), (compression, violated)
    const int own[7] = {er + 0, er + 1, er + 2, er + 3, er2 + 0, fQ0 + e, er2 + 1};
    for (int c = 0; c < 7; ++c) tnxe.push_back((uint16_t)own[c]);
    for (int q = 0; q < 2; ++q) {
      const int n = conn[2 * e + q];
      const int nr = D.b_nraw + 4 * n;
      const int nd[7] = {fX + n, fY + n, res[2 * n] ? ONE : ZERO, res[2 * n + 1] ? ONE : ZERO, nr + 0, nr + 2, nr + 3};
      for (int c = 0; c < 7; ++c) tnxe.push_back((uint16_t)nd[c]);
    }
  }
  tmat.assign((size_t)N * N, (uint16_t)E);   // row-major = chunk order
  for (int e = 0; e < E; ++e) {
    tmat[(size_t)conn[2 * e] * N + conn[2 * e + 1]] = (uint16_t)e;
    tmat[(size_t)conn[2 * e + 1] * N + conn[2 * e]] = (uint16_t)e;
  }
  D.nc_xn = (int32_t)txn.size() / 4;
  D.nc_nxn = (int32_t)tnxn.size() / 4;
  D.nxe_cw = (E & 3) == 0 ? 4 : (E & 1) == 0 ? 2 : 1;    // 21 E floats per env: 16-, 8- or 4-byte aligned rows
  D.nc_nxe = (int32_t)tnxe.size() / D.nxe_cw;
  D.nc_mat = (int32_t)tmat.size() / 4;
  // pad to `iters` iterations of G chunks (entries past the end repeat the last chunk); chunk q = iteration q / G, lane q % G
  auto padded = [&](const std::vector<uint16_t> &v, int iters) {
    const size_t nc = v.size() / 4;
    std::vector<uint16_t> out((size_t)iters * G * 4);
    for (size_t q = 0; q < (size_t)iters * G; ++q)
      for (int j = 0; j < 4; ++j) out[4 * q + j] = v[4 * std::min(q, nc - 1) + j];
    return out;
  };
  // register tables: [pair][G][2]: the entries of iterations (2p, 2p+1) of a lane are adjacent
  auto paired = [&](const std::vector<uint16_t> &v, int iters) {
    const int ip = (iters + 1) / 2;
    std::vector<uint16_t> lin = padded(v, 2 * ip), out(lin.size());
    for (int p2 = 0; p2 < ip; ++p2)
      for (int g = 0; g < G; ++g)
        for (int h = 0; h < 2; ++h)
          for (int j = 0; j < 4; ++j) out[(((size_t)p2 * G + g) * 2 + h) * 4 + j] = lin[((size_t)(2 * p2 + h) * G + g) * 4 + j];
    return out;
  };
  std::vector<char> tab;
  D.et_xn = (int32_t)tb_push(tab, paired(txn, IX));
  D.et_nxn = (int32_t)tb_push(tab, paired(tnxn, IN_));
  D.et_mat = (int32_t)tb_push(tab, paired(tmat, IM));
  tab.resize((tab.size() + 15) & ~size_t(15));
  t->etab = tb_dev_alloc(tab.size());
  if (!t->etab || !tb_dev_upload(t->etab, tab.data(), tab.size())) {
    if (t->etab) tb_dev_free(t->etab);
    t->etab = nullptr;
    return false;
  }
  D.etab = (const char *)t->etab;
  D.f_tnxe = (int32_t)tb_push(blob, D.nxe_cw == 4 ? padded(tnxe, IE) : tnxe);   // LDS-resident, staged by EMIT launches only
  blob.resize((blob.size() + 15) & ~size_t(15));
  D.emit_ok = 1;
  return true;
}

// --- DOF numbering, FEM_2Dtruss.py:227-261 ---------------------------------------------------
static void tb_dof_numbering(const uint8_t *res, int N, std::vector<int32_t> &nsc, int &ndof) {
  nsc.assign(2 * N, 0);
  int c = 1;
  for (int i = 0; i < 2 * N; ++i)
    if (res[i] == 0) nsc[i] = c++;
  ndof = c - 1;
  for (int i = 0; i < 2 * N; ++i)
    if (res[i] != 0) nsc[i] = c++;
}

// half bandwidth (in DOFs) of K when nodes are visited in `order`
static int tb_bandwidth(const std::vector<int> &order, const uint8_t *res, const int32_t *conn, int N, int E,
                        std::vector<int> *dofpos_out) {
  std::vector<int> pos(2 * N, -1);
  int c = 0;
  for (int i = 0; i < N; ++i) {
    int n = order[i];
    for (int k = 0; k < 2; ++k)
      if (res[2 * n + k] == 0) pos[2 * n + k] = c++;
  }
  int bw = 0;
  for (int n = 0; n < N; ++n)
    if (pos[2 * n] >= 0 && pos[2 * n + 1] >= 0) bw = std::max(bw, std::abs(pos[2 * n] - pos[2 * n + 1]));
  for (int e = 0; e < E; ++e) {
    int a = conn[2 * e], b = conn[2 * e + 1];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) {
        int pa = pos[2 * a + i], pb = pos[2 * b + j];
        if (pa >= 0 && pb >= 0) bw = std::max(bw, std::abs(pa - pb));
      }
  }
  if (dofpos_out) *dofpos_out = pos;
  return bw;
}

// reverse Cuthill-McKee from `start` on the node graph
static std::vector<int> tb_rcm(int start, const std::vector<std::vector<int>> &adj, int N) {
  std::vector<int> order;
  std::vector<char> seen(N, 0);
  auto bfs = [&](int s) {
    std::queue<int> q;
    q.push(s);
    seen[s] = 1;
    while (!q.empty()) {
      int u = q.front();
      q.pop();
      order.push_back(u);
      std::vector<int> nb;
      for (int v : adj[u])
        if (!seen[v]) {
          seen[v] = 1;
          nb.push_back(v);
        }
      std::sort(nb.begin(), nb.end(), [&](int a, int b) {
        return adj[a].size() != adj[b].size() ? adj[a].size() < adj[b].size() : a < b;
      });
      for (int v : nb) q.push(v);
    }
  };
  bfs(start);
  for (int i = 0; i < N; ++i)
    if (!seen[i]) bfs(i);
  std::reverse(order.begin(), order.end());
  return order;
}

extern "C" int truss_abi_version(void) { return TRUSS_ABI_VERSION; }
extern "C" const char *truss_last_error(void) { return g_truss_err.c_str(); }
extern "C" const char *truss_backend(void) { return TRUSS_BACKEND_NAME; }

extern "C" int truss_topo_create(truss_topo_t **out, int32_t N, int32_t E, const int32_t *conn, const uint8_t *res,
                                 const uint8_t *top, const int32_t *pair, const uint8_t *load_mask,
                                 int32_t n_sym_nodes, const int32_t *sym_nodes, int32_t n_sym_elems,
                                 const int32_t *sym_elems, int32_t n_sections, const double *sections, double e_mod,
                                 double long_stress, const int32_t *node_order) {
  if (!out || !conn || !res || !top || !sections) return tb_fail(TRUSS_EINVAL, "NULL argument");
  if (N < 2 || E < 1 || N > 16000) return tb_fail(TRUSS_EINVAL, "bad N/E");
  if (n_sections < 1) return tb_fail(TRUSS_EINVAL, "need at least one section");
  if ((n_sym_nodes > 0 && !sym_nodes) || (n_sym_elems > 0 && !sym_elems)) return tb_fail(TRUSS_EINVAL, "sym table NULL");
  for (int e = 0; e < E; ++e)
    if (conn[2 * e] < 0 || conn[2 * e] >= N || conn[2 * e + 1] < 0 || conn[2 * e + 1] >= N || conn[2 * e] == conn[2 * e + 1])
      return tb_fail(TRUSS_EINVAL, "conn out of range");
  for (int i = 0; i < n_sym_nodes * 2; ++i)
    if (sym_nodes[i] < 0 || sym_nodes[i] >= N) return tb_fail(TRUSS_EINVAL, "sym_nodes out of range");
  for (int i = 0; i < n_sym_elems * 2; ++i)
    if (sym_elems[i] < 0 || sym_elems[i] >= E) return tb_fail(TRUSS_EINVAL, "sym_elems out of range");

  for (int e = 0; e < E; ++e)
    for (int f = e + 1; f < E; ++f)
      if ((conn[2 * e] == conn[2 * f] && conn[2 * e + 1] == conn[2 * f + 1]) ||
          (conn[2 * e] == conn[2 * f + 1] && conn[2 * e + 1] == conn[2 * f]))
        return tb_fail(TRUSS_EUNSUPPORTED, "two elements join the same pair of nodes (parallel members are not supported)");
  truss_topo *t = new truss_topo();
  t->N = N;
  t->E = E;
  t->n_sections = n_sections;
  tb_dof_numbering(res, N, t->nsc, t->ndof);
  if (t->ndof < 1) {
    delete t;
    return tb_fail(TRUSS_EINVAL, "no free DOF");
  }
  t->n_rest = 2 * N - t->ndof;
  t->ttnsc.resize(4 * E);
  for (int e = 0; e < E; ++e) {  // FEM_2Dtruss.py:311-317
    int a = conn[2 * e], b = conn[2 * e + 1];
    t->ttnsc[4 * e + 0] = t->nsc[2 * a];
    t->ttnsc[4 * e + 1] = t->nsc[2 * a + 1];
    t->ttnsc[4 * e + 2] = t->nsc[2 * b];
    t->ttnsc[4 * e + 3] = t->nsc[2 * b + 1];
  }
  // vertical pairs
  std::vector<int16_t> pairs;
  if (pair) {
    for (int i = 0; i < N; ++i) {
      int j = pair[i];
      if (j < 0 || j >= N || j == i || pair[j] != i) {
        delete t;
        return tb_fail(TRUSS_EINVAL, "pair[] must be an involution without fixed points");
      }
      if (i < j) {
        pairs.push_back((int16_t)i);
        pairs.push_back((int16_t)j);
      }
    }
  }
  t->NP = (int)pairs.size() / 2;

  // ---- node ordering for the banded solver: hint, natural, RCM from every start ----
  std::vector<std::vector<int>> adj(N);
  for (int e = 0; e < E; ++e) {
    adj[conn[2 * e]].push_back(conn[2 * e + 1]);
    adj[conn[2 * e + 1]].push_back(conn[2 * e]);
  }
  std::vector<std::vector<int>> cands;
  if (node_order) {
    std::vector<int> o(node_order, node_order + N);
    std::vector<char> seen(N, 0);
    bool ok = true;
    for (int v : o) {
      if (v < 0 || v >= N || seen[v]) ok = false;
      else seen[v] = 1;
    }
    if (!ok) {
      delete t;
      return tb_fail(TRUSS_EINVAL, "node_order is not a permutation");
    }
    cands.push_back(o);
  }
  {
    std::vector<int> nat(N);
    for (int i = 0; i < N; ++i) nat[i] = i;
    cands.push_back(nat);
  }
  for (int s = 0; s < N && N <= 1024; ++s) cands.push_back(tb_rcm(s, adj, N));
  int best = -1, best_bw = 1 << 30;
  for (size_t i = 0; i < cands.size(); ++i) {
    int bw = tb_bandwidth(cands[i], res, conn, N, E, nullptr);
    if (bw < best_bw) {
      best_bw = bw;
      best = (int)i;
    }
  }
  std::vector<int> dofpos;
  t->bw = tb_bandwidth(cands[best], res, conn, N, E, &dofpos);

  // ---- kernel variant: smallest window that holds the band; TRUSS_LANES / TRUSS_RPL override ----
  // Preference (TRUSS_LANES / TRUSS_WLANES / TRUSS_RPL override): the narrowest window that holds the
  // band; 16 lanes per env (4 envs per wave) so that a 4096-env batch puts a wave on every SIMD.
  int want_G = tb_env_int("TRUSS_LANES", 0), want_WL = tb_env_int("TRUSS_WLANES", 0), want_RPL = tb_env_int("TRUSS_RPL", 0);
  int pick = -1;
  auto score = [&](const TbVariant &v) {
    int W = v.WL * v.RPL;
    int pref = (v.G == 16 && v.WL == 8) ? 0 : (v.G == 8 && v.WL == 8) ? 1 : (v.G == 16) ? 2 : 3;
    return W * 10000 + v.RPL * 1000 + pref * 100 + v.EPL;
  };
  for (int i = 0; i < kNumVariants; ++i) {
    const TbVariant &v = kVariants[i];
    int W = v.WL * v.RPL;
    if (W <= t->bw) continue;
    if ((E + v.G - 1) / v.G > v.EPL) continue;
    if (want_G && v.G != want_G) continue;
    if (want_WL && v.WL != want_WL) continue;
    if (want_RPL && v.RPL != want_RPL) continue;
    if (pick < 0 || score(v) < score(kVariants[pick])) pick = i;
  }
  if (pick < 0) {
    char buf[256];
    snprintf(buf, sizeof buf,
             "no compiled kernel for half-bandwidth %d with %d elements (windows: 8, 16; up to 640 elements)", t->bw, E);
    delete t;
    return tb_fail(TRUSS_EUNSUPPORTED, buf);
  }
  t->variant = pick;
  t->G = kVariants[pick].G;
  t->WL = kVariants[pick].WL;
  t->RPL = kVariants[pick].RPL;
  t->EPL = kVariants[pick].EPL;
  t->W = t->WL * t->RPL;
  const int W = t->W;
  t->n_pad = ((t->ndof + W - 1) / W) * W;
  // ---- solver geometry (see TopoDev) ----
  const int n = t->ndof;
  const int nteams = (t->G / t->WL >= 2 && t->RPL == 1 && !tb_env_int("TRUSS_ONE_SIDED", 0)) ? 2 : 1;
  int KA, mid, rowsA, rowsB, zlen, dlen, zslot;
  if (nteams == 2) {
    KA = n > W ? (n - W + 1) / 2 : 0;
    mid = n - 2 * KA;
    rowsA = KA + 2 * W;
    rowsB = KA + W;
    zlen = KA + 2 * W;
    dlen = KA + 2 * W;
    zslot = n + 2 * W;
  } else {
    KA = t->n_pad;
    mid = 0;
    rowsA = t->n_pad + W;
    rowsB = 0;
    zlen = t->n_pad + W;
    dlen = t->n_pad + W;
    zslot = t->n_pad;
  }
  // offset (in doubles) of the lower-band entry (r, c), r >= c, original solver positions.  A band row
  // stores its W entries by COLUMN RESIDUE in the owning team's frame: entry (row, col) at row*W + col%W
  // (distinct for the W consecutive columns of a row), which is the order of the solver's window registers.
  auto band_off = [&](int r, int c) -> int {
    if (nteams == 1 || r < n - KA) return r * W + (c % W);              // team A: its part + middle rows
    int rb = n - 1 - r, cb = n - 1 - c;                                    // team B frame: cb >= rb
    return (rowsA + cb) * W + (rb % W);
  };

  const int trash_off = (rowsA + rowsB) * W;  // a spare double right behind the band: sink for entries on restrained DOFs

  // solver position -> reference DOF (0-based) and node*2+comp
  t->perm.assign(t->ndof, -1);
  std::vector<int16_t> posnode(t->n_pad, -1), dofpos16(2 * N, -1), restslot(2 * N, -1);
  for (int i = 0; i < 2 * N; ++i) {
    if (dofpos[i] >= 0) {
      t->perm[dofpos[i]] = t->nsc[i] - 1;
      posnode[dofpos[i]] = (int16_t)i;
      dofpos16[i] = (int16_t)dofpos[i];
    } else {
      restslot[i] = (int16_t)(t->nsc[i] - t->ndof - 1);
    }
  }
  // band offsets of every element's four off-diagonal entries (FEM_2Dtruss.py:320-324 restricted to the
  // lower band); the node-diagonal blocks go through `diagoff` below
  std::vector<int16_t> asm_code(4 * (size_t)E, -1);
  bool band_ok = true;
  for (int e = 0; e < E; ++e) {
    int a = conn[2 * e], b = conn[2 * e + 1];
    int pa[2] = {dofpos[2 * a], dofpos[2 * a + 1]}, pb[2] = {dofpos[2 * b], dofpos[2 * b + 1]};
    int cnt = 0;
    auto add = [&](int r, int c) {
      int code = trash_off;
      if (r >= 0 && c >= 0) {
        if (r < c) std::swap(r, c);
        if (r - c >= W) band_ok = false;
        code = band_off(r, c);
        if (code > 32767) band_ok = false;
      }
      asm_code[4 * (size_t)e + cnt++] = (int16_t)code;
    };
    add(pa[0], pb[0]);
    add(pa[1], pb[1]);
    add(pa[0], pb[1]);
    add(pa[1], pb[0]);
    for (int i = 0; i < 2; ++i)   // the node's own x/y pair must fit the band too
      for (int j = 0; j < 2; ++j) {
        const int *pp = i ? pb : pa;
        if (pp[0] >= 0 && pp[1] >= 0 && std::abs(pp[0] - pp[1]) >= W) band_ok = false;
        (void)j;
      }
  }
  if (!band_ok) {
    delete t;
    return tb_fail(TRUSS_EUNSUPPORTED, "internal: band wider than window / offsets beyond int16");
  }
  // band offsets of the node-diagonal 2x2 block entries (x,x), (y,y), (x,y)
  std::vector<int16_t> diagoff(3 * (size_t)N, (int16_t)trash_off);
  for (int nd = 0; nd < N; ++nd) {
    int px = dofpos[2 * nd], py = dofpos[2 * nd + 1];
    if (px >= 0) diagoff[3 * nd + 0] = (int16_t)band_off(px, px);
    if (py >= 0) diagoff[3 * nd + 1] = (int16_t)band_off(py, py);
    if (px >= 0 && py >= 0) diagoff[3 * nd + 2] = (int16_t)band_off(std::max(px, py), std::min(px, py));
  }
  std::vector<int16_t> conn16(2 * (size_t)E);
  for (int i = 0; i < 2 * E; ++i) conn16[i] = (int16_t)conn[i];
  std::vector<uint8_t> nflags(N, 0);
  for (int n = 0; n < N; ++n) {
    uint8_t f = 0;
    if (top[n]) f |= TF_TOP;
    if (res[2 * n]) f |= TF_RESX;
    if (res[2 * n + 1]) f |= TF_RESY;
    bool lb, lr;
    if (load_mask) {
      lb = load_mask[n] != 0;
      lr = load_mask[N + n] != 0;
    } else {  // truss2D_GEN.py:421-430
      lb = !top[n] && res[2 * n + 1] == 0;
      lr = top[n] != 0;
    }
    if (lb) f |= TF_LOAD_BRIDGE;
    if (lr) f |= TF_LOAD_ROOF;
    nflags[n] = f;
  }
  std::vector<int16_t> symn(2 * (size_t)n_sym_nodes), syme(2 * (size_t)n_sym_elems);
  for (int i = 0; i < 2 * n_sym_nodes; ++i) symn[i] = (int16_t)sym_nodes[i];
  for (int i = 0; i < 2 * n_sym_elems; ++i) syme[i] = (int16_t)sym_elems[i];
  std::vector<double> area(n_sections);
  for (int i = 0; i < n_sections; ++i) area[i] = sections[2 * i];

  std::vector<double> isr(n_sections);
  for (int i = 0; i < n_sections; ++i) isr[i] = 1.0 / (area[i] * long_stress);
  std::vector<char> blob;
  size_t o_conn = tb_push(blob, conn16), o_pairs = tb_push(blob, pairs), o_nf = tb_push(blob, nflags);
  size_t o_dp = tb_push(blob, dofpos16), o_rs = tb_push(blob, restslot), o_asm = tb_push(blob, asm_code);
  size_t o_pn = tb_push(blob, posnode), o_sn = tb_push(blob, symn), o_se = tb_push(blob, syme);
  // elements incident to each node (ascending element index), padded to 8 with E = "zero slot"
  std::vector<int16_t> adj8((size_t)N * 8, (int16_t)E);
  for (int n = 0; n < N; ++n) {
    int c = 0;
    for (int e = 0; e < E; ++e)
      if (conn[2 * e] == n || conn[2 * e + 1] == n) {
        if (c >= 8) {
          delete t;
          return tb_fail(TRUSS_EUNSUPPORTED, "a node joins more than 8 elements");
        }
        adj8[(size_t)n * 8 + c++] = (int16_t)e;
      }
  }
  size_t o_ar = tb_push(blob, area), o_isr = tb_push(blob, isr);
  std::vector<float> areaf(n_sections), vsf(n_sections);
  for (int i = 0; i < n_sections; ++i) {
    areaf[i] = (float)area[i];
    vsf[i] = (float)(area[i] / area[n_sections - 1]);   // truss2D_ENV.py:86-87 (area / truss[-1] area)
  }
  size_t o_arf = tb_push(blob, areaf), o_vsf = tb_push(blob, vsf);
  // load code of every z/P slot in team frames (see TopoDev::f_zcode)
  std::vector<uint8_t> zcode((size_t)zlen * nteams, 0);
  for (int tm = 0; tm < nteams; ++tm)
    for (int pz = 0; pz < zlen; ++pz) {
      int orig = -1;
      if (nteams == 1) orig = pz < n ? pz : -1;
      else if (tm == 0) orig = pz < n - KA ? pz : -1;
      else orig = pz < KA ? n - 1 - pz : -1;
      if (orig < 0) continue;
      int nd = posnode[orig];
      if (nd < 0) continue;
      uint8_t fl = nflags[nd >> 1];
      zcode[(size_t)tm * zlen + pz] = (uint8_t)((nd & 1) | ((fl & TF_LOAD_BRIDGE) ? 2 : 0) | ((fl & TF_LOAD_ROOF) ? 4 : 0));
    }
  // slots of every element's four end displacements in the solution vector (restrained -> the zero slot)
  std::vector<int16_t> exs(4 * (size_t)E);
  for (int e = 0; e < E; ++e)
    for (int q = 0; q < 4; ++q) {
      const int dp = dofpos[2 * conn[2 * e + (q >> 1)] + (q & 1)];
      exs[4 * (size_t)e + q] = (int16_t)(dp < 0 ? zslot : dp);
    }
  size_t o_ad = tb_push(blob, adj8), o_do = tb_push(blob, diagoff), o_zc = tb_push(blob, zcode), o_exs = tb_push(blob, exs);
  blob.resize((blob.size() + 15) & ~size_t(15));
  TopoDev &D = t->dev;
  D.N = N;
  D.E = E;
  D.NP = t->NP;
  D.ndof = t->ndof;
  D.n_pad = t->n_pad;
  D.n_rest = t->n_rest;
  D.n_sym_nodes = n_sym_nodes;
  D.n_sym_elems = n_sym_elems;
  D.n_sections = n_sections;
  D.has_pairs = t->NP > 0 && 2 * t->NP == N;
  D.blob_bytes = (int32_t)blob.size();
  D.f_conn = (int32_t)o_conn;
  D.f_pairs = (int32_t)o_pairs;
  D.f_nflags = (int32_t)o_nf;
  D.f_dofpos = (int32_t)o_dp;
  D.f_restslot = (int32_t)o_rs;
  D.f_asm = (int32_t)o_asm;
  D.f_posnode = (int32_t)o_pn;
  D.f_symn = (int32_t)o_sn;
  D.f_syme = (int32_t)o_se;
  D.f_area = (int32_t)o_ar;
  D.f_isr = (int32_t)o_isr;
  D.f_areaf = (int32_t)o_arf;
  D.f_vsf = (int32_t)o_vsf;
  D.f_adj8 = (int32_t)o_ad;
  D.f_diagoff = (int32_t)o_do;
  D.f_zcode = (int32_t)o_zc;
  D.f_exs = (int32_t)o_exs;
  D.nteams = nteams;
  D.KA = KA;
  D.mid = mid;
  D.rowsA = rowsA;
  D.rowsB = rowsB;
  D.zlen = zlen;
  D.dlen = dlen;
  D.zslot = zslot;
  D.e_mod = e_mod;
  D.long_stress = long_stress;
  // LDS layout: [copy of the blob][env 0][env 1]...; every array 16-byte aligned
  size_t off = 0;
  std::vector<TbFrag> fr_early, fr_late;   // dead bytes for the bank of the fused observation writer (tb_build_emit)
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off = (off + bytes + 15) & ~size_t(15);
    return (int32_t)o;
  };
  // band region; after the back substitution it is reused as the output staging area
  {
    size_t band = sizeof(double) * ((size_t)(rowsA + rowsB) * W + 2);  // + trash slot
    size_t so = 0;
    auto sub = [&](size_t bytes) {
      size_t o = so;
      so = (so + bytes + 15) & ~size_t(15);
      return (int32_t)o;
    };
    D.so_q0 = sub(sizeof(float) * E);
    D.so_sr = sub(sizeof(float) * E);
    D.so_disp = sub(sizeof(float) * 2 * N);
    D.so_mu = sub(sizeof(float) * N);
    D.so_md = sub(sizeof(float) * N);
    D.so_comp = sub((size_t)E);
    D.o_red = -1;  // set below: objective partials live behind the staging rows (the band is dead by then)
    size_t red_off = so;
    so += sizeof(double) * TRUSS_NRED * (size_t)t->G;
    D.o_kb = carve(std::max(band, so));
    D.o_red = D.o_kb + (int32_t)red_off;
    fr_early.push_back({(size_t)D.o_kb + so, (size_t)D.o_kb + std::max(band, so)});
  }
  // solver scratch; before the solver the same bytes hold the per-element (k cc, k cs, k ss), after
  // the back substitution `red` (objective partials) reuses the z vectors
  {
    size_t o0 = off;
    D.o_zs = carve(sizeof(double) * (size_t)zlen * nteams);  // z vector per team (zlen is even)
    D.o_xsol = carve(sizeof(double) * (zslot + 4));  // + zero slot, dummy slot, 16-byte fill
    D.o_rbuf = carve(sizeof(double) * std::max(t->n_rest, 1));
    D.o_zring = carve(sizeof(double) * W * nteams);   // per-lane trash slots
    D.o_ev = (int32_t)o0;
    size_t need = sizeof(double) * 3 * ((size_t)E + 1);
    if (off - o0 < need) off = (o0 + need + 15) & ~size_t(15);
    fr_late.push_back({o0, (size_t)D.o_rbuf});                     // z vectors, solution vector
    fr_late.push_back({(size_t)D.o_zring, off});                    // behind the reactions
  }
  D.o_par = carve(sizeof(double) * 8);
  D.o_y = carve(sizeof(float) * N);
  D.o_x = carve(sizeof(float) * N);
  D.o_tg = carve(sizeof(float) * N);
  {
    size_t o0 = off;
    D.o_geo = carve(sizeof(float) * 2 * N);
    D.o_tac = carve(sizeof(float) * 3 * N);
    D.o_mrg = (int32_t)o0;  // merge scratch of the two-sided solver: the actions are dead by then
    size_t need = nteams == 2 ? sizeof(double) * (size_t)(W * W + W) : 0;
    if (off - o0 < need) off = (o0 + need + 15) & ~size_t(15);
    fr_early.push_back({o0, off});
  }
  D.o_sec = carve(sizeof(int32_t) * E);
  // stagger env regions across LDS banks: stride = 64 B (mod 256 B)
  size_t stride = (off + 255) & ~size_t(255);
  stride += 64;
  D.env_stride = (int32_t)stride;
  D.o_env0 = (int32_t)((blob.size() + 255) & ~size_t(255));
  t->lds_bytes = D.o_env0 + stride * (64 / t->G);
  tb_build_emit(t, conn, res, top, fr_early, fr_late, blob);   // may append the LDS-resident nN_x_e table to the blob
  t->blob = tb_dev_alloc(blob.size());
  if (!t->blob || !tb_dev_upload(t->blob, blob.data(), blob.size())) {
    if (t->blob) tb_dev_free(t->blob);
    if (t->etab) tb_dev_free(t->etab);
    delete t;
    return tb_fail(TRUSS_ENOMEM, "device allocation/upload of topology tables failed");
  }
  D.blob = (const char *)t->blob;
  t->dev_emit = D;
  t->dev_emit.blob_bytes = (int32_t)blob.size();
  t->dev_emit.o_flag = (int32_t)blob.size();   // progress word right behind the staged tables
  t->dev_emit.o_env0 = (int32_t)((blob.size() + 16 + 255) & ~size_t(255));
  t->lds_bytes_emit = t->dev_emit.o_env0 + stride * (64 / t->G);
  if (D.emit_ok && t->lds_bytes_emit > 160 * 1024) D.emit_ok = t->dev_emit.emit_ok = 0;
  if (tb_env_int("TRUSS_VERBOSE", 0))
    fprintf(stderr,
            "[truss_mi355] N=%d E=%d ndof=%d bw=%d | G=%d WL=%d RPL=%d EPL=%d teams=%d KA=%d mid=%d | tables %d B, "
            "env %zu B, LDS/workgroup %zu B (%zu workgroups/CU), fused observation writer %s\n",
            N, E, t->ndof, t->bw, t->G, t->WL, t->RPL, t->EPL, nteams, KA, mid, D.blob_bytes, stride, t->lds_bytes,
            (size_t)(160 * 1024) / t->lds_bytes, D.emit_ok ? "yes" : "no");
  if (tb_env_int("TRUSS_VERBOSE", 0) && D.emit_ok)
    fprintf(stderr, "[truss_mi355]   fused launch: tables %d B, LDS/workgroup %zu B (%zu workgroups/CU)\n", t->dev_emit.blob_bytes,
            t->lds_bytes_emit, (size_t)(160 * 1024) / t->lds_bytes_emit);
  if (t->lds_bytes > 160 * 1024) {
    tb_dev_free(t->blob);
    if (t->etab) tb_dev_free(t->etab);
    delete t;
    return tb_fail(TRUSS_EUNSUPPORTED, "topology needs more than 160 KiB of LDS per workgroup");
  }
  *out = t;
  return TRUSS_OK;
}

extern "C" int truss_topo_destroy(truss_topo_t *t) {
  if (!t) return TRUSS_OK;
  if (t->blob) tb_dev_free(t->blob);
  if (t->etab) tb_dev_free(t->etab);
  delete t;
  return TRUSS_OK;
}

extern "C" int truss_topo_dofs(const truss_topo_t *t, int32_t *nsc, int32_t *ttnsc) {
  if (!t) return tb_fail(TRUSS_EINVAL, "NULL topology");
  if (nsc) memcpy(nsc, t->nsc.data(), t->nsc.size() * sizeof(int32_t));
  if (ttnsc) memcpy(ttnsc, t->ttnsc.data(), t->ttnsc.size() * sizeof(int32_t));
  return t->ndof;
}

extern "C" int truss_topo_solver_info(const truss_topo_t *t, int32_t *perm, int32_t *half_bandwidth,
                                      int32_t *lanes_per_env, int32_t *rows_per_lane) {
  if (!t) return tb_fail(TRUSS_EINVAL, "NULL topology");
  if (perm) memcpy(perm, t->perm.data(), t->perm.size() * sizeof(int32_t));
  if (half_bandwidth) *half_bandwidth = t->bw;
  if (lanes_per_env) *lanes_per_env = t->G;
  if (rows_per_lane) *rows_per_lane = t->RPL;
  return TRUSS_OK;
}

extern "C" int truss_topo_fused_obs(const truss_topo_t *t) { return t && t->dev.emit_ok ? 1 : 0; }

static int tb_make_step_args(const truss_topo_t *t, const truss_step_args_t *a, StepArgsDev &D) {
  if (!t || !a) return tb_fail(TRUSS_EINVAL, "NULL argument");
  if (a->struct_size != sizeof(truss_step_args_t)) return tb_fail(TRUSS_EINVAL, "truss_step_args_t size mismatch (ABI)");
  if (a->n_envs < 1) return tb_fail(TRUSS_EINVAL, "n_envs < 1");
  const bool decode = !(a->flags & TRUSS_F_NO_DECODE);
  if (!a->x || !a->y_in || !a->sec_in || !a->target || !a->env_params || !a->y_out || !a->disp || !a->q0 || !a->sr ||
      !a->comp || !a->point)
    return tb_fail(TRUSS_EINVAL, "a required device pointer is NULL");
  if (((size_t)a->point & 15) != 0 || ((size_t)a->obj & 7) != 0) return tb_fail(TRUSS_EINVAL, "point must be 16-byte aligned, obj 8-byte aligned");
  if (decode && (!a->a_geo || !a->a_topo)) return tb_fail(TRUSS_EINVAL, "actions are NULL");
  if (decode && !t->dev.has_pairs) return tb_fail(TRUSS_EINVAL, "action decode needs a vertical-pair table");
  if ((a->max_up_in == nullptr) != (a->max_down_in == nullptr)) return tb_fail(TRUSS_EINVAL, "max_up_in/max_down_in must come together");
  if ((a->max_up_out == nullptr) != (a->max_down_out == nullptr)) return tb_fail(TRUSS_EINVAL, "max_up_out/max_down_out must come together");
  if (decode && t->dev.n_sym_nodes + t->dev.n_sym_elems > 0 && !a->coin) return tb_fail(TRUSS_EINVAL, "symmetric topology needs coin[]");
  D.B = a->n_envs;
  D.flags = a->flags;
  D.x = a->x;
  D.y_in = a->y_in;
  D.sec_in = a->sec_in;
  D.mu_in = a->max_up_in;
  D.md_in = a->max_down_in;
  D.a_geo = a->a_geo;
  D.a_topo = a->a_topo;
  D.coin = a->coin;
  D.target = a->target;
  D.env_params = a->env_params;
  D.y_out = a->y_out;
  D.sec_out = a->sec_out;
  D.mu_out = a->max_up_out;
  D.md_out = a->max_down_out;
  D.disp = a->disp;
  D.q0 = a->q0;
  D.sr = a->sr;
  D.comp = a->comp;
  D.point = a->point;
  D.obj = a->obj;
  D.disp64 = a->disp_f64;
  D.q064 = a->q0_f64;
  D.energy = a->energy;
  D.react = a->reactions;
  D.status = a->status;
  D.x_n = a->x_n;
  D.A_s = a->A_s;
  D.A_ts = a->A_n_ts;
  D.A_cs = a->A_n_cs;
  D.nxn = a->nN_x_n;
  D.nxe = a->nN_x_e;
  if (a->flags & TRUSS_F_EMIT_OBS) {
    if (!a->sec_out || !a->max_up_out) return tb_fail(TRUSS_EINVAL, "TRUSS_F_EMIT_OBS needs sec_out and max_up_out / max_down_out");
    if (!t->dev.emit_ok && tb_obs_lds_bytes(t->N) > 160 * 1024) return tb_fail(TRUSS_EUNSUPPORTED, "N too large for the observation kernel");
  }
  return TRUSS_OK;
}

// One step; with TRUSS_F_EMIT_OBS the observation tensors come from the same launch where the topology allows
// it (dev.emit_ok) and from the observation kernel, launched right behind the step on the same stream, elsewhere.
static int tb_step_dispatch(const truss_topo *t, const StepArgsDev &D, void *stream) {
  if (!(D.flags & TB_EMIT_OBS)) return tb_launch_step(t, D, false, stream);
  if (t->dev.emit_ok) return tb_launch_step(t, D, true, stream);
  int rc = tb_launch_step(t, D, false, stream);
  if (rc != TRUSS_OK) return rc;
  ObsArgsDev O;
  O.B = D.B;
  O.flags = 0;
  O.x = D.x;
  O.y = D.y_out;
  O.sec = D.sec_out;
  O.mu = D.mu_out;
  O.md = D.md_out;
  O.target = D.target;
  O.disp = D.disp;
  O.q0 = D.q0;
  O.sr = D.sr;
  O.comp = D.comp;
  O.env_params = D.env_params;
  O.x_n = D.x_n;
  O.A_s = D.A_s;
  O.A_ts = D.A_ts;
  O.A_cs = D.A_cs;
  O.nxn = D.nxn;
  O.nxe = D.nxe;
  O.tile_rows = tb_obs_tile_rows(t->N);
  O.n_split = tb_obs_splits(t->N) ? (t->N + O.tile_rows - 1) / O.tile_rows : 1;
  return tb_launch_obs(t, O, stream);
}

// the persistent rollout kernel keeps rows in LDS with the staging fast path's 16-byte accesses: same conditions
static bool tb_rollout_one_launch(const truss_topo *t, const StepArgsDev &D) {
  if (tb_env_int("TRUSS_ROLLOUT_LAUNCHES", 0)) return false;          // diagnostic: one launch per step, as in round 1
  if ((D.flags & (TB_NO_DECODE | TB_EMIT_OBS | TB_CLAMP_INPLACE)) || !D.a_geo) return false;
  if (!tb_variant_rolls(t->G, t->WL, t->RPL, t->EPL)) return false;
  const int NPL = (2 * t->EPL + 4) / 5, ncap = std::max(t->G * NPL, 64), ecap = std::max(t->G * t->EPL, 128);
  return (t->N & 3) == 0 && t->N <= ncap && t->E <= ecap && t->dev.blob_bytes <= 20 * 64 * 16;
}

extern "C" int truss_topo_persistent_rollout(const truss_topo_t *t) {
  if (!t) return 0;
  StepArgsDev D{};
  D.a_geo = (float *)1;   // "decode steps with actions": what the query is about
  return tb_rollout_one_launch(t, D) ? 1 : 0;
}

extern "C" int truss_step(const truss_topo_t *t, const truss_step_args_t *a, void *stream) {
  StepArgsDev D;
  int rc = tb_make_step_args(t, a, D);
  if (rc != TRUSS_OK) return rc;
  return tb_step_dispatch(t, D, stream);
}

extern "C" int truss_rollout(const truss_topo_t *t, const truss_step_args_t *a, int32_t n_steps, int32_t n_action_sets,
                             void *stream) {
  StepArgsDev D;
  int rc = tb_make_step_args(t, a, D);
  if (rc != TRUSS_OK) return rc;
  if (n_steps < 1 || n_action_sets < 1) return tb_fail(TRUSS_EINVAL, "n_steps/n_action_sets < 1");
  if (!a->sec_out) return tb_fail(TRUSS_EINVAL, "rollout needs sec_out");
  if (a->max_up_in) return tb_fail(TRUSS_EINVAL, "rollout recomputes move ranges; pass max_up_in = NULL");
  if (tb_rollout_one_launch(t, D)) return tb_launch_rollout(t, D, n_steps, n_action_sets, stream);
  const float *ybuf[2] = {a->y_in, a->y_out};
  const int32_t *sbuf[2] = {a->sec_in, a->sec_out};
  const size_t gstride = (size_t)a->n_envs * t->N * 2, tstride = (size_t)a->n_envs * t->N * 3;
  for (int s = 0; s < n_steps; ++s) {
    StepArgsDev S = D;
    S.y_in = ybuf[s & 1];
    S.y_out = (float *)ybuf[(s + 1) & 1];
    S.sec_in = sbuf[s & 1];
    S.sec_out = (int32_t *)sbuf[(s + 1) & 1];
    if (D.a_geo) {
      S.a_geo = D.a_geo + (size_t)(s % n_action_sets) * gstride;
      S.a_topo = D.a_topo + (size_t)(s % n_action_sets) * tstride;
    }
    rc = tb_step_dispatch(t, S, stream);
    if (rc != TRUSS_OK) return rc;
  }
  return TRUSS_OK;
}

extern "C" int truss_obs(const truss_topo_t *t, const truss_obs_args_t *a, void *stream) {
  if (!t || !a) return tb_fail(TRUSS_EINVAL, "NULL argument");
  if (a->struct_size != sizeof(truss_obs_args_t)) return tb_fail(TRUSS_EINVAL, "truss_obs_args_t size mismatch (ABI)");
  if (a->n_envs < 1) return tb_fail(TRUSS_EINVAL, "n_envs < 1");
  if (!a->x || !a->y || !a->sec || !a->max_up || !a->max_down || !a->target || !a->disp || !a->q0 || !a->sr || !a->comp ||
      !a->env_params)
    return tb_fail(TRUSS_EINVAL, "a required device pointer is NULL");
  if (tb_obs_lds_bytes(t->N) > 160 * 1024) return tb_fail(TRUSS_EUNSUPPORTED, "N too large for the observation kernel");
  // (with row tiles that is N > ~3000)
  ObsArgsDev D;
  D.B = a->n_envs;
  D.flags = a->flags;
  D.x = a->x;
  D.y = a->y;
  D.sec = a->sec;
  D.mu = a->max_up;
  D.md = a->max_down;
  D.target = a->target;
  D.disp = a->disp;
  D.q0 = a->q0;
  D.sr = a->sr;
  D.comp = a->comp;
  D.env_params = a->env_params;
  D.x_n = a->x_n;
  D.A_s = a->A_s;
  D.A_ts = a->A_n_ts;
  D.A_cs = a->A_n_cs;
  D.nxn = a->nN_x_n;
  D.nxe = a->nN_x_e;
  D.tile_rows = tb_obs_tile_rows(t->N);
  D.n_split = tb_obs_splits(t->N) ? (t->N + D.tile_rows - 1) / D.tile_rows : 1;
  return tb_launch_obs(t, D, stream);
}
