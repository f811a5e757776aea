// truss_emu.cpp -- CPU lane emulator of the HIP step kernel.  *** TEST INFRASTRUCTURE ONLY ***
//
// Builds the same C ABI (include/truss_mi355.h) from the same lane program
// (mop-truss-marl_amd/csrc/truss_body.h) and the same host code (truss_host.h), but runs every
// phase for the 64 lanes of a workgroup one after the other on the CPU.  It exists so that the
// kernel's indexing (band offsets, window rotation, pair/symmetry tables) can be debugged in a
// container without a GPU.  The product (mop-truss-marl_amd/truss_mi355) never loads this library;
// truss_backend() returns "emu" so tests can tell the two apart.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

struct float4 {
  float x, y, z, w;
};
#define TRUSS_HD inline
#define TRUSS_UNROLL
#define TB_STREAM_STORE(p, v) (*(p) = (v))
#define TB_OBS_STORE(p, v) (*(p) = (v))
static inline void tb_lds_add(double *p, double v) { *p += v; }
static inline double tb_rcp(double d) { return 1.0 / d; }
// The emulator's lanes know their wave (cross-lane stand-ins below); the product's lane struct does not.
template <class LN>
struct EmuLane : LN {
  EmuLane *peers = nullptr;   // the 64 lanes of the workgroup
};
template <class LN>
static inline EmuLane<LN> *emu_peers(LN &ln) { return static_cast<EmuLane<LN> &>(ln).peers; }
// DPP row broadcast stand-in: the value lane `src` of this lane's team computed in the previous phase
template <class LN>
static inline double tb_team_bcast(LN &ln, int src) { return emu_peers(ln)[ln.lane - ln.gs + src].bx; }
static inline double tb_rsqrt(double x) { return 1.0 / sqrt(x); }
static inline float tb_rcpf(float d) { return 1.0f / d; }
// min / max over the lanes of an env: the partials the lanes left in the previous phase
template <class LN>
static inline float tb_group_min(LN &ln, int c) {
  float v = INFINITY;
  for (int j = 0; j < LN::G_; ++j) v = fminf(v, emu_peers(ln)[ln.lane - ln.g + j].pmn[c]);
  return v;
}
template <class LN>
static inline float tb_group_max(LN &ln, int c) {
  float v = -INFINITY;
  for (int j = 0; j < LN::G_; ++j) v = fmaxf(v, emu_peers(ln)[ln.lane - ln.g + j].pmx[c]);
  return v;
}

// objective partials folded over the lanes of an env (HIP: DPP butterfly; the float64 sums of float32 terms are exact)
template <class LN>
static inline double tb_group_sum_d(LN &ln, int which) {
  double v = 0.0;
  for (int j = 0; j < LN::G_; ++j) {
    const LN &p = emu_peers(ln)[ln.lane - ln.g + j];
    v += which == 0 ? p.p_vol : which == 1 ? p.p_dt : p.p_en;
  }
  return v;
}
template <class LN>
static inline float tb_group_max_f(LN &ln, int which) {
  float v = 0.0f;
  for (int j = 0; j < LN::G_; ++j) {
    const LN &p = emu_peers(ln)[ln.lane - ln.g + j];
    v = fmaxf(v, which == 0 ? p.p_c1 : which == 1 ? p.p_c2 : (float)p.bad);
  }
  return v;
}

// TRUSS_EMU_FORCE_OBS_TIMEOUT=1 (tests of the status plumbing): segments 2 and 3 of the observation stream behave as if the
// streaming wave had given up waiting -- nothing is written, status gets TRUSS_STATUS_OBS_TIMEOUT
static int g_emu_force_obs_timeout = 0;
template <class LN, class TD>
static inline bool tb_obs_timed_out(LN &, const TD &) { return LN::EMIT_ && g_emu_force_obs_timeout != 0; }

#include "../../mop-truss-marl_amd/csrc/truss_body.h"

#define TRUSS_BACKEND_NAME "emu"
struct truss_topo;
static void *tb_dev_alloc(size_t n) { return malloc(n); }
static void tb_dev_free(void *p) { free(p); }
static bool tb_dev_upload(void *dst, const void *src, size_t n) {
  memcpy(dst, src, n);
  return true;
}
static int tb_launch_step(const truss_topo *t, const StepArgsDev &A, bool emit, void *stream);
static int tb_launch_obs(const truss_topo *t, const ObsArgsDev &A, void *stream);
static int tb_launch_rollout(const truss_topo *t, const StepArgsDev &A, int n_steps, int n_sets, void *stream);

#include "../../mop-truss-marl_amd/csrc/truss_host.h"

template <int G, int WL, int RPL, int EPL, bool EMIT>
static void emu_run(const truss_topo *t, const StepArgsDev &A) {
  using Lane = EmuLane<StepLane<G, WL, RPL, EPL, EMIT>>;
  constexpr int W_ = Lane::W;
  constexpr bool EMIT_ = EMIT;
  const TopoDev &T = EMIT ? t->dev_emit : t->dev;
  const int nblocks = (A.B + Lane::EPB - 1) / Lane::EPB;
  std::vector<char> lds(EMIT ? t->lds_bytes_emit : t->lds_bytes);
  std::vector<Lane> lanes(64);
  for (int b = 0; b < nblocks; ++b) {
    memset(lds.data(), 0xA5, lds.size());  // poison: catches reads of uninitialised LDS
    for (int l = 0; l < 64; ++l) {
      lanes[l].init(l, b, T, A, lds.data());
      lanes[l].peers = lanes.data();
      lanes[l].emit_tables_load(T);   // (HIP: the streaming wave of an EMIT workgroup)
    }
#define PH(call) \
  for (auto &ln : lanes) ln.call
#define PH_NS(call) \
  for (auto &ln : lanes) ln.call
#define BAR() (void)0
#define EMIT_POINT(k) if (!(g_emu_force_obs_timeout && (k) >= 2)) { TRUSS_STREAM_SEG##k(PH, T, A) }   /* HIP: the streaming wave's work; here: in place */
    g_emu_force_obs_timeout = getenv("TRUSS_EMU_FORCE_OBS_TIMEOUT") && atoi(getenv("TRUSS_EMU_FORCE_OBS_TIMEOUT"));
    TRUSS_STEP_SCHEDULE(PH, PH_NS, BAR, T, A)
#undef PH
#undef PH_NS
#undef BAR
#undef EMIT_POINT
  }
}

// the persistent rollout: all steps of a workgroup back to back on ONE LDS image (no re-poisoning between steps: what a
// later step reads without re-staging must really have been left there by the previous one)
template <int G, int WL, int RPL, int EPL>
static void emu_rollout(const truss_topo *t, StepArgsDev A, int n_steps, int n_sets) {
  using Lane = EmuLane<StepLane<G, WL, RPL, EPL, false>>;
  constexpr int W_ = Lane::W;
  constexpr bool EMIT_ = false;
  const TopoDev &T = t->dev;
  const int nblocks = (A.B + Lane::EPB - 1) / Lane::EPB;
  const size_t gstride = (size_t)A.B * t->N * 2, tstride = (size_t)A.B * t->N * 3;
  float *ybuf[2] = {(float *)A.y_in, A.y_out};
  int32_t *sbuf[2] = {(int32_t *)A.sec_in, A.sec_out};
  float *g0 = A.a_geo, *t0 = A.a_topo;
  std::vector<char> lds(t->lds_bytes);
  std::vector<Lane> lanes(64);
  for (int b = 0; b < nblocks; ++b) {
    memset(lds.data(), 0xA5, lds.size());
    for (int s = 0; s < n_steps; ++s) {
      const int nset = (s + 1) % n_sets;
      for (int l = 0; l < 64; ++l) {
        lanes[l].init(l, b, T, A, lds.data());
        lanes[l].peers = lanes.data();
        lanes[l].rs_first_step = s;
        lanes[l].rs_y_out = ybuf[(s + 1) & 1];
        lanes[l].rs_sec_out = sbuf[(s + 1) & 1];
        lanes[l].rs_next_geo = s + 1 < n_steps ? g0 + (size_t)nset * gstride : nullptr;
        lanes[l].rs_next_topo = s + 1 < n_steps ? t0 + (size_t)nset * tstride : nullptr;
      }
#define PH(call) \
  for (auto &ln : lanes) ln.call
#define PH_NS(call) \
  for (auto &ln : lanes) ln.call
#define BAR() (void)0
#define EMIT_POINT(k) (void)0
      TRUSS_STEP_SCHEDULE(PH, PH_NS, BAR, T, A)
#undef PH
#undef PH_NS
#undef BAR
#undef EMIT_POINT
    }
  }
}
static int tb_launch_rollout(const truss_topo *t, const StepArgsDev &A, int n_steps, int n_sets, void *) {
  const TbVariant &v = kVariants[t->variant];
#define X(g, wl, r, e) \
  if (v.G == g && v.WL == wl && v.RPL == r && v.EPL == e) { emu_rollout<g, wl, r, e>(t, A, n_steps, n_sets); return TRUSS_OK; }
  TRUSS_ROLLOUT_VARIANTS(X)
#undef X
  return tb_fail(TRUSS_EUNSUPPORTED, "variant not compiled with the persistent rollout");
}

static int tb_launch_step(const truss_topo *t, const StepArgsDev &A, bool emit, void *) {
  const TbVariant &v = kVariants[t->variant];
  if (emit) {
#define X(g, wl, r, e) \
  if (v.G == g && v.WL == wl && v.RPL == r && v.EPL == e) { emu_run<g, wl, r, e, true>(t, A); return TRUSS_OK; }
    TRUSS_EMIT_VARIANTS(X)
#undef X
    return tb_fail(TRUSS_EUNSUPPORTED, "variant not compiled with the observation writer");
  }
#define X(g, wl, r, e) \
  if (v.G == g && v.WL == wl && v.RPL == r && v.EPL == e) { emu_run<g, wl, r, e, false>(t, A); return TRUSS_OK; }
  TRUSS_VARIANTS(X)
#undef X
  return tb_fail(TRUSS_EUNSUPPORTED, "variant not compiled into the emulator");
}

static int tb_launch_obs(const truss_topo *t, const ObsArgsDev &A, void *) {
  const TopoDev &T = t->dev;
  std::vector<char> lds(tb_obs_lds_bytes(t->N));
  std::vector<ObsLane> lanes(64);
  for (int b = 0; b < A.B; ++b)
    for (int tile = 0; tile < (A.n_split > 1 ? A.n_split + 1 : 1); ++tile) {       // grid (B, tiles + the rows' workgroup)
      memset(lds.data(), 0xA5, lds.size());
      for (int l = 0; l < 64; ++l) lanes[l].init(l, b, tile, T, A, lds.data());
      const int r0_tile = lanes[0].role == 1 ? (tile - 1) * A.tile_rows : 0;
#define PH(call) \
  for (auto &ln : lanes) ln.call
      TRUSS_OBS_SCHEDULE(PH, PH, T, A, r0_tile)
#undef PH
    }
  return TRUSS_OK;
}

// ---- truss_front: serial restatement for the CPU test backend (the HIP kernel is in truss_hip.hip) ----
#include "../../mop-truss-marl_amd/csrc/truss_front.h"
#include <cmath>
#include <vector>
static double tb_hv_sorted(const std::vector<double> &cx, const std::vector<double> &cy, double minx, double miny, double rx,
                           double ry) {
  double area = 0.0, runmin = 1.0;
  const int n = (int)cx.size();
  for (int k = 0; k < n; ++k) {
    runmin = cy[k] < runmin ? cy[k] : runmin;
    const double nx = k + 1 < n ? cx[k + 1] : 1.0;
    area += (nx - cx[k]) * (1.0 - runmin);
  }
  return area - ((1.0 - rx) * (1.0 - minx) + (1.0 - ry) * (1.0 - miny) - (1.0 - rx) * (1.0 - ry));
}
extern "C" int truss_front(const truss_front_args_t *a, void *) {
  if (int rc = tb_front_check(a)) return rc;
  const int P = a->max_points;
  for (int b = 0; b < a->n_envs; ++b) {
    int n = a->n_points[b];
    n = n < 0 ? 0 : (n > P ? P : n);
    const double *pt = a->points + (size_t)b * P * 4;
    std::vector<int> fr;
    for (int i = 0; i < n; ++i) {
      const double *r = pt + 4 * i;
      if (r[2] > 1.0 || r[3] > 1.0) continue;
      bool out = false;
      for (int j = 0; j < n && !out; ++j) {
        const double *q = pt + 4 * j;
        if (q[2] > 1.0 || q[3] > 1.0) continue;
        if (q[0] < r[0] && q[1] < r[1]) out = true;
        if (j < i && q[0] == r[0] && q[1] == r[1] && q[2] == r[2] && q[3] == r[3]) out = true;
      }
      if (!out) fr.push_back(i);
    }
    std::stable_sort(fr.begin(), fr.end(), [&](int u, int v) {
      const double *p = pt + 4 * u, *q = pt + 4 * v;
      return p[0] < q[0] || (p[0] == q[0] && p[1] < q[1]);
    });
    auto dist = [&](int u, int v) {
      const double dx = pt[4 * u] - pt[4 * v], dy = pt[4 * u + 1] - pt[4 * v + 1];
      return std::sqrt(dx * dx + dy * dy);
    };
    int nf = (int)fr.size();
    if ((a->flags & TRUSS_FRONT_TRUNCATE) && nf > a->max_front) {
      std::vector<double> cr(nf, 0.0);
      for (int k = 0; k < nf; ++k)
        cr[k] = k == 0 ? dist(fr[0], fr[1]) : (k == nf - 1 ? dist(fr[nf - 2], fr[nf - 1]) : dist(fr[k - 1], fr[k]) + dist(fr[k], fr[k + 1]));
      std::vector<int> mid;
      for (int k = 1; k < nf - 1; ++k) mid.push_back(k);
      std::stable_sort(mid.begin(), mid.end(), [&](int u, int v) { return cr[u] > cr[v]; });
      std::vector<char> kp(nf, 0);
      kp[0] = kp[nf - 1] = 1;
      for (int k = 0; k < a->max_front - 2; ++k) kp[mid[k]] = 1;
      std::vector<int> f2;
      for (int k = 0; k < nf; ++k)
        if (kp[k]) f2.push_back(fr[k]);
      fr.swap(f2);
      nf = (int)fr.size();
    }
    if (a->front_idx)
      for (int k = 0; k < P; ++k) a->front_idx[(size_t)b * P + k] = k < nf ? fr[k] : -1;
    if (a->n_front) a->n_front[b] = nf;
    const double rx = a->ref_points ? a->ref_points[2 * b] : 1.0, ry = a->ref_points ? a->ref_points[2 * b + 1] : 1.0;
    if (a->metrics) {
      double maxd = 0.0, disd = 1.0, sumd = 0.0, stdcd = 1.0, pn = 0.0;
      std::vector<double> d;
      for (int k = 0; k + 1 < nf; ++k) d.push_back(dist(fr[k], fr[k + 1]));
      if (nf >= 2) {
        for (double v : d) { maxd = v > maxd ? v : maxd; sumd += v; }
        double acc = 0.0;
        for (double v : d) acc += (v - maxd / d.size()) * (v - maxd / d.size());
        disd = std::sqrt(acc / d.size());
      }
      if (nf > 3) {
        std::vector<double> cd;
        double s = 0.0, mx = 0.0;
        for (int k = 1; k < nf - 1; ++k) {
          const double v = std::fabs(pt[4 * fr[k - 1]] - pt[4 * fr[k + 1]]) + std::fabs(pt[4 * fr[k - 1] + 1] - pt[4 * fr[k + 1] + 1]);
          cd.push_back(v); s += v; mx = v > mx ? v : mx;
        }
        if (s != 0.0) {
          double mean = 0.0, var = 0.0, p10 = 0.0;
          for (double &v : cd) { v /= mx; mean += v; }
          mean /= cd.size();
          for (double v : cd) { var += (v - mean) * (v - mean); p10 += std::pow(v, 10.0); }
          stdcd = std::sqrt(var / cd.size());
          pn = std::pow(p10, 0.1);
        }
      }
      double *M = a->metrics + (size_t)b * 5;
      M[0] = maxd; M[1] = disd; M[2] = pn; M[3] = sumd; M[4] = stdcd;
    }
    if (a->hv_front) {
      double hv = 0.0;
      if (nf > 0 && !(nf == 1 && pt[4 * fr[0]] == 1.0 && pt[4 * fr[0] + 1] == 1.0)) {
        std::vector<double> cx, cy;
        double minx = pt[4 * fr[0]], miny = pt[4 * fr[0] + 1];
        for (int k : fr) {
          cx.push_back(std::fmin(pt[4 * k], 1.0)); cy.push_back(std::fmin(pt[4 * k + 1], 1.0));
          minx = std::fmin(minx, pt[4 * k]); miny = std::fmin(miny, pt[4 * k + 1]);
        }
        hv = tb_hv_sorted(cx, cy, minx, miny, rx, ry);
      }
      a->hv_front[b] = hv;
    }
    if (a->hv_all) {
      double hv = 0.0;
      if (n > 0 && !(n == 1 && pt[0] == 1.0 && pt[1] == 1.0)) {
        std::vector<int> o(n);
        for (int k = 0; k < n; ++k) o[k] = k;
        std::stable_sort(o.begin(), o.end(), [&](int u, int v) { return std::fmin(pt[4 * u], 1.0) < std::fmin(pt[4 * v], 1.0); });
        std::vector<double> cx, cy;
        double minx = pt[0], miny = pt[1];
        for (int k : o) {
          cx.push_back(std::fmin(pt[4 * k], 1.0)); cy.push_back(std::fmin(pt[4 * k + 1], 1.0));
          minx = std::fmin(minx, pt[4 * k]); miny = std::fmin(miny, pt[4 * k + 1]);
        }
        hv = tb_hv_sorted(cx, cy, minx, miny, rx, ry);
      }
      a->hv_all[b] = hv;
    }
  }
  return TRUSS_OK;
}

extern "C" int truss_gcn_aggregate_sparse(const float *adj, int64_t a_batch_stride, const int16_t *nbr, int32_t k_nbr, const float *h,
                                          const float *bias, float *out, int32_t n_batch, int32_t n_nodes, int32_t n_channels,
                                          int32_t act, void *) {
  if (!adj || !nbr || !h || !out) return tb_fail(TRUSS_EINVAL, "truss_gcn_aggregate_sparse: NULL argument");
  if (n_batch < 0 || n_nodes < 1 || n_nodes > 32767 || k_nbr < 1 || k_nbr > 16 || n_channels < 4 || (n_channels & 3) || act < 0 || act > 2)
    return tb_fail(TRUSS_EINVAL, "truss_gcn_aggregate_sparse: n_nodes 1..32767, k_nbr 1..16, n_channels a multiple of 4, act 0..2");
  if ((((size_t)h | (size_t)out | (size_t)bias) & 15) != 0 || h == out)
    return tb_fail(TRUSS_EINVAL, "truss_gcn_aggregate_sparse: h / out / bias must be 16-byte aligned, out must not alias h");
  const int N = n_nodes, C = n_channels;
  for (int b = 0; b < n_batch; ++b) {
    const float *A = adj + (size_t)b * a_batch_stride, *H = h + (size_t)b * N * C;
    for (int i = 0; i < N; ++i)
      for (int c = 0; c < C; ++c) {
        float acc = bias ? bias[c] : 0.0f;
        for (int k = 0; k < k_nbr; ++k) {
          const int j = nbr[i * k_nbr + k];
          if (j >= 0) acc = std::fmaf(A[i * N + j], H[(size_t)j * C + c], acc);
        }
        if (act == 1) acc = acc > 0.0f ? acc : 0.0f;
        else if (act == 2) acc = 1.0f / (1.0f + std::exp(-acc));
        out[((size_t)b * N + i) * C + c] = acc;
      }
  }
  return TRUSS_OK;
}

extern "C" int truss_gcn_aggregate(const float *adj, int64_t a_batch_stride, const float *h, const float *bias, float *out,
                                   int32_t n_batch, int32_t n_nodes, int32_t n_channels, int32_t act, void *) {
  if (!adj || !h || !out) return tb_fail(TRUSS_EINVAL, "truss_gcn_aggregate: NULL argument");
  if (n_batch < 0 || n_nodes < 1 || n_nodes > 64 || n_channels < 1 || act < 0 || act > 2)
    return tb_fail(TRUSS_EINVAL, "truss_gcn_aggregate: n_nodes must be 1..64, act 0..2");
  const int N = n_nodes, C = n_channels;
  std::vector<float> tmp((size_t)N * C);
  for (int b = 0; b < n_batch; ++b) {
    const float *A = adj + (size_t)b * a_batch_stride, *H = h + (size_t)b * N * C;
    for (int i = 0; i < N; ++i)
      for (int c = 0; c < C; ++c) {
        float acc = bias ? bias[c] : 0.0f;
        for (int j = 0; j < N; ++j) acc = std::fmaf(A[i * N + j], H[(size_t)j * C + c], acc);
        if (act == 1) acc = acc > 0.0f ? acc : 0.0f;
        else if (act == 2) acc = 1.0f / (1.0f + std::exp(-acc));
        tmp[(size_t)i * C + c] = acc;
      }
    std::copy(tmp.begin(), tmp.end(), out + (size_t)b * N * C);
  }
  return TRUSS_OK;
}

// truss_gcn_layer, CPU stand-in of the MFMA kernel (plain loops, the same operation order: aggregate the input rows, then the
// product with W^T, then bias / activation / accumulation) -- lets the host-side plumbing of the actors run in the CPU tests
extern "C" int truss_gcn_layer(const truss_gcn_layer_args_t *a, void *) {
  if (!a || a->struct_size != sizeof(truss_gcn_layer_args_t)) return tb_fail(TRUSS_EINVAL, "truss_gcn_layer: bad argument block");
  if (!a->x || !a->adj || !a->w || !a->out) return tb_fail(TRUSS_EINVAL, "truss_gcn_layer: a required pointer is NULL");
  if (a->c_out > 224 || a->n_nodes > 256 || (a->nbr ? (a->k_nbr < 1 || a->k_nbr > 16) : a->n_nodes > 64))
    return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_layer: shape outside the kernel's envelope");
  const int N = a->n_nodes, K = a->k_in, C = a->c_out;
  const long xs = a->x_row_stride ? a->x_row_stride : K, os = a->out_row_stride ? a->out_row_stride : C;
  std::vector<float> xa((size_t)N * K);
  for (int b = 0; b < a->n_batch; ++b) {
    const float *A = a->adj + (size_t)b * a->a_batch_stride;
    const float *X = a->x + (size_t)b * N * xs;
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < K; ++k) {
        float acc = 0.0f;
        if (a->nbr) {
          for (int t = 0; t < a->k_nbr; ++t) {
            const int j = a->nbr[i * a->k_nbr + t];
            if (j >= 0) acc += A[(size_t)i * N + j] * X[(size_t)j * xs + k];
          }
        } else {
          for (int j = 0; j < N; ++j) acc += A[(size_t)i * N + j] * X[(size_t)j * xs + k];
        }
        xa[(size_t)i * K + k] = acc;
      }
    for (int i = 0; i < N; ++i)
      for (int c = 0; c < C; ++c) {
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc += xa[(size_t)i * K + k] * a->w[(size_t)c * K + k];
        acc += a->bias ? a->bias[c] : 0.0f;
        if (a->act == 1) acc = acc > 0.0f ? acc : 0.0f;
        else if (a->act == 2) acc = 1.0f / (1.0f + expf(-acc));
        float *o = a->out + ((size_t)b * N + i) * os + c;
        *o = a->accumulate ? *o + acc : acc;
      }
  }
  return TRUSS_OK;
}

// truss_gcn_level, CPU stand-in: the layers one after the other through the stand-in above, X' = A X stored where asked for
extern "C" int truss_gcn_level(const truss_gcn_layer_args_t *layers, int32_t n_layers, float *const *x_agg, void *st) {
  if (n_layers < 0 || (n_layers > 0 && !layers)) return tb_fail(TRUSS_EINVAL, "truss_gcn_level: bad argument");
  for (int l = 0; l < n_layers; ++l) {
    const truss_gcn_layer_args_t *a = layers + l;
    if (a->struct_size != sizeof(truss_gcn_layer_args_t)) return tb_fail(TRUSS_EINVAL, "truss_gcn_level: bad argument block");
    if (a->n_batch == 0) continue;
    if (a->accumulate || a->w_bf16x3 || a->n_nodes > 128 || a->k_in > 256)
      return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_level: shape / mode outside the kernel's envelope");
    if (int rc = truss_gcn_layer(a, st)) return rc;
    if (!x_agg || !x_agg[l]) continue;
    const int N = a->n_nodes, K = a->k_in;
    const long xs = a->x_row_stride ? a->x_row_stride : K;
    for (int b = 0; b < a->n_batch; ++b) {
      const float *A = a->adj + (size_t)b * a->a_batch_stride;
      const float *X = a->x + (size_t)b * N * xs;
      for (int i = 0; i < N; ++i)
        for (int k = 0; k < K; ++k) {
          float acc = 0.0f;
          for (int t = 0; t < (a->nbr ? a->k_nbr : N); ++t) {
            const int j = a->nbr ? a->nbr[i * a->k_nbr + t] : t;
            if (j >= 0) acc += A[(size_t)i * N + j] * X[(size_t)j * xs + k];
          }
          x_agg[l][((size_t)b * N + i) * K + k] = acc;
        }
    }
  }
  return TRUSS_OK;
}

// the exact three-term bfloat16 split of the bf16x3 path (host restatement); the emulated layer itself sums in float32
extern "C" int truss_gcn_split_w(const float *w, int32_t c_out, int32_t k_in, uint16_t *out, void *) {
  if (!w || !out || c_out < 1 || k_in < 1 || c_out > 224) return tb_fail(TRUSS_EINVAL, "truss_gcn_split_w: bad argument");
  const int KP = (k_in + 15) & ~15, CP = 224;
  for (int c = 0; c < CP; ++c)
    for (int k = 0; k < KP; ++k) {
      const float x = (c < c_out && k < k_in) ? w[(size_t)c * k_in + k] : 0.0f;
      uint32_t u, u0, u1, u2;
      memcpy(&u, &x, 4);
      u0 = u & 0xffff0000u;
      float f0, r1, f1, r2;
      memcpy(&f0, &u0, 4);
      r1 = x - f0;
      memcpy(&u1, &r1, 4);
      u1 &= 0xffff0000u;
      memcpy(&f1, &u1, 4);
      r2 = r1 - f1;
      memcpy(&u2, &r2, 4);
      const size_t i = (size_t)c * KP + k;
      out[i] = (uint16_t)(u0 >> 16);
      out[(size_t)CP * KP + i] = (uint16_t)(u1 >> 16);
      out[2 * (size_t)CP * KP + i] = (uint16_t)(u2 >> 16);
    }
  return TRUSS_OK;
}
