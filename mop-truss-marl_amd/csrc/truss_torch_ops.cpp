// truss_torch_ops.cpp -- PyTorch custom operators in front of the C ABI of include/truss_mi355.h.
//
//   torch.ops.truss_mi355.step / rollout / obs / front / gcn_aggregate / gcn_aggregate_sparse / gcn_layer / gcn_level
//
// The reference's hot path runs inside TensorFlow ops on its side of the loop (truss2D_RL.py:328-354); here the env
// step itself is an operator of the host framework: tensors in, tensors mutated in place, launched on the stream the
// caller names (torch's current stream), capturable in a hipGraph, traceable (Meta kernels).  The operators do no
// arithmetic: they check tensors (device, dtype, contiguity, sizes), fill the ABI's argument block with data
// pointers and call the entry point of the native library the caller bound (`lib`: the HIP product library, or the
// CPU lane emulator of the test-suite) -- the same entry points a ctypes / cffi binding would call.
// Built by csrc/Makefile with g++ against the installed libtorch; no HIP headers needed.
#include <torch/library.h>
#include <ATen/ATen.h>

#include <array>
#include <vector>
#include <string>

#include "../../include/truss_mi355.h"

namespace {

struct Backend {
  int (*step)(const truss_topo_t *, const truss_step_args_t *, void *) = nullptr;
  int (*rollout)(const truss_topo_t *, const truss_step_args_t *, int32_t, int32_t, void *) = nullptr;
  int (*obs)(const truss_topo_t *, const truss_obs_args_t *, void *) = nullptr;
  int (*front)(const truss_front_args_t *, void *) = nullptr;
  int (*gcn)(const float *, int64_t, const float *, const float *, float *, int32_t, int32_t, int32_t, int32_t, void *) = nullptr;
  int (*gcn_sparse)(const float *, int64_t, const int16_t *, int32_t, const float *, const float *, float *, int32_t, int32_t, int32_t,
                    int32_t, void *) = nullptr;
  int (*gcn_layer)(const truss_gcn_layer_args_t *, void *) = nullptr;
  int (*gcn_split)(const float *, int32_t, int32_t, uint16_t *, void *) = nullptr;
  int (*gcn_level)(const truss_gcn_layer_args_t *, int32_t, float *const *, void *) = nullptr;
  const char *(*last_error)(void) = nullptr;
  bool device = false;   // true: the HIP library (tensors must be on a cuda device)
};
std::array<Backend, 8> g_backends;

const Backend &backend(int64_t lib) {
  TORCH_CHECK(lib >= 0 && lib < (int64_t)g_backends.size() && g_backends[lib].step, "truss_mi355: native library ", lib,
              " is not bound (truss_mi355.ops binds it when it loads the library)");
  return g_backends[lib];
}

void check_rc(const Backend &b, int rc, const char *what) {
  TORCH_CHECK(rc == TRUSS_OK, what, " failed (", rc, "): ", b.last_error ? b.last_error() : "?");
}

// data pointer of a tensor the ABI will read / write as `dtype`, nullptr for an absent optional
template <typename T>
T *ptr(const Backend &b, const at::Tensor &t, at::ScalarType dtype, const char *name, int64_t min_numel = 0) {
  TORCH_CHECK(t.scalar_type() == dtype, "truss_mi355: ", name, " must be ", dtype, ", got ", t.scalar_type());
  TORCH_CHECK(t.is_contiguous(), "truss_mi355: ", name, " must be contiguous");
  TORCH_CHECK(b.device ? t.is_cuda() : t.is_cpu(), "truss_mi355: ", name, " is on ", t.device(), ", the bound library needs ",
              b.device ? "a cuda (ROCm) device" : "the cpu");
  TORCH_CHECK(t.numel() >= min_numel, "truss_mi355: ", name, " has ", t.numel(), " elements, needs ", min_numel);
  return (T *)t.data_ptr();
}
template <typename T>
T *ptr(const Backend &b, const c10::optional<at::Tensor> &t, at::ScalarType dtype, const char *name, int64_t min_numel = 0) {
  return t.has_value() && t->defined() ? ptr<T>(b, *t, dtype, name, min_numel) : nullptr;
}

using OT = c10::optional<at::Tensor>;

void fill_step(const Backend &b, truss_step_args_t &a, int64_t flags, int64_t n_envs, int64_t N, int64_t E, int64_t sets,
               const at::Tensor &x, const at::Tensor &y_in, const at::Tensor &sec_in, const OT &max_up_in, const OT &max_down_in,
               const OT &a_geo, const OT &a_topo, const OT &coin, const at::Tensor &target, const at::Tensor &env_params,
               const at::Tensor &y_out, const OT &sec_out, const OT &max_up_out, const OT &max_down_out, const at::Tensor &disp,
               const at::Tensor &q0, const at::Tensor &sr, const at::Tensor &comp, const at::Tensor &point, const OT &obj,
               const OT &disp_f64, const OT &q0_f64, const OT &energy, const OT &reactions, const OT &status, const OT &x_n,
               const OT &A_s, const OT &A_n_ts, const OT &A_n_cs, const OT &nN_x_n, const OT &nN_x_e) {
  TORCH_CHECK(n_envs >= 1 && N >= 2 && E >= 1, "truss_mi355: bad n_envs / n_nodes / n_elems");
  const int64_t B = n_envs, BN = B * N, BE = B * E;
  const auto f32 = at::kFloat, f64 = at::kDouble, i32 = at::kInt, u8 = at::kByte;
  a = truss_step_args_t{};
  a.struct_size = sizeof(truss_step_args_t);
  a.n_envs = (int32_t)B;
  a.flags = (uint32_t)flags;
  a.x = ptr<const float>(b, x, f32, "x", BN);
  a.y_in = ptr<const float>(b, y_in, f32, "y_in", BN);
  a.sec_in = ptr<const int32_t>(b, sec_in, i32, "sec_in", BE);
  a.max_up_in = ptr<const float>(b, max_up_in, f32, "max_up_in", BN);
  a.max_down_in = ptr<const float>(b, max_down_in, f32, "max_down_in", BN);
  a.a_geo = ptr<float>(b, a_geo, f32, "a_geo", sets * BN * 2);
  a.a_topo = ptr<float>(b, a_topo, f32, "a_topo", sets * BN * 3);
  a.coin = ptr<const uint8_t>(b, coin, u8, "coin", B);
  a.target = ptr<const float>(b, target, f32, "target", BN);
  a.env_params = ptr<const double>(b, env_params, f64, "env_params", B * TRUSS_NPARAM);
  a.y_out = ptr<float>(b, y_out, f32, "y_out", BN);
  a.sec_out = ptr<int32_t>(b, sec_out, i32, "sec_out", BE);
  a.max_up_out = ptr<float>(b, max_up_out, f32, "max_up_out", BN);
  a.max_down_out = ptr<float>(b, max_down_out, f32, "max_down_out", BN);
  a.disp = ptr<float>(b, disp, f32, "disp", BN * 2);
  a.q0 = ptr<float>(b, q0, f32, "q0", BE);
  a.sr = ptr<float>(b, sr, f32, "sr", BE);
  a.comp = ptr<uint8_t>(b, comp, u8, "comp", BE);
  a.point = ptr<float>(b, point, f32, "point", B * 4);
  a.obj = ptr<float>(b, obj, f32, "obj", B * 2);
  a.disp_f64 = ptr<double>(b, disp_f64, f64, "disp_f64", BN * 2);
  a.q0_f64 = ptr<double>(b, q0_f64, f64, "q0_f64", BE);
  a.energy = ptr<double>(b, energy, f64, "energy", B);
  a.reactions = ptr<double>(b, reactions, f64, "reactions", B);
  a.status = ptr<int32_t>(b, status, i32, "status", B);
  a.x_n = ptr<float>(b, x_n, f32, "x_n", BN * 13);
  a.A_s = ptr<float>(b, A_s, f32, "A_s", BN * N);
  a.A_n_ts = ptr<float>(b, A_n_ts, f32, "A_n_ts", BN * N);
  a.A_n_cs = ptr<float>(b, A_n_cs, f32, "A_n_cs", BN * N);
  a.nN_x_n = ptr<float>(b, nN_x_n, f32, "nN_x_n", BN * 12);
  a.nN_x_e = ptr<float>(b, nN_x_e, f32, "nN_x_e", BE * 21);
}

#define STEP_TENSOR_PARAMS                                                                                                     \
  const at::Tensor &x, const at::Tensor &y_in, const at::Tensor &sec_in, const OT &max_up_in, const OT &max_down_in,          \
      const OT &a_geo, const OT &a_topo, const OT &coin, const at::Tensor &target, const at::Tensor &env_params,              \
      const at::Tensor &y_out, const OT &sec_out, const OT &max_up_out, const OT &max_down_out, const at::Tensor &disp,       \
      const at::Tensor &q0, const at::Tensor &sr, const at::Tensor &comp, const at::Tensor &point, const OT &obj,             \
      const OT &disp_f64, const OT &q0_f64, const OT &energy, const OT &reactions, const OT &status, const OT &x_n,           \
      const OT &A_s, const OT &A_n_ts, const OT &A_n_cs, const OT &nN_x_n, const OT &nN_x_e
#define STEP_TENSOR_ARGS                                                                                                       \
  x, y_in, sec_in, max_up_in, max_down_in, a_geo, a_topo, coin, target, env_params, y_out, sec_out, max_up_out, max_down_out,  \
      disp, q0, sr, comp, point, obj, disp_f64, q0_f64, energy, reactions, status, x_n, A_s, A_n_ts, A_n_cs, nN_x_n, nN_x_e

// One Game_research04._game_modify per env (truss2D_ENV.py:370-525) == truss_step
void step(int64_t lib, int64_t topo, int64_t stream, int64_t flags, int64_t n_envs, int64_t n_nodes, int64_t n_elems,
          STEP_TENSOR_PARAMS) {
  const Backend &b = backend(lib);
  truss_step_args_t a;
  fill_step(b, a, flags, n_envs, n_nodes, n_elems, 1, STEP_TENSOR_ARGS);
  check_rc(b, b.step((const truss_topo_t *)topo, &a, (void *)stream), "truss_step");
}
// n_steps chained transitions (truss_rollout); a_geo / a_topo hold n_action_sets action sets
void rollout(int64_t lib, int64_t topo, int64_t stream, int64_t flags, int64_t n_envs, int64_t n_nodes, int64_t n_elems,
             int64_t n_steps, int64_t n_action_sets, STEP_TENSOR_PARAMS) {
  const Backend &b = backend(lib);
  truss_step_args_t a;
  fill_step(b, a, flags, n_envs, n_nodes, n_elems, n_action_sets, STEP_TENSOR_ARGS);
  check_rc(b, b.rollout((const truss_topo_t *)topo, &a, (int32_t)n_steps, (int32_t)n_action_sets, (void *)stream), "truss_rollout");
}
void step_meta(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, STEP_TENSOR_PARAMS) {}
void rollout_meta(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, STEP_TENSOR_PARAMS) {}

// state_data + state_data_not_norm (truss2D_ENV.py:40-193) == truss_obs
void obs(int64_t lib, int64_t topo, int64_t stream, int64_t n_envs, int64_t N, int64_t E, const at::Tensor &x, const at::Tensor &y,
         const at::Tensor &sec, const at::Tensor &max_up, const at::Tensor &max_down, const at::Tensor &target, const at::Tensor &disp,
         const at::Tensor &q0, const at::Tensor &sr, const at::Tensor &comp, const at::Tensor &env_params, const OT &x_n, const OT &A_s,
         const OT &A_n_ts, const OT &A_n_cs, const OT &nN_x_n, const OT &nN_x_e) {
  const Backend &b = backend(lib);
  TORCH_CHECK(n_envs >= 1, "truss_mi355: n_envs < 1");
  const int64_t B = n_envs, BN = B * N, BE = B * E;
  const auto f32 = at::kFloat;
  truss_obs_args_t a{};
  a.struct_size = sizeof(truss_obs_args_t);
  a.n_envs = (int32_t)B;
  a.x = ptr<const float>(b, x, f32, "x", BN);
  a.y = ptr<const float>(b, y, f32, "y", BN);
  a.sec = ptr<const int32_t>(b, sec, at::kInt, "sec", BE);
  a.max_up = ptr<const float>(b, max_up, f32, "max_up", BN);
  a.max_down = ptr<const float>(b, max_down, f32, "max_down", BN);
  a.target = ptr<const float>(b, target, f32, "target", BN);
  a.disp = ptr<const float>(b, disp, f32, "disp", BN * 2);
  a.q0 = ptr<const float>(b, q0, f32, "q0", BE);
  a.sr = ptr<const float>(b, sr, f32, "sr", BE);
  a.comp = ptr<const uint8_t>(b, comp, at::kByte, "comp", BE);
  a.env_params = ptr<const double>(b, env_params, at::kDouble, "env_params", B * TRUSS_NPARAM);
  a.x_n = ptr<float>(b, x_n, f32, "x_n", BN * 13);
  a.A_s = ptr<float>(b, A_s, f32, "A_s", BN * N);
  a.A_n_ts = ptr<float>(b, A_n_ts, f32, "A_n_ts", BN * N);
  a.A_n_cs = ptr<float>(b, A_n_cs, f32, "A_n_cs", BN * N);
  a.nN_x_n = ptr<float>(b, nN_x_n, f32, "nN_x_n", BN * 12);
  a.nN_x_e = ptr<float>(b, nN_x_e, f32, "nN_x_e", BE * 21);
  check_rc(b, b.obs((const truss_topo_t *)topo, &a, (void *)stream), "truss_obs");
}
void obs_meta(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const at::Tensor &, const at::Tensor &, const at::Tensor &,
              const at::Tensor &, const at::Tensor &, const at::Tensor &, const at::Tensor &, const at::Tensor &, const at::Tensor &,
              const at::Tensor &, const at::Tensor &, const OT &, const OT &, const OT &, const OT &, const OT &, const OT &) {}

// Pareto cull + 2-D hypervolume of B small point sets (utils.py:11-342) == truss_front
void front(int64_t lib, int64_t stream, int64_t max_front, int64_t flags, const at::Tensor &points, const at::Tensor &n_points,
           const OT &ref_points, const OT &front_idx, const OT &n_front, const OT &hv_front, const OT &hv_all, const OT &metrics) {
  const Backend &b = backend(lib);
  TORCH_CHECK(points.dim() == 3 && points.size(2) == 4, "truss_mi355: points must be [B, P, 4]");
  const int64_t B = points.size(0), P = points.size(1);
  truss_front_args_t a{};
  a.struct_size = sizeof(truss_front_args_t);
  a.n_envs = (int32_t)B;
  a.max_points = (int32_t)P;
  a.max_front = (int32_t)max_front;
  a.flags = (uint32_t)flags;
  a.points = ptr<const double>(b, points, at::kDouble, "points");
  a.n_points = ptr<const int32_t>(b, n_points, at::kInt, "n_points", B);
  a.ref_points = ptr<const double>(b, ref_points, at::kDouble, "ref_points", B * 2);
  a.front_idx = ptr<int32_t>(b, front_idx, at::kInt, "front_idx", B * P);
  a.n_front = ptr<int32_t>(b, n_front, at::kInt, "n_front", B);
  a.hv_front = ptr<double>(b, hv_front, at::kDouble, "hv_front", B);
  a.hv_all = ptr<double>(b, hv_all, at::kDouble, "hv_all", B);
  a.metrics = ptr<double>(b, metrics, at::kDouble, "metrics", B * 5);
  if (B == 0) return;
  check_rc(b, b.front(&a, (void *)stream), "truss_front");
}
void front_meta(int64_t, int64_t, int64_t, int64_t, const at::Tensor &, const at::Tensor &, const OT &, const OT &, const OT &, const OT &,
                const OT &, const OT &) {}

// act(A @ H + bias) of a GCN layer for B small graphs (truss2D_RL.py:49-120, inference) == truss_gcn_aggregate
void gcn_aggregate(int64_t lib, int64_t stream, const at::Tensor &adj, const at::Tensor &h, const OT &bias, const at::Tensor &out,
                   int64_t act) {
  const Backend &b = backend(lib);
  TORCH_CHECK(h.dim() == 3 && out.sizes() == h.sizes(), "truss_mi355: h / out must be [B, N, C] of equal shape");
  const int64_t B = h.size(0), N = h.size(1), C = h.size(2);
  TORCH_CHECK((adj.dim() == 2 || adj.dim() == 3) && adj.size(-1) == N && adj.size(-2) == N && (adj.dim() == 2 || adj.size(0) == B),
              "truss_mi355: adj must be [N, N] or [B, N, N]");
  const float *pa = ptr<const float>(b, adj, at::kFloat, "adj");
  const float *ph = ptr<const float>(b, h, at::kFloat, "h");
  const float *pb = ptr<const float>(b, bias, at::kFloat, "bias", C);
  float *po = ptr<float>(b, out, at::kFloat, "out");
  check_rc(b, b.gcn(pa, adj.dim() == 3 ? N * N : 0, ph, pb, po, (int32_t)B, (int32_t)N, (int32_t)C, (int32_t)act, (void *)stream),
           "truss_gcn_aggregate");
}
void gcn_meta(int64_t, int64_t, const at::Tensor &, const at::Tensor &, const OT &, const at::Tensor &, int64_t) {}

// the same over a fixed sparsity pattern (nbr [N, K] int16: the columns that may be non-zero in row i) == truss_gcn_aggregate_sparse
void gcn_aggregate_sparse(int64_t lib, int64_t stream, const at::Tensor &adj, const at::Tensor &nbr, const at::Tensor &h, const OT &bias,
                          const at::Tensor &out, int64_t act) {
  const Backend &b = backend(lib);
  TORCH_CHECK(b.gcn_sparse, "truss_mi355: the bound native library has no truss_gcn_aggregate_sparse");
  TORCH_CHECK(h.dim() == 3 && out.sizes() == h.sizes(), "truss_mi355: h / out must be [B, N, C] of equal shape");
  const int64_t B = h.size(0), N = h.size(1), C = h.size(2);
  TORCH_CHECK((adj.dim() == 2 || adj.dim() == 3) && adj.size(-1) == N && adj.size(-2) == N && (adj.dim() == 2 || adj.size(0) == B),
              "truss_mi355: adj must be [N, N] or [B, N, N]");
  TORCH_CHECK(nbr.dim() == 2 && nbr.size(0) == N, "truss_mi355: nbr must be [N, K]");
  const float *pa = ptr<const float>(b, adj, at::kFloat, "adj");
  const int16_t *pn = ptr<const int16_t>(b, nbr, at::kShort, "nbr");
  const float *ph = ptr<const float>(b, h, at::kFloat, "h");
  const float *pb = ptr<const float>(b, bias, at::kFloat, "bias", C);
  float *po = ptr<float>(b, out, at::kFloat, "out");
  check_rc(b, b.gcn_sparse(pa, adj.dim() == 3 ? N * N : 0, pn, (int32_t)nbr.size(1), ph, pb, po, (int32_t)B, (int32_t)N, (int32_t)C,
                           (int32_t)act, (void *)stream),
           "truss_gcn_aggregate_sparse");
}
void gcn_sparse_meta(int64_t, int64_t, const at::Tensor &, const at::Tensor &, const at::Tensor &, const OT &, const at::Tensor &, int64_t) {}

// one whole GCN layer, out = act(adj @ (x @ w^T) + bias) [+ out], on the matrix cores == truss_gcn_layer
void gcn_layer(int64_t lib, int64_t stream, const at::Tensor &x, const at::Tensor &adj, const OT &nbr, const at::Tensor &w, const OT &bias,
               const at::Tensor &out, int64_t act, bool accumulate, const OT &w_split) {
  const Backend &b = backend(lib);
  TORCH_CHECK(b.gcn_layer, "truss_mi355: the bound native library has no truss_gcn_layer");
  TORCH_CHECK(x.dim() == 3 && out.dim() == 3 && w.dim() == 2, "truss_mi355: x [B, N, K], w [C, K], out [B, N, C]");
  const int64_t B = x.size(0), N = x.size(1), K = x.size(2), C = w.size(0);
  TORCH_CHECK(w.size(1) == K && out.size(0) == B && out.size(1) == N && out.size(2) == C, "truss_mi355: gcn_layer shapes do not match");
  TORCH_CHECK((adj.dim() == 2 || adj.dim() == 3) && adj.size(-1) == N && adj.size(-2) == N && (adj.dim() == 2 || adj.size(0) == B),
              "truss_mi355: adj must be [N, N] or [B, N, N]");
  truss_gcn_layer_args_t a{};
  a.struct_size = sizeof a;
  a.n_batch = (int32_t)B;
  a.n_nodes = (int32_t)N;
  a.k_in = (int32_t)K;
  a.c_out = (int32_t)C;
  a.act = (int32_t)act;
  a.accumulate = accumulate ? 1 : 0;
  a.x = ptr<const float>(b, x, at::kFloat, "x");
  a.adj = ptr<const float>(b, adj, at::kFloat, "adj");
  a.a_batch_stride = adj.dim() == 3 ? N * N : 0;
  if (nbr.has_value() && nbr->defined()) {
    TORCH_CHECK(nbr->dim() == 2 && nbr->size(0) == N, "truss_mi355: nbr must be [N, K]");
    a.nbr = ptr<const int16_t>(b, *nbr, at::kShort, "nbr");
    a.k_nbr = (int32_t)nbr->size(1);
  }
  a.w = ptr<const float>(b, w, at::kFloat, "w");
  a.bias = ptr<const float>(b, bias, at::kFloat, "bias", C);
  a.out = ptr<float>(b, out, at::kFloat, "out");
  a.w_bf16x3 = (const uint16_t *)ptr<const int16_t>(b, w_split, at::kShort, "w_split", 3 * 224 * ((K + 15) / 16 * 16));
  check_rc(b, b.gcn_layer(&a, (void *)stream), "truss_gcn_layer");
}
void gcn_layer_meta(int64_t, int64_t, const at::Tensor &, const at::Tensor &, const OT &, const at::Tensor &, const OT &, const at::Tensor &,
                    int64_t, bool, const OT &) {}

// a whole level of GCN layers in one launch: out[i] = act[i](adj[i] @ (x[i] @ w[i]^T) + bias[i]); x_agg (empty, or one tensor per
// layer) receives adj[i] @ x[i] == truss_gcn_level
void gcn_level(int64_t lib, int64_t stream, at::TensorList x, at::TensorList adj, const c10::List<OT> &nbr, at::TensorList w, at::TensorList bias,
               at::TensorList out, at::TensorList x_agg, at::IntArrayRef act) {
  const Backend &b = backend(lib);
  TORCH_CHECK(b.gcn_level, "truss_mi355: the bound native library has no truss_gcn_level");
  const size_t L = x.size();
  TORCH_CHECK(adj.size() == L && w.size() == L && bias.size() == L && out.size() == L && act.size() == L && (x_agg.empty() || x_agg.size() == L) &&
                  (nbr.empty() || nbr.size() == L),
              "truss_mi355: gcn_level takes one entry per layer in every list");
  std::vector<truss_gcn_layer_args_t> args(L);
  std::vector<float *> xa(L, nullptr);
  for (size_t i = 0; i < L; ++i) {
    TORCH_CHECK(x[i].dim() == 3 && out[i].dim() == 3 && w[i].dim() == 2, "truss_mi355: x [B, N, K], w [C, K], out [B, N, C]");
    const int64_t B = x[i].size(0), N = x[i].size(1), K = x[i].size(2), C = w[i].size(0);
    TORCH_CHECK(w[i].size(1) == K && out[i].size(0) == B && out[i].size(1) == N && out[i].size(2) == C, "truss_mi355: gcn_level shapes do not match");
    TORCH_CHECK((adj[i].dim() == 2 || adj[i].dim() == 3) && adj[i].size(-1) == N && adj[i].size(-2) == N && (adj[i].dim() == 2 || adj[i].size(0) == B),
                "truss_mi355: adj must be [N, N] or [B, N, N]");
    truss_gcn_layer_args_t &a = args[i];
    a = truss_gcn_layer_args_t{};
    a.struct_size = sizeof a;
    a.n_batch = (int32_t)B;
    a.n_nodes = (int32_t)N;
    a.k_in = (int32_t)K;
    a.c_out = (int32_t)C;
    a.act = (int32_t)act[i];
    a.x = ptr<const float>(b, x[i], at::kFloat, "x");
    a.adj = ptr<const float>(b, adj[i], at::kFloat, "adj");
    a.a_batch_stride = adj[i].dim() == 3 ? N * N : 0;
    if (!nbr.empty()) {
      const OT pat = nbr.get(i);
      if (pat.has_value() && pat->defined()) {
        TORCH_CHECK(pat->dim() == 2 && pat->size(0) == N, "truss_mi355: nbr must be [N, K]");
        a.nbr = ptr<const int16_t>(b, *pat, at::kShort, "nbr");
        a.k_nbr = (int32_t)pat->size(1);
      }
    }
    a.w = ptr<const float>(b, w[i], at::kFloat, "w");
    a.bias = ptr<const float>(b, bias[i], at::kFloat, "bias", C);
    a.out = ptr<float>(b, out[i], at::kFloat, "out");
    if (!x_agg.empty()) {
      TORCH_CHECK(x_agg[i].numel() == B * N * K, "truss_mi355: x_agg[i] must hold B * N * K elements");
      xa[i] = ptr<float>(b, x_agg[i], at::kFloat, "x_agg");
    }
  }
  check_rc(b, b.gcn_level(args.data(), (int32_t)L, x_agg.empty() ? nullptr : xa.data(), (void *)stream), "truss_gcn_level");
}
void gcn_level_meta(int64_t, int64_t, at::TensorList, at::TensorList, const c10::List<OT> &, at::TensorList, at::TensorList, at::TensorList, at::TensorList,
                    at::IntArrayRef) {}

// w [C, K] float32 -> out [3, 224, KP] int16 (bfloat16 bit patterns, zero rows / columns beyond C / K): the exact three-term split of the bf16x3 path == truss_gcn_split_w
void gcn_split_w(int64_t lib, int64_t stream, const at::Tensor &w, const at::Tensor &out) {
  const Backend &b = backend(lib);
  TORCH_CHECK(b.gcn_split, "truss_mi355: the bound native library has no truss_gcn_split_w");
  TORCH_CHECK(w.dim() == 2 && out.dim() == 3 && out.size(0) == 3 && out.size(1) == 224 && w.size(0) <= 224 && out.size(2) == (w.size(1) + 15) / 16 * 16,
              "truss_mi355: gcn_split_w: w [C <= 224, K], out [3, 224, (K + 15) / 16 * 16]");
  check_rc(b, b.gcn_split(ptr<const float>(b, w, at::kFloat, "w"), (int32_t)w.size(0), (int32_t)w.size(1),
                          (uint16_t *)ptr<int16_t>(b, out, at::kShort, "out"), (void *)stream), "truss_gcn_split_w");
}
void gcn_split_meta(int64_t, int64_t, const at::Tensor &, const at::Tensor &) {}

}  // namespace

// Bind the entry points of a loaded native library (addresses from ctypes) under a small index.
extern "C" int truss_torch_bind(int lib, void *step_fn, void *rollout_fn, void *obs_fn, void *front_fn, void *gcn_fn, void *gcn_sparse_fn,
                                void *gcn_layer_fn, void *gcn_split_fn, void *gcn_level_fn, void *last_error_fn, int is_device) {
  if (lib < 0 || lib >= (int)g_backends.size() || !step_fn) return -1;
  Backend &b = g_backends[lib];
  b.step = (decltype(b.step))step_fn;
  b.rollout = (decltype(b.rollout))rollout_fn;
  b.obs = (decltype(b.obs))obs_fn;
  b.front = (decltype(b.front))front_fn;
  b.gcn = (decltype(b.gcn))gcn_fn;
  b.gcn_sparse = (decltype(b.gcn_sparse))gcn_sparse_fn;
  b.gcn_layer = (decltype(b.gcn_layer))gcn_layer_fn;
  b.gcn_split = (decltype(b.gcn_split))gcn_split_fn;
  b.gcn_level = (decltype(b.gcn_level))gcn_level_fn;
  b.last_error = (decltype(b.last_error))last_error_fn;
  b.device = is_device != 0;
  return 0;
}

#define STEP_SCHEMA_TENSORS                                                                                                          \
  "Tensor x, Tensor y_in, Tensor sec_in, Tensor? max_up_in, Tensor? max_down_in, Tensor(a!)? a_geo, Tensor(b!)? a_topo, "            \
  "Tensor? coin, Tensor target, Tensor env_params, Tensor(c!) y_out, Tensor(d!)? sec_out, Tensor(e!)? max_up_out, "                  \
  "Tensor(f!)? max_down_out, Tensor(g!) disp, Tensor(h!) q0, Tensor(i!) sr, Tensor(j!) comp, Tensor(k!) point, Tensor(l!)? obj, "    \
  "Tensor(m!)? disp_f64, Tensor(n!)? q0_f64, Tensor(o!)? energy, Tensor(p!)? reactions, Tensor(q!)? status, Tensor(r!)? x_n, "        \
  "Tensor(s!)? A_s, Tensor(t!)? A_n_ts, Tensor(u!)? A_n_cs, Tensor(v!)? nN_x_n, Tensor(w!)? nN_x_e) -> ()"

TORCH_LIBRARY(truss_mi355, m) {
  m.def("step(int lib, int topo, int stream, int flags, int n_envs, int n_nodes, int n_elems, " STEP_SCHEMA_TENSORS);
  m.def("rollout(int lib, int topo, int stream, int flags, int n_envs, int n_nodes, int n_elems, int n_steps, int n_action_sets, "
        STEP_SCHEMA_TENSORS);
  m.def("obs(int lib, int topo, int stream, int n_envs, int n_nodes, int n_elems, Tensor x, Tensor y, Tensor sec, Tensor max_up, "
        "Tensor max_down, Tensor target, Tensor disp, Tensor q0, Tensor sr, Tensor comp, Tensor env_params, Tensor(a!)? x_n, "
        "Tensor(b!)? A_s, Tensor(c!)? A_n_ts, Tensor(d!)? A_n_cs, Tensor(e!)? nN_x_n, Tensor(f!)? nN_x_e) -> ()");
  m.def("front(int lib, int stream, int max_front, int flags, Tensor points, Tensor n_points, Tensor? ref_points, "
        "Tensor(a!)? front_idx, Tensor(b!)? n_front, Tensor(c!)? hv_front, Tensor(d!)? hv_all, Tensor(e!)? metrics) -> ()");
  m.def("gcn_aggregate(int lib, int stream, Tensor adj, Tensor h, Tensor? bias, Tensor(a!) out, int act) -> ()");
  m.def("gcn_aggregate_sparse(int lib, int stream, Tensor adj, Tensor nbr, Tensor h, Tensor? bias, Tensor(a!) out, int act) -> ()");
  m.def("gcn_layer(int lib, int stream, Tensor x, Tensor adj, Tensor? nbr, Tensor w, Tensor? bias, Tensor(a!) out, int act, bool accumulate, "
        "Tensor? w_split) -> ()");
  m.def("gcn_split_w(int lib, int stream, Tensor w, Tensor(a!) out) -> ()");
  m.def("gcn_level(int lib, int stream, Tensor[] x, Tensor[] adj, Tensor?[] nbr, Tensor[] w, Tensor[] bias, Tensor(a!)[] out, Tensor(b!)[] x_agg, "
        "int[] act) -> ()");
}
TORCH_LIBRARY_IMPL(truss_mi355, CPU, m) {   // the emulator library of the test-suite binds here
  m.impl("step", step);
  m.impl("rollout", rollout);
  m.impl("obs", obs);
  m.impl("front", front);
  m.impl("gcn_aggregate", gcn_aggregate);
  m.impl("gcn_aggregate_sparse", gcn_aggregate_sparse);
  m.impl("gcn_layer", gcn_layer);
  m.impl("gcn_split_w", gcn_split_w);
  m.impl("gcn_level", gcn_level);
}
TORCH_LIBRARY_IMPL(truss_mi355, CUDA, m) {  // = HIP on ROCm: the product library
  m.impl("step", step);
  m.impl("rollout", rollout);
  m.impl("obs", obs);
  m.impl("front", front);
  m.impl("gcn_aggregate", gcn_aggregate);
  m.impl("gcn_aggregate_sparse", gcn_aggregate_sparse);
  m.impl("gcn_layer", gcn_layer);
  m.impl("gcn_split_w", gcn_split_w);
  m.impl("gcn_level", gcn_level);
}
TORCH_LIBRARY_IMPL(truss_mi355, Meta, m) {  // tracing: every operator only mutates its outputs
  m.impl("step", step_meta);
  m.impl("rollout", rollout_meta);
  m.impl("obs", obs_meta);
  m.impl("front", front_meta);
  m.impl("gcn_aggregate", gcn_meta);
  m.impl("gcn_aggregate_sparse", gcn_sparse_meta);
  m.impl("gcn_layer", gcn_layer_meta);
  m.impl("gcn_split_w", gcn_split_meta);
  m.impl("gcn_level", gcn_level_meta);
}
