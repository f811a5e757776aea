/*
 * truss_mi355.h -- C ABI of the MI355X-native batched 2-D truss FEM environment step.
 *
 * This is the drop-in boundary for the hot path of kupc25648/MOP-truss-MARL.
 * The reference has no FFI of its own: its boundary is Python duck typing between
 * master_DDPG_truss2D_MO.py and truss2D_ENV.Game_research04.  The entry points below are what a
 * binding for that path needs; each one cites the reference interface it replaces
 * (paths relative to the reference checkout, train/code/ unless a test copy is named).
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).
 *   - every pointer inside truss_step_args_t is DEVICE memory owned by the caller; the library
 *     never allocates per call and never synchronises the stream (safe inside hipGraph capture).
 *   - all functions return TRUSS_OK (0) or a negative error code; truss_last_error() gives text.
 *   - batch layout is struct-of-arrays with the env index outermost: a[B][N], a[B][E], ...
 *   - heights/actions/observations are float32 exactly as in the reference's state arrays; the
 *     stiffness assembly and the solve are float64 (SURVEY.md §7 "Precision").
 */
#ifndef TRUSS_MI355_H
#define TRUSS_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRUSS_ABI_VERSION 3

#define TRUSS_OK 0
#define TRUSS_EINVAL (-1)       /* bad argument (NULL, size mismatch, pair table not an involution ...) */
#define TRUSS_EUNSUPPORTED (-2) /* topology outside the compiled kernel envelope (bandwidth, N, E) */
#define TRUSS_EHIP (-3)         /* a HIP runtime call failed */
#define TRUSS_ENOMEM (-4)

/* flags for truss_step_args_t.flags */
#define TRUSS_F_NO_DECODE 0x1u     /* analysis only: y_out = y_in, sec_out = sec_in (reset path) */
#define TRUSS_F_CLAMP_INPLACE 0x2u /* write the clamped actions back into a_geo / a_topo
                                      (truss2D_ENV.py:376-388 mutates the caller's arrays) */
#define TRUSS_F_EMIT_OBS 0x4u      /* also write the observation tensors of the NEW design (x_n ... nN_x_e below):
                                      _game_modify builds them in the same call (truss2D_ENV.py:497-500).  One
                                      launch where the topology allows it (truss_topo_fused_obs), otherwise the
                                      observation kernel is queued behind the step on the same stream */

/* columns of env_params[B][TRUSS_NPARAM] (float64) */
#define TRUSS_NPARAM 8
#define TRUSS_P_YMAX 0    /* gen_model.y_max            truss2D_GEN.py:80  */
#define TRUSS_P_DMIN 1    /* gen_model.d_min            truss2D_GEN.py:82  */
#define TRUSS_P_MAXDEF 2  /* gen_model.max_deformation  truss2D_GEN.py:78  */
#define TRUSS_P_LOADX 3   /* Load.size[0]               truss2D_GEN.py:362-364 */
#define TRUSS_P_LOADY 4   /* Load.size[1] */
#define TRUSS_P_INTOBJ1 5 /* Game_research04.int_obj1   truss2D_ENV.py:273 */
#define TRUSS_P_INTOBJ2 6 /* Game_research04.int_obj2   truss2D_ENV.py:274 */
#define TRUSS_P_ISROOF 7  /* truss_type == 'roof' (1.0) / 'bridge' (0.0)  truss2D_GEN.py:70 */

typedef struct truss_topo truss_topo_t; /* opaque: host + device tables of one topology */

/* ---- library ---------------------------------------------------------------------------- */
int truss_abi_version(void);
const char *truss_last_error(void);
/* "hip" for the product library; the CPU lane emulator used by the build's own tests says "emu". */
const char *truss_backend(void);

/* ---- topology (reset time) ----------------------------------------------------------------
 * Replaces Model.add_node/add_element + gen_nsc/gen_tnsc/gen_ndof + the ttnsc half of gen_ssm
 * (FEM_2Dtruss.py:189-198, 227-261, 310-317) and the static output of gen_model.generate
 * (truss2D_GEN.py:241-434): connectivity, supports, top_node, vertical_pair, load placement.
 *
 *   conn[E][2]        0-based node indices (element.nodes[0], element.nodes[1])
 *   res[N][2]         Node.res (1 = restrained)
 *   top[N]            Node.top_node
 *   pair[N]           index of Node.vertical_pair[0]; must be an involution without fixed points when
 *                     the action decode is used; may be NULL for analysis-only topologies
 *   load_mask[2][N]   nodes that carry the load for bridge (row 0) / roof (row 1)
 *                     (truss2D_GEN.py:421-430); NULL = derive by that rule from top/res
 *   sym_nodes[n][2]   (dst, src): y[dst] = y[src] when coin != 0, y[src] = y[dst] when coin == 0
 *                     (hard-coded mirror blocks of test/<star>/code/truss2D_ENV.py:459-553 / :459-675);
 *                     n = 0 for the train copy
 *   sym_elems[n][2]   both elements take min(section) (same blocks)
 *   sections[S][2]    area [m^2], inertia [m^4]  (section_data/01_brace_rod2.csv * 1e-4 / 1e-8,
 *                     truss2D_GEN.py:63-64, 291-292)
 *   node_order[N]     optional hint: node ordering for the banded solver (NULL = library chooses)
 */
int truss_topo_create(truss_topo_t **out, int32_t n_nodes, int32_t n_elems, const int32_t *conn,
                      const uint8_t *res, const uint8_t *top, const int32_t *pair,
                      const uint8_t *load_mask, int32_t n_sym_nodes, const int32_t *sym_nodes,
                      int32_t n_sym_elems, const int32_t *sym_elems, int32_t n_sections,
                      const double *sections, double e_mod, double long_stress,
                      const int32_t *node_order);
int truss_topo_destroy(truss_topo_t *t);

/* DOF numbering exactly as the reference computes it (bit-exact integers):
 *   nsc[2N]    Model.nsc   (FEM_2Dtruss.py:227-245), 1-based, free DOFs first
 *   ttnsc[E][4] Model.ttnsc (FEM_2Dtruss.py:311-317)
 * returns ndof (FEM_2Dtruss.py:255-261) or a negative error. Host pointers, may be NULL. */
int truss_topo_dofs(const truss_topo_t *t, int32_t *nsc, int32_t *ttnsc);

/* Solver view of the same system: perm[ndof] = reference DOF id (0-based) at solver position i,
 * half_bandwidth of K under that ordering, and the kernel window (lanes-per-env x rows-per-lane)
 * that was selected.  Host pointers, may be NULL. */
int truss_topo_solver_info(const truss_topo_t *t, int32_t *perm, int32_t *half_bandwidth,
                           int32_t *lanes_per_env, int32_t *rows_per_lane);

/* 1 when truss_step writes the observation tensors from the step's own launch for this topology
 * (TRUSS_F_EMIT_OBS), 0 when it queues the observation kernel behind it.  Same results either way. */
int truss_topo_fused_obs(const truss_topo_t *t);

/* ---- the batched environment step ----------------------------------------------------------
 * One Game_research04._game_modify(set_node, set_element, nC_e, actions) per env
 * (truss2D_ENV.py:370-525): clamp -> _set_model -> geometry move -> supports/round -> section
 * +-1 -> sequential height repairs -> [symmetry] -> set_moveRange -> Model.gen_all -> objectives.
 * With TRUSS_F_NO_DECODE it is Model.restore(); Model.gen_all() + the objective sums of
 * Game_research04.__init__ (truss2D_ENV.py:264-274) = the reset path (_game_get_1_state :336-340).
 */
#define TRUSS_STATUS_NOT_SPD 1
#define TRUSS_STATUS_OBS_TIMEOUT 2

typedef struct truss_step_args {
  size_t struct_size; /* = sizeof(truss_step_args_t), ABI guard */
  int32_t n_envs;     /* B */
  uint32_t flags;

  /* design state in (device) */
  const float *x;         /* [B][N]   node x (Node.coord[0]) */
  const float *y_in;      /* [B][N]   set_node[:,1]   (truss2D_ENV.py:362) */
  const int32_t *sec_in;  /* [B][E]   set_element[:,0] (truss2D_ENV.py:366) */
  const float *max_up_in; /* [B][N] or NULL. The reference uses the move ranges left on its model by
                             the PREVIOUS analysis (truss2D_ENV.py:402,407); NULL = recompute from
                             y_in (the ranges the parent design itself has, nN_x_n[:,7:9]) */
  const float *max_down_in;
  float *a_geo;           /* [B][N][2] actions[0]; written only with TRUSS_F_CLAMP_INPLACE */
  float *a_topo;          /* [B][N][3] actions[1] */
  const uint8_t *coin;    /* [B] or NULL: 1 when random.random() >= 0.5 (test copies :459) */
  const float *target;    /* [B][N]   Node.target of top nodes (truss2D_GEN.py:369-374) */
  const double *env_params; /* [B][TRUSS_NPARAM] */

  /* design state out */
  float *y_out;        /* [B][N] */
  int32_t *sec_out;    /* [B][E] */
  float *max_up_out;   /* [B][N] or NULL  set_moveRange of the new design (truss2D_GEN.py:118-133) */
  float *max_down_out; /* [B][N] or NULL */

  /* analysis results */
  float *disp;      /* [B][N][2] Node.global_d (FEM_2Dtruss.py:341-352); restrained DOFs = 0 */
  float *q0;        /* [B][E]    Element.e_q[0][0] (FEM_2Dtruss.py:383-386) */
  float *sr;        /* [B][E]    Element.prop_yeield (FEM_2Dtruss.py:414-431) */
  uint8_t *comp;    /* [B][E]    Element.iscompress */
  float *point;     /* [B][4]    [obj1/int_obj1, obj2/int_obj2, con1, con2] (truss2D_ENV.py:518-523); 16-byte aligned (one store per env) */
  float *obj;       /* [B][2] or NULL: raw obj1, obj2 (= int_obj1/2 when called at reset) */
  double *disp_f64; /* [B][N][2] or NULL: float64 copy for solver-level parity checks */
  double *q0_f64;   /* [B][E] or NULL */
  double *energy;   /* [B] or NULL: Model.U_full (FEM_2Dtruss.py:374-379) */
  double *reactions; /* [B][2N-ndof] or NULL: Model.r[ndof:] (FEM_2Dtruss.py:393-411) */
  int32_t *status;  /* [B] or NULL: bit mask, 0 = ok.
                       TRUSS_STATUS_NOT_SPD (1): non-positive pivot (K not SPD: the reference would raise
                         numpy.linalg.LinAlgError from FEM_2Dtruss.py:337 or return garbage);
                       TRUSS_STATUS_OBS_TIMEOUT (2): TRUSS_F_EMIT_OBS only -- the wavefront that streams the observation
                         tensors gave up waiting for the step's results (bounded wait: the launch always drains); the
                         observation tensors of this env were NOT (completely) written by this call */

  /* observation tensors of the new design, written with TRUSS_F_EMIT_OBS (needs sec_out and max_up_out /
   * max_down_out); same contents and layouts as truss_obs_args_t below; any of them may be NULL */
  float *x_n;     /* [B][N][13]  state_data          truss2D_ENV.py:50-102 */
  float *A_s;     /* [B][N][N]                       :86-87 */
  float *A_n_ts;  /* [B][N][N]                       :92-97 */
  float *A_n_cs;  /* [B][N][N]                       :98-100 */
  float *nN_x_n;  /* [B][N][12]  state_data_not_norm :134-146 */
  float *nN_x_e;  /* [B][E][21]                      :148-169 */
} truss_step_args_t;

int truss_step(const truss_topo_t *t, const truss_step_args_t *args, void *stream);

/* Run `n_steps` chained steps without returning to the host in between: step s reads design
 * state from buffer (s & 1) and writes buffer ((s+1) & 1); actions for step s are taken at
 * a_geo + (s % n_action_sets) * B*N*2 (resp. a_topo ... *3).  args->y_in/sec_in = buffer 0, y_out/sec_out =
 * buffer 1; all other outputs are overwritten every step (every step writes them: the last step's survive).
 * This is the `for sol in Pf: for agent: _game_modify(...)` chain of run() (master…:211-260) with the actions known
 * up front.  Where truss_topo_persistent_rollout() says so the whole chain is ONE launch: every workgroup plays its
 * envs through all steps with the design state, the per-env constants and the topology tables resident in LDS and
 * the next step's actions prefetched during the solve (envs are independent: nothing crosses workgroups); otherwise
 * one launch per step.  Same results either way (bit for bit). */
int truss_rollout(const truss_topo_t *t, const truss_step_args_t *args, int32_t n_steps,
                  int32_t n_action_sets, void *stream);
/* 1 when truss_rollout runs plain decode steps of this topology as one persistent launch (0: one launch per step;
 * also 0 with the environment variable TRUSS_ROLLOUT_LAUNCHES=1, a diagnostic switch read at every call). */
int truss_topo_persistent_rollout(const truss_topo_t *t);

/* ---- observation tensors ---------------------------------------------------------------------
 * state_data + state_data_not_norm (truss2D_ENV.py:40-193) for the design/analysis a truss_step
 * just produced.  A_n, mask and nC_e are topology-static and are NOT written per env (the host
 * mirror builds them once).  All outputs float32, any of them may be NULL.
 */
typedef struct truss_obs_args {
  size_t struct_size;
  int32_t n_envs;
  uint32_t flags;
  const float *x;          /* [B][N] */
  const float *y;          /* [B][N]  (truss_step y_out) */
  const int32_t *sec;      /* [B][E] */
  const float *max_up;     /* [B][N] */
  const float *max_down;   /* [B][N] */
  const float *target;     /* [B][N] */
  const float *disp;       /* [B][N][2] */
  const float *q0;         /* [B][E] */
  const float *sr;         /* [B][E] */
  const uint8_t *comp;     /* [B][E] */
  const double *env_params; /* [B][TRUSS_NPARAM] */
  float *x_n;     /* [B][N][13]  normalised node features (truss2D_ENV.py:50-102) */
  float *A_s;     /* [B][N][N]   area / max area on edges (:86-87) */
  float *A_n_ts;  /* [B][N][N]   tension stress ratio on edges (:92-97) */
  float *A_n_cs;  /* [B][N][N]   compression stress ratio on edges (:98-100) */
  float *nN_x_n;  /* [B][N][12]  raw node features (:134-146) */
  float *nN_x_e;  /* [B][E][21]  raw element features (:148-169) */
} truss_obs_args_t;

int truss_obs(const truss_topo_t *t, const truss_obs_args_t *args, void *stream);

/* ---- Pareto front + hypervolume of a batch of small point sets --------------------------------
 * replaces: utils.simple_cull / simple_cull_final (train copy utils.py:11-217, test copies :220-403) and
 * utils.union_rectangles_fastest (:275-342), which the reward block of master_DDPG_truss2D_MO.run()
 * (:263-368) calls 5 + 6 times per archived solution, for B envs at once.
 *
 * Per env: n_points[b] rows [obj1, obj2, con1, con2] (n <= max_points <= 64).
 *   feasible  = not (con1 > 1 or con2 > 1)                                   (utils.py:18-22)
 *   front     = feasible rows no other feasible row beats in BOTH objectives (strict), identical rows once,
 *               sorted by obj1 (ties: obj2, then input order -- the reference's tie order is a set order)
 *   truncation (flags & TRUSS_FRONT_TRUNCATE, train copy): fronts longer than max_front keep both ends and
 *               the max_front-2 interior points of largest crowding distance, in obj1 order.  DEVIATION:
 *               the reference draws the interior points with random.sample (utils.py:118-123) and keeps
 *               them in draw order; a batched kernel has no Python RNG stream to follow.
 *   metrics   = [max_distance, dis_distance, p_norm_inv_cd, sum_distance, std_cd]  (utils.py:126-214)
 *   hv_front  = union_rectangles_fastest(front, ref_point); hv_all = the same over ALL input rows
 *               (the reference also calls it on raw archives).  Closed form over the x-sorted points
 *               instead of the sweep + segment tree: same area, summation order differs (<= 1e-12).
 * All pointers are device memory; outputs may be NULL to skip them.
 */
#define TRUSS_FRONT_TRUNCATE 0x1u
#define TRUSS_FRONT_MAXP 64

typedef struct truss_front_args {
  size_t struct_size;
  int32_t n_envs;
  int32_t max_points;      /* P: row stride of points / front_idx (<= TRUSS_FRONT_MAXP) */
  int32_t max_front;       /* MAX_FRONT (utils.py:6-7: 20 train, 50 test); used with TRUSS_FRONT_TRUNCATE */
  uint32_t flags;
  const double *points;    /* [B][P][4] */
  const int32_t *n_points; /* [B] */
  const double *ref_points;/* [B][2] or NULL = (1, 1) */
  int32_t *front_idx;      /* [B][P]  input row of the k-th front point, -1 beyond n_front */
  int32_t *n_front;        /* [B] */
  double *hv_front;        /* [B] */
  double *hv_all;          /* [B] */
  double *metrics;         /* [B][5] */
} truss_front_args_t;

int truss_front(const truss_front_args_t *args, void *stream);

/* ---- GCN neighbourhood aggregation for the actors' inference in batched rollouts -----------------
 * replaces (inference only): the `A @ (X W) + b` + activation half of spektral GCNConv as used by
 * truss2D_RL.multimodes_actor (truss2D_RL.py:49-120): out[b][i][c] = act(sum_j A[b][i][j] H[b][j][c] + bias[c])
 * with H = X W computed by the caller (one large GEMM, rocBLAS).  float32; n_nodes <= 64.
 * a_batch_stride = n_nodes * n_nodes for per-env adjacencies, 0 for one adjacency shared by the batch.
 * act: 0 none, 1 relu, 2 sigmoid.  Device pointers; `out` may alias `h`.
 */
int truss_gcn_aggregate(const float *adj, int64_t a_batch_stride, const float *h, const float *bias, float *out,
                        int32_t n_batch, int32_t n_nodes, int32_t n_channels, int32_t act, void *stream);

/* The same aggregation for graphs whose adjacency is known to be zero outside a fixed sparsity pattern -- the truss's own
 * connectivity plus the diagonal, which is the pattern of every node-graph adjacency the reference builds (A_n, A_s, A_n_ts,
 * A_n_cs: truss2D_ENV.py:40-193): out[b][i][c] = act(sum_k A[b][i][nbr[i][k]] H[b][nbr[i][k]][c] + bias[c]), k < k_nbr,
 * instead of a sum over all n_nodes columns (256-node trusses: 9 terms instead of 256).  `adj` stays the DENSE [n_nodes][n_nodes]
 * matrix (per env or shared, as above): only the listed entries are read.  nbr: int16 [n_nodes][k_nbr] device table, ascending
 * column indices per row, -1 = unused slot; k_nbr <= 16.  n_channels must be a multiple of 4, h / out / bias 16-byte aligned.
 * Any n_nodes <= 32767.  `out` must NOT alias `h` (rows are read by their neighbours' threads).
 */
int truss_gcn_aggregate_sparse(const float *adj, int64_t a_batch_stride, const int16_t *nbr, int32_t k_nbr, const float *h,
                               const float *bias, float *out, int32_t n_batch, int32_t n_nodes, int32_t n_channels, int32_t act,
                               void *stream);

/* ---- one whole GCN layer on the matrix cores -------------------------------------------------------------
 * replaces: spektral GCNConv as the reference's actors / critics call it (truss2D_RL.py:49-127; 13 layers per actor, 21 per
 * critic): out[b] = act(A[b] (X[b] W^T) + bias) for a batch of small graphs, float32 throughout.  Evaluated as ((A X) W^T): the
 * neighbourhood sum is applied to the input rows on their way into LDS, the product with W^T runs on MFMA (v_mfma_f32_32x32x2_f32,
 * float32 accumulation), bias / activation / accumulation are the epilogue -- H = X W never exists in HBM.  Same value as the
 * reference's order of operations up to float32 rounding (<= 2e-5 relative against the float32 PyTorch layer in the tests).
 *   x    [n_batch][n_nodes][k_in], rows x_row_stride floats apart (0 = k_in)
 *   adj  dense [n_nodes][n_nodes] per graph (a_batch_stride = n_nodes^2) or one for all (0)
 *   nbr  optional sparsity pattern as for truss_gcn_aggregate_sparse: int16 [n_nodes][k_nbr], -1 = unused, k_nbr <= 16;
 *        NULL = dense, n_nodes <= 64 (the Pareto graph)
 *   w    [c_out][k_in], k contiguous (the layout of torch.nn.Linear.weight), c_out <= 224; bias [c_out] or NULL
 *   out  [n_batch][n_nodes][c_out], rows out_row_stride floats apart (0 = c_out); must not alias x
 *   act  0 none, 1 relu, 2 sigmoid;  accumulate != 0: out += act(...) (the sum of the five second-level layers of the actor)
 * n_nodes <= 256.  Device pointers. */
typedef struct truss_gcn_layer_args {
  size_t struct_size;
  int32_t n_batch, n_nodes, k_in, c_out, act, accumulate, k_nbr, reserved;
  const float *x;
  int64_t x_row_stride;
  const float *adj;
  int64_t a_batch_stride;
  const int16_t *nbr;
  const float *w;
  const float *bias;
  float *out;
  int64_t out_row_stride;
  const uint16_t *w_bf16x3; /* NULL: the product runs on the float32 matrix cores.  Otherwise: `w` split into three bfloat16 terms by
                               truss_gcn_split_w ([3][224][kp], kp = k_in rounded up to 16); the product then runs on the bf16
                               matrix cores as six partial products with float32 accumulation ("bf16x3": every float32 operand is
                               split EXACTLY into three bf16 terms, the three dropped cross terms are below 2^-24 |a||b|) -- 2.7 x
                               fewer matrix-core cycles at float32 accuracy.  Hidden layers only: c_out 33..224, k_in % 4 == 0, x
                               16-byte aligned, <= 9 terms per row; other shapes are refused with this pointer set. */
} truss_gcn_layer_args_t;

int truss_gcn_layer(const truss_gcn_layer_args_t *args, void *stream);

/* A whole LEVEL of GCN layers in one launch: layers[i] as for truss_gcn_layer, for n_layers layers that do not depend on each other
 * -- the layers of one depth of the reference's actor / critic graphs (truss2D_RL.py:86-103, 116-133), of one network or of
 * several.  replaces: the forward passes of the MADDPG update (truss2D_RL.py:561-629: 19 network passes of 13 / 21 GCN layers at
 * batch 32), which are bound by their kernel count when run layer by layer.  Grid = (row tiles, layers, 32-column blocks).
 * float32 product only (w_bf16x3 = NULL, accumulate = 0), n_nodes <= 128 (dense: <= 64), any mix of shapes within one call.
 *   x_agg  NULL, or n_layers pointers (entries may be NULL): where given, the kernel also stores X' = A X of that layer,
 *          [n_batch * n_nodes][k_in] contiguous -- what the backward pass of the layer needs (dW = dZ^T X'). */
int truss_gcn_level(const truss_gcn_layer_args_t *layers, int32_t n_layers, float *const *x_agg, void *stream);

/* w [c_out <= 224][k_in] float32 -> w_bf16x3 [3][224][kp] bfloat16 bit patterns (kp = (k_in + 15) & ~15; rows >= c_out and columns >=
 * k_in are written as zeros: the layer kernel reads whole 224 x 16 slabs by LDS-DMA): the exact
 * three-term split the bf16x3 path of truss_gcn_layer reads.  Once per weight version; device pointers, 16-byte aligned output. */
int truss_gcn_split_w(const float *w, int32_t c_out, int32_t k_in, uint16_t *w_bf16x3, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRUSS_MI355_H */
