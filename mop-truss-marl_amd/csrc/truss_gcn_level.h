// truss_gcn_level.h -- a whole LEVEL of GCN layers in one launch (gfx950): out_i = act(A_i (X_i W_i^T) + b_i) for up to TGL_MAX layers
// that do not depend on each other -- the layers of one depth of the reference's actors / critics (truss2D_RL.py:49-127), of one
// network or of several (the three target actors, the three critics ...).  This is the forward of the MADDPG update (batch 32, i.e.
// 512 .. 1 536 rows per layer): there the layer-by-layer evaluation is bound by its kernel count, not by arithmetic, so the tiling
// is chosen for LATENCY, not for operand re-use as in truss_gcn.h: grid = (row tiles, layers, 32-column blocks), every workgroup
// (4 waves) computes 128 rows x 32 columns of ONE layer -- 924 workgroups for the 33 second-level layers of three critics.  Same
// evaluation order as truss_gcn_layer's float32 kernel ((A X) W^T, K slabs of 16 through LDS, v_mfma_f32_32x32x2_f32 with float32
// accumulation, bias / activation in the epilogue), same operand layout in LDS.  The column-block 0 workgroups also store X' = A X
// (optional): the backward pass needs it (dW = dZ^T X') and it exists here anyway.
#pragma once

#define TGL_MAX 24               // layers per launch (24 x 120 bytes of kernel arguments)

struct GcnLevelDev {
  GcnLayerDev l[TGL_MAX];
  float *xagg[TGL_MAX];          // [B * N][K] per layer, or nullptr
};

__global__ __launch_bounds__(256) void truss_gcn_level_kernel(const GcnLevelDev LV) {
  constexpr int MT = 128, NT = 256;
  extern __shared__ __attribute__((aligned(16))) char tg_smem[];
  const int layer = blockIdx.y, cb = blockIdx.z;
  const GcnLayerDev &P = LV.l[layer];
  const int N = P.N, Kn = P.Kn, K = P.K;
  const int g0 = blockIdx.x * P.GB, c0 = cb * 32;
  if (g0 >= P.B || c0 >= P.C) return;                      // (uniform: before any barrier)
  const int C = P.C - c0 < 32 ? P.C - c0 : 32;             // columns of this block
  const float *W = P.w + (long)c0 * K;
  float *xagg = cb == 0 ? LV.xagg[layer] : nullptr;
  const bool xv = P.x_vec != 0, wv = P.w_vec != 0;
  float *sXraw = (float *)tg_smem;                         // [MT][TG_LD]     raw input rows of a slab
  float *sXa = sXraw + MT * TG_LD;                         // [2][MT][TG_LD]  aggregated rows (MFMA A operand)
  float *sW = sXa + 2 * MT * TG_LD;                        // [2][32][TG_LD]  W slab, [col][k]
  float *sCoef = sW + 2 * 32 * TG_LD;                      // [MT][Kn]
  int16_t *sIdx = (int16_t *)(sCoef + MT * Kn);            // [N][Kn]         source node of term t (-1: none)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ng = (P.B - g0 < P.GB) ? P.B - g0 : P.GB;
  const int rows = ng * N;                                 // live rows of this tile (<= MT)
  const long row0 = (long)g0 * N;
  const int ar = tid >> 1, ak = (tid & 1) * 8;             // aggregation item: row ar, eight k's from ak
  const int ag = ar / N, an = ar - ag * N;
  const bool alive = ar < rows;

  for (int i = tid; i < N * Kn; i += NT) sIdx[i] = P.nbr ? P.nbr[i] : (int16_t)(i % Kn);
  for (int i = tid; i < rows * Kn; i += NT) {
    const int r = i / Kn, t = i - r * Kn, g = r / N, n = r - g * N;
    const int j = P.nbr ? (int)P.nbr[n * Kn + t] : t;
    sCoef[i] = j < 0 ? 0.0f : P.adj[(long)(g0 + g) * P.a_stride + (long)n * N + j];
  }

  tg_f4 rx[2], rw;
  auto load4 = [&](const float *src, int kk, bool vec) {
    tg_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (vec) {
      if (kk < K) v = *(const tg_f4 *)src;                 // k_in % 4 == 0: a chunk is whole or past the end
    } else {
      if (kk + 0 < K) v[0] = src[0];
      if (kk + 1 < K) v[1] = src[1];
      if (kk + 2 < K) v[2] = src[2];
      if (kk + 3 < K) v[3] = src[3];
    }
    return v;
  };
  auto fetch = [&](int k0) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int q = tid + c * NT, r = q >> 2, kk = k0 + (q & 3) * 4;
      tg_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (r < rows) v = load4(P.x + (row0 + r) * P.x_stride + kk, kk, xv);
      rx[c] = v;
    }
    const int col = tid >> 2, kk = k0 + (tid & 3) * 4;
    tg_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (tid < 128 && col < C) v = load4(W + (long)col * K + kk, kk, wv);
    rw = v;
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int q = tid + c * NT;
      *(tg_f4 *)(sXraw + (q >> 2) * TG_LD + (q & 3) * 4) = rx[c];
    }
    if (tid < 128) *(tg_f4 *)(sW + (buf * 32 + (tid >> 2)) * TG_LD + (tid & 3) * 4) = rw;
  };
  auto aggregate = [&](int buf, int k0) {
    tg_f4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0;
    if (alive) {
      const float *cfl = sCoef + ar * Kn;
      const int16_t *ix = sIdx + an * Kn;
      const float *base = sXraw + (ag * N) * TG_LD + ak;
      for (int t = 0; t < Kn; ++t) {
        const int j = ix[t];
        if (j < 0) continue;
        const float c = cfl[t];
        a0 += c * *(const tg_f4 *)(base + j * TG_LD);
        a1 += c * *(const tg_f4 *)(base + j * TG_LD + 4);
      }
    }
    float *dst = sXa + (buf * MT + ar) * TG_LD + ak;
    *(tg_f4 *)dst = a0;
    *(tg_f4 *)(dst + 4) = a1;
    if (xagg && alive) {
      const int kk = k0 + ak;
      float *xo = xagg + (row0 + ar) * (long)K + kk;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kk + j < K) xo[j] = a0[j];
        if (kk + 4 + j < K) xo[4 + j] = a1[j];
      }
    }
  };

  tg_f16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  const int mrow = wave * 32 + (lane & 31), mh = lane >> 5;
  auto mfma_slab = [&](int buf) {
    const float *pa = sXa + (buf * MT + mrow) * TG_LD + mh * 8;
    const float *pb = sW + (buf * 32 + (lane & 31)) * TG_LD + mh * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const tg_f4 xa = *(const tg_f4 *)(pa + 4 * h);
      const tg_f4 wb = *(const tg_f4 *)(pb + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q], wb[q], acc, 0, 0, 0);
    }
  };

  // slab s is multiplied while slab s + 1 is aggregated and slab s + 2 is in flight (the schedule of truss_gcn_layer_kernel)
  const int nslab = (K + TG_KS - 1) / TG_KS;
  fetch(0);
  stash(0);
  tg_lds_barrier();
  if (nslab > 1) fetch(TG_KS);
  aggregate(0, 0);
  tg_lds_barrier();
  if (nslab > 1) stash(1);
  for (int s = 0; s < nslab; ++s) {
    tg_lds_barrier();
    if (s + 2 < nslab) fetch((s + 2) * TG_KS);
    if (s + 1 < nslab) aggregate((s + 1) & 1, (s + 1) * TG_KS);
    mfma_slab(s & 1);
    tg_lds_barrier();
    if (s + 2 < nslab) stash(s & 1);
  }

  // epilogue: accumulator register i of a lane = row 8 (i / 4) + 4 (l / 32) + i % 4, column l % 32 of the 32 x 32 block
  const int act = P.act;
  const int rbase = wave * 32 + 4 * (lane >> 5);
  const long ostride = P.out_stride;
  const int col = lane & 31;
  const bool colok = col < C;
  const float bc = (P.bias && colok) ? P.bias[c0 + col] : 0.0f;
  float *po = P.out + (row0 + rbase) * ostride + c0 + (colok ? col : 0);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int dr = 8 * (i >> 2) + (i & 3);
    float v = acc[i] + bc;
    if (act == 1) v = fmaxf(v, 0.0f);
    else if (act == 2) v = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    if (colok && rbase + dr < rows) po[dr * ostride] = v;
  }
}

extern "C" int truss_gcn_level(const truss_gcn_layer_args_t *layers, int32_t n_layers, float *const *x_agg, void *stream) {
  if (n_layers < 0 || (n_layers > 0 && !layers)) return tb_fail(TRUSS_EINVAL, "truss_gcn_level: bad argument");
  for (int i = 0; i < n_layers; ++i) {
    const truss_gcn_layer_args_t *a = layers + i;
    if (a->struct_size != sizeof(truss_gcn_layer_args_t)) return tb_fail(TRUSS_EINVAL, "truss_gcn_layer_args_t size mismatch (ABI)");
    if (!a->x || !a->adj || !a->w || !a->out) return tb_fail(TRUSS_EINVAL, "truss_gcn_level: a required pointer is NULL");
    if (a->n_batch < 0 || a->n_nodes < 1 || a->k_in < 1 || a->c_out < 1 || a->act < 0 || a->act > 2)
      return tb_fail(TRUSS_EINVAL, "truss_gcn_level: bad sizes / act");
    if (a->accumulate || a->w_bf16x3) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_level: float32 product, no accumulation into out");
    if (a->c_out > 224) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_level: c_out <= 224");
    if (a->n_nodes > 128) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_level: n_nodes <= 128");
    if (a->nbr ? (a->k_nbr < 1 || a->k_nbr > 16) : a->n_nodes > 64)
      return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_level: a sparsity pattern of 1..16 terms per row, or a dense adjacency of at most 64 nodes");
    if (a->x == a->out) return tb_fail(TRUSS_EINVAL, "truss_gcn_level: out must not alias x");
  }
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n_layers; i0 += TGL_MAX) {
    const int nl = n_layers - i0 < TGL_MAX ? n_layers - i0 : TGL_MAX;
    GcnLevelDev LV;
    memset(&LV, 0, sizeof LV);
    size_t lds = 0;
    unsigned tiles = 0, cbs = 0;
    int live = 0;
    for (int i = 0; i < nl; ++i) {
      const truss_gcn_layer_args_t *a = layers + i0 + i;
      if (a->n_batch == 0) continue;
      GcnLayerDev &P = LV.l[live];
      P.x = a->x; P.adj = a->adj; P.nbr = a->nbr; P.w = a->w; P.bias = a->bias; P.out = a->out;
      P.x_stride = a->x_row_stride ? a->x_row_stride : a->k_in;
      P.out_stride = a->out_row_stride ? a->out_row_stride : a->c_out;
      P.a_stride = a->a_batch_stride;
      P.B = a->n_batch; P.N = a->n_nodes; P.K = a->k_in; P.C = a->c_out; P.act = a->act; P.accumulate = 0;
      P.Kn = a->nbr ? a->k_nbr : a->n_nodes;
      P.GB = 128 / a->n_nodes;
      P.x_vec = ((size_t)a->x % 16 == 0 && P.x_stride % 4 == 0 && a->k_in % 4 == 0) ? 1 : 0;
      P.w_vec = ((size_t)a->w % 16 == 0 && a->k_in % 4 == 0) ? 1 : 0;
      LV.xagg[live] = x_agg ? x_agg[i0 + i] : nullptr;
      const size_t b = tg_lds_bytes(4, 1, a->n_nodes, P.Kn);
      lds = b > lds ? b : lds;
      const unsigned t = (unsigned)((a->n_batch + P.GB - 1) / P.GB), c = (unsigned)((a->c_out + 31) / 32);
      tiles = t > tiles ? t : tiles;
      cbs = c > cbs ? c : cbs;
      ++live;
    }
    if (!live) continue;
    if (lds > 160 * 1024) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_level: tile does not fit the LDS");
    static TbLdsOptIn optin;
    if (int rc = optin.ensure((const void *)truss_gcn_level_kernel)) return rc;
    hipLaunchKernelGGL(truss_gcn_level_kernel, dim3(tiles, (unsigned)live, cbs), dim3(256), lds, st, LV);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("gcn level kernel launch failed: ") + hipGetErrorString(e));
  }
  return TRUSS_OK;
}
