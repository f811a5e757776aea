// truss_gcn_level.h -- a whole LEVEL of GCN layers in one launch (gfx950): out_i = act(A_i (X_i W_i^T) + b_i) for up to TGL_MAX layers
// that do not depend on each other -- the layers of one depth of the reference's actors / critics (truss2D_RL.py:49-127), of one
// network or of several (the three target actors, the three critics ...).  This is the forward of the MADDPG update (batch 32, i.e.
// 512 .. 1 536 rows per layer): there the layer-by-layer evaluation is bound by its kernel count, not by arithmetic, so the tiling
// is chosen for LATENCY, not for operand re-use as in truss_gcn.h: grid = (row tiles, layers, 32-column blocks [+ K slabs of X']), every workgroup
// (4 waves) computes 128 rows x 32 columns of ONE layer -- 924 workgroups for the 33 second-level layers of three critics.  Same
// evaluation order as the reference, A (X W^T): K slabs of 64 through LDS, v_mfma_f32_32x32x2_f32 with float32 accumulation, the 128 x 32
// product tile back through LDS, neighbourhood sums from there (six times fewer LDS reads than summing the input rows in each of the
// seven column blocks), bias / activation, 64-byte row pieces to HBM.  Extra slices of the grid store X' = A X where asked for (one
// workgroup per 128 rows x 64 k's): the backward pass needs it (dW = dZ^T X').  A workgroup's time is a chain of memory round trips:
// its whole K range is requested in one go (see the request sequence below and tools/experiments/README.md for the versions before).
#pragma once

#define TGL_MAX 24               // layers per launch (24 x 120 bytes of kernel arguments)

struct GcnLevelDev {
  GcnLayerDev l[TGL_MAX];
  float *xagg[TGL_MAX];          // [B * N][K] per layer, or nullptr
};

#define TGL_KS 64                // k's per slab of the product, and per workgroup of the X' slices of the grid
#define TGL_LD 68                // floats per LDS row of a slab (64 + 4 padding: rows 272 bytes apart)
#define TGL_HLD 36               // floats per LDS row of the 128 x 32 product tile (32 + 4 padding: rows 144 bytes apart)
#define TGL_SLABS 4              // slabs of the product (three requested up front, the fourth behind the first): k_in <= 256

#ifdef TRUSS_GCN_STAMPS   // diagnostic build: shader-clock stamps of block (0, 0, 0) / thread 0 (tools/gcn_level_probe.py)
__device__ unsigned long long g_level_stamps[8];
#define TGL_T(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_level_stamps[i] = clock64(); } while (0)
extern "C" int truss_debug_level_stamps(unsigned long long *out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_level_stamps), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#else
#define TGL_T(i)
#endif

#ifndef TGL_WAVES_PER_EU
#define TGL_WAVES_PER_EU 2
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TGL_WAVES_PER_EU))) void truss_gcn_level_kernel(const GcnLevelDev LV, int cbs) {
  constexpr int MT = 128, NT = 256;
  extern __shared__ __attribute__((aligned(16))) char tg_smem[];
  TGL_T(0);
  const int layer = blockIdx.y, cb = blockIdx.z;
  const GcnLayerDev &P = LV.l[layer];
  const int N = P.N, Kn = P.Kn, K = P.K;
  const int g0 = blockIdx.x * P.GB, c0 = cb * 32;
  // z < cbs: the 32-column block cb of the layer's output.  z >= cbs: the k's [64 (z - cbs), + 64) of X' = A X for the backward pass
  // (only where asked for).
  const bool xrole = cb >= cbs;
  const int xk0 = (cb - cbs) * TGL_KS;
  float *xagg = LV.xagg[layer];
  if (g0 >= P.B || (xrole ? (xagg == nullptr || xk0 >= K) : c0 >= P.C)) return;          // (uniform: before any barrier)
  const int C = P.C - c0 < 32 ? P.C - c0 : 32;             // columns of this block
  const float *W = P.w + (long)c0 * K;
  const bool xv = P.x_vec != 0, wv = P.w_vec != 0;
  float *sT = (float *)tg_smem;                            // the product tile [MT][TGL_HLD], or (X' workgroups) the input slab [MT][TGL_LD]
  float *sCoef = sT + (MT + 32) * TGL_LD;                  // [MT][Kn]      (only without the register path below)
  int16_t *sIdx = (int16_t *)(sCoef + MT * Kn);            // [N][Kn]       source node of term t (-1: none)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ng = (P.B - g0 < P.GB) ? P.B - g0 : P.GB;
  const int rows = ng * N;                                 // live rows of this tile (<= MT)
  const long row0 = (long)g0 * N;
  const int ar = tid >> 1, ah = tid & 1;                   // neighbourhood sums: row ar, half ah of the columns (k's) at hand
  const int ag = ar / N, an = ar - ag * N;
  const bool alive = ar < rows;
  const int gbase = alive ? ag * N : 0;                    // first row of this thread's graph in the tile

  // ---- operands from HBM / L2, ALL in flight at once: a workgroup's time is a chain of memory round trips (~1.5 us each), so the
  // whole K range is requested up front -- coalesced (16 lanes per 256-byte row piece), into registers -- and then walked slab by
  // slab through LDS.  X' workgroups: their one 128 x 64 slab.
  // A 16-byte piece (k's kk .. kk + 3 of a row).  VEC (k_in % 4 == 0, 16-byte alignment: the piece is whole or past the end): one
  // load from a clamped address and NOTHING else -- the pieces are masked when they go to LDS; a select (or a branch) next to the
  // load would make every slab wait for its own data before the next one is requested.  Otherwise element-wise guarded loads (the
  // 13 / 2 / 3 / 4-feature input layers: one short slab).
  // Three slabs (192 k's) are requested up front; the fourth takes the registers of the first as soon as that one is in LDS and is
  // in flight during the arithmetic of the first three.
  tg_f4 rx[3][8], rw[3][2];
  auto piece_ok = [&](int q, int k0, int lim) { return (q >> 4) < lim && k0 + (q & 15) * 4 < K; };
  auto guarded4 = [&](const float *src, int kk, bool ok) {
    tg_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ok) {
      if (kk + 0 < K) v[0] = src[0];
      if (kk + 1 < K) v[1] = src[1];
      if (kk + 2 < K) v[2] = src[2];
      if (kk + 3 < K) v[3] = src[3];
    }
    return v;
  };
  const bool vec = xv && wv;
  auto request_vec = [&](int sl, tg_f4 *px, tg_f4 *pw) {   // (slabs past the end too: clamped to the safe address, masked later)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int q = tid + c * NT, r = q >> 4, kk = sl * TGL_KS + (q & 15) * 4;
      px[c] = *(const tg_f4 *)(piece_ok(q, sl * TGL_KS, rows) ? P.x + (row0 + r) * P.x_stride + kk : P.x);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int q = tid + c * NT, col = q >> 4, kk = sl * TGL_KS + (q & 15) * 4;
      pw[c] = *(const tg_f4 *)(piece_ok(q, sl * TGL_KS, C) ? W + (long)col * K + kk : P.w);
    }
  };
  auto request_any = [&](int sl, tg_f4 *px, tg_f4 *pw) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int q = tid + c * NT, r = q >> 4, kk = sl * TGL_KS + (q & 15) * 4;
      px[c] = guarded4(P.x + (row0 + (r < rows ? r : 0)) * P.x_stride + kk, kk, r < rows);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int q = tid + c * NT, col = q >> 4, kk = sl * TGL_KS + (q & 15) * 4;
      pw[c] = guarded4(W + (long)(col < C ? col : 0) * K + kk, kk, col < C);
    }
  };
  if (xrole) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int q = tid + c * NT, r = q >> 4, kk = xk0 + (q & 15) * 4;
      const float *src = P.x + (row0 + (r < rows ? r : 0)) * P.x_stride + kk;
      if (xv) rx[0][c] = *(const tg_f4 *)(piece_ok(q, xk0, rows) ? src : P.x);
      else rx[0][c] = guarded4(src, kk, r < rows);
    }
  } else if (vec) {
    request_vec(0, rx[0], rw[0]);
    request_vec(1, rx[1], rw[1]);
    request_vec(2, rx[2], rw[2]);
  } else {
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
      if (sl * TGL_KS < K) request_any(sl, rx[sl], rw[sl]);
  }
  TGL_T(1);

  // A dense adjacency of at most 16 nodes (the node graphs of small_roof; every graph of the update at batch 32 is dense): the row of A
  // stays in 16 registers and the gathers below have no table look-ups.  Otherwise: tables in LDS.
  const bool fast = !P.nbr && Kn <= 16;                    // uniform
  float cf[16];
  if (fast) {
    const float *arow = P.adj + (alive ? (long)(g0 + ag) * P.a_stride + (long)an * N : 0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const bool use = alive && t < Kn;
      const float c = arow[use ? t : 0];                   // always a valid address: no branch, the 16 loads are in flight together
      cf[t] = use ? c : 0.0f;
    }
  } else {
    for (int i = tid; i < N * Kn; i += NT) sIdx[i] = P.nbr ? P.nbr[i] : (int16_t)(i % Kn);
    for (int i = tid; i < rows * Kn; i += NT) {
      const int r = i / Kn, t = i - r * Kn, g = r / N, n = r - g * N;
      const int j = P.nbr ? (int)P.nbr[n * Kn + t] : t;
      sCoef[i] = j < 0 ? 0.0f : P.adj[(long)(g0 + g) * P.a_stride + (long)n * N + j];
    }
  }
  // (product workgroups: the row of A waits in LDS while the staging registers of the K loop are live)
  float *sCf = sT + (MT + 32) * TGL_LD;                     // [MT][16], the place of the tables
  if (fast && !xrole && ah == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *(tg_f4 *)(sCf + ar * 16 + 4 * q) = tg_f4{cf[4 * q], cf[4 * q + 1], cf[4 * q + 2], cf[4 * q + 3]};
  }
  TGL_T(2);

  // sum_t A[row][t] * tile[graph row t][off .. off + 15] (four 4-float pieces); tile rows `ld` floats apart
  auto gather = [&](const float *tile, int ld, int off, tg_f4 *a) {
    if (fast) {
      // terms past Kn have coefficient 0 and re-read the graph's first row: no branch, all reads independent
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float *src = tile + (gbase + (t < Kn ? t : 0)) * ld + off;
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += cf[t] * *(const tg_f4 *)(src + 4 * q);
      }
    } else if (alive) {
      const float *cfl = sCoef + ar * Kn;
      const int16_t *ix = sIdx + an * Kn;
      for (int t = 0; t < Kn; ++t) {
        const int j = ix[t];
        if (j < 0) continue;
        const float c = cfl[t];
        const float *src = tile + (gbase + j) * ld + off;
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += c * *(const tg_f4 *)(src + 4 * q);
      }
    }
  };

  if (xrole) {
    // X'[row][xk0 + 32 ah .. + 31] = sum_t A[row][t] X[graph row t][same k's], in two passes of 16 k's
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int q = tid + c * NT;
      *(tg_f4 *)(sT + (q >> 4) * TGL_LD + (q & 15) * 4) = piece_ok(q, xk0, rows) ? rx[0][c] : tg_f4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    tg_lds_barrier();
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
      tg_f4 a[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = tg_f4{0.0f, 0.0f, 0.0f, 0.0f};
      gather(sT, TGL_LD, ah * 32 + hp * 16, a);
      if (alive) {
        const int kk = xk0 + ah * 32 + hp * 16;
        float *xo = xagg + (row0 + ar) * (long)K + kk;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (kk + 4 * q + j < K) xo[4 * q + j] = a[q][j];
      }
    }
    return;
  }

  // ---- H = X W^T of this tile, slab by slab: registers -> LDS (X [128][64 + 4], W [32][64 + 4]) | barrier | per 16 k's: lane (row |
  // column l % 32, half h = l / 32) reads k = 8 h .. 8 h + 7 of its row / column; MFMA q of half-step hh multiplies k = 4 hh + q
  // (lanes of half 0) and 8 + 4 hh + q (half 1) -- the order of a sum over k is free | barrier.  No memory round trip in here.
  const int mrow = wave * 32 + (lane & 31), mh = lane >> 5;
  float *sW = sT + MT * TGL_LD;                             // (the tables, if any, start behind it: see the host side)
  tg_f16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
  for (int sl = 0; sl < TGL_SLABS; ++sl)
    if (sl * TGL_KS < K) {
      constexpr int NB = 3;
      const int b = sl % NB;                                // (compile-time after unrolling)
      if (sl) tg_lds_barrier();                             // everybody is done with the previous slab
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int q = tid + c * NT;
        *(tg_f4 *)(sT + (q >> 4) * TGL_LD + (q & 15) * 4) = piece_ok(q, sl * TGL_KS, rows) ? rx[b][c] : tg_f4{0.0f, 0.0f, 0.0f, 0.0f};
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int q = tid + c * NT;
        *(tg_f4 *)(sW + (q >> 4) * TGL_LD + (q & 15) * 4) = piece_ok(q, sl * TGL_KS, C) ? rw[b][c] : tg_f4{0.0f, 0.0f, 0.0f, 0.0f};
      }
      if (sl == 0 && 3 * TGL_KS < K) {                      // the fourth slab, into the registers just emptied
        if (vec) request_vec(3, rx[0], rw[0]);
        else request_any(3, rx[0], rw[0]);
      }
      tg_lds_barrier();
      const float *pa = sT + mrow * TGL_LD + mh * 8;
      const float *pb = sW + (lane & 31) * TGL_LD + mh * 8;
#pragma unroll
      for (int u = 0; u < TGL_KS / 16; ++u)
        if (sl * TGL_KS + 16 * u < K) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const tg_f4 xa = *(const tg_f4 *)(pa + 16 * u + 4 * hh);
            const tg_f4 wb = *(const tg_f4 *)(pb + 16 * u + 4 * hh);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q], wb[q], acc, 0, 0, 0);
          }
        }
    }
  TGL_T(3);

  tg_lds_barrier();                                         // every wave is done reading the last slab: the tile goes over it
  // H (accumulator register i of a lane = row 8 (i / 4) + 4 (l / 32) + i % 4, column l % 32) -> LDS; then
  // out[row][c0 + 16 ah .. + 15] = act(sum_t A[row][t] H[graph row t][...] + bias): 64 contiguous bytes per thread
  {
    const int rbase = wave * 32 + 4 * (lane >> 5), col = lane & 31;
#pragma unroll
    for (int i = 0; i < 16; ++i) sT[(rbase + 8 * (i >> 2) + (i & 3)) * TGL_HLD + col] = acc[i];
  }
  tg_lds_barrier();
  TGL_T(4);
  if (fast) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const tg_f4 v = *(const tg_f4 *)(sCf + ar * 16 + 4 * q);
      cf[4 * q] = v[0], cf[4 * q + 1] = v[1], cf[4 * q + 2] = v[2], cf[4 * q + 3] = v[3];
    }
  }
  tg_f4 a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = tg_f4{0.0f, 0.0f, 0.0f, 0.0f};
  gather(sT, TGL_HLD, ah * 16, a);
  TGL_T(5);
  if (!alive) return;
  const int act = P.act;
  const int cl = ah * 16;                                   // first column (within the block) of this thread
  float *po = P.out + (row0 + ar) * P.out_stride + c0 + cl;
  const bool whole = C == 32 && (P.out_stride & 3) == 0 && ((size_t)P.out & 15) == 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    tg_f4 v = a[q];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cl + 4 * q + j;
      float t = v[j] + ((P.bias && c < C) ? P.bias[c0 + c] : 0.0f);
      if (act == 1) t = fmaxf(t, 0.0f);
      else if (act == 2) t = __builtin_amdgcn_rcpf(1.0f + __expf(-t));
      v[j] = t;
    }
    if (whole) {
      *(tg_f4 *)(po + 4 * q) = v;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (cl + 4 * q + j < C) po[4 * q + j] = v[j];
    }
  }
#ifdef TRUSS_GCN_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  TGL_T(6);
#endif
}

extern "C" int truss_gcn_level(const truss_gcn_layer_args_t *layers, int32_t n_layers, float *const *x_agg, void *stream) {
  if (n_layers < 0 || (n_layers > 0 && !layers)) return tb_fail(TRUSS_EINVAL, "truss_gcn_level: bad argument");
  for (int i = 0; i < n_layers; ++i) {
    const truss_gcn_layer_args_t *a = layers + i;
    if (a->struct_size != sizeof(truss_gcn_layer_args_t)) return tb_fail(TRUSS_EINVAL, "truss_gcn_layer_args_t size mismatch (ABI)");
    if (a->n_batch == 0) continue;                           // an empty layer: nothing to read, nothing to write
    if (!a->x || !a->adj || !a->w || !a->out) return tb_fail(TRUSS_EINVAL, "truss_gcn_level: a required pointer is NULL");
    if (a->n_batch < 0 || a->n_nodes < 1 || a->k_in < 1 || a->c_out < 1 || a->act < 0 || a->act > 2)
      return tb_fail(TRUSS_EINVAL, "truss_gcn_level: bad sizes / act");
    if (a->accumulate || a->w_bf16x3) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_level: float32 product, no accumulation into out");
    if (a->c_out > 224) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_level: c_out <= 224");
    if (a->n_nodes > 128) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_level: n_nodes <= 128");
    if (a->k_in > TGL_SLABS * TGL_KS) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_level: k_in <= 256");
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
    unsigned tiles = 0, cbs = 0, xslabs = 0;
    int live = 0;
    size_t tables = 0;
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
      const size_t tb = (!a->nbr && P.Kn <= 16) ? 0 : sizeof(float) * 128 * (size_t)P.Kn + sizeof(int16_t) * (size_t)a->n_nodes * P.Kn;
      tables = tb > tables ? tb : tables;
      if (LV.xagg[live]) {
        const unsigned xs = (unsigned)((a->k_in + TGL_KS - 1) / TGL_KS);
        xslabs = xs > xslabs ? xs : xslabs;
      }
      const unsigned t = (unsigned)((a->n_batch + P.GB - 1) / P.GB), c = (unsigned)((a->c_out + 31) / 32);
      tiles = t > tiles ? t : tiles;
      cbs = c > cbs ? c : cbs;
      ++live;
    }
    if (!live) continue;
    // a slab of X (128 x 68 floats) and of W (32 x 68) -- the product tile (128 x 36) goes over it afterwards -- and the tables
    if (tables < sizeof(float) * 128 * 16) tables = sizeof(float) * 128 * 16;      // (or the parked rows of A)
    lds = sizeof(float) * (128 + 32) * TGL_LD + tables;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_level: tile does not fit the LDS");
    static TbLdsOptIn optin;
    if (int rc = optin.ensure((const void *)truss_gcn_level_kernel)) return rc;
    hipLaunchKernelGGL(truss_gcn_level_kernel, dim3(tiles, (unsigned)live, cbs + xslabs), dim3(256), lds, st, LV, (int)cbs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("gcn level kernel launch failed: ") + hipGetErrorString(e));
  }
  return TRUSS_OK;
}
