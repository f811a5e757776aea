// truss_gcn.h -- one whole GCN layer on the matrix cores (gfx950): out[b] = act(A[b] (X[b] W^T) + bias), Spektral GCNConv as the
// reference's actors / critics use it (truss2D_RL.py:49-127: 13 layers per actor, 21 per critic, hidden width 200).
//
// Evaluated as ((A X) W^T): the neighbourhood sum is applied to the INPUT rows on their way into LDS (a row of A has <= 9
// non-zeros on a truss -- the sparsity pattern of every node-graph adjacency the reference builds -- or N <= 64 dense entries on the
// Pareto graph), the product with W^T runs on v_mfma_f32_32x32x2_f32 with float32 accumulation, bias / activation (/ accumulation into
// `out`) sit in the epilogue.  H = X W never exists in HBM, and the [M, C] result is written exactly once.
//
// Tiling.  A workgroup of NW wavefronts owns MT = 32 NW rows = whole graphs (GB = MT / N graphs of N nodes; the 256-node class takes
// NW = 8, everything else NW = 4) and ALL c_out <= 32 CB columns (CB = 7 column blocks for the hidden width 200, 1 for the action
// heads): wave w accumulates the 32 x (32 CB) block of its 32 rows in 16 CB accumulator registers.  K is streamed in slabs of 16:
//   global -> registers (next slab: X rows as 64-byte pieces, W rows as 64-byte pieces of nn.Linear's [c_out][k_in] layout)
//   registers -> LDS (raw X slab, W slab [col][16 k])          | barrier
//   aggregate: X'[r][k] = sum_t coef[r][t] * Xraw[row(r, t)][k]  (VALU, from LDS to LDS)      | barrier
//   MFMA: lane (row | col = l % 32, half h = l / 32) holds k = 8 h .. 8 h + 7 of its row / column (two ds_read_b128); step s of the
//         eight multiplies k = s (lanes of half 0) and k = 8 + s (half 1): the order of a sum over k is free.
// W (160 KB at 200 x 200) is re-read by every workgroup from L2; X is read once, `out` written once.
#pragma once

typedef float tg_f4 __attribute__((ext_vector_type(4)));
typedef float tg_f16 __attribute__((ext_vector_type(16)));

struct GcnLayerDev {
  const float *x;
  const float *adj;
  const int16_t *nbr;
  const float *w;
  const float *bias;
  float *out;
  long x_stride, out_stride, a_stride;     // floats between rows of x / out; between graphs of adj (0: one adjacency for all)
  int B, N, K, C, act, accumulate;
  int Kn;                                  // terms per row: k_nbr (pattern) or N (dense)
  int GB;                                  // graphs per tile
  int x_vec, w_vec;                        // 16-byte loads allowed (alignment + k_in % 4 == 0)
};

#define TG_KS 16                 // K slab
#define TG_LD 20                 // floats per LDS row of a slab (16 + 4 padding: rows 80 bytes apart)

// Workgroup barrier for data handed over THROUGH LDS: waits for this wave's LDS operations only.  __syncthreads() also waits for
// vmcnt(0), i.e. for the global loads of the slab after next that are meant to stay in flight across the barrier.
__device__ __forceinline__ void tg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef TRUSS_GCN_STAMPS   // diagnostic build: cycles of the slab loop's sections, block 0 / wave 0 (tools/gcn_stamps.py)
__device__ unsigned long long g_gcn_stamps[8];
#define TG_T(v) unsigned long long v = clock64()
#else
#define TG_T(v)
#endif

#define TG_KREG 9                // neighbourhood terms per row held in registers (a truss node joins <= 8 elements: <= 9 terms with the diagonal)

// VEC: x and w are 16-byte aligned with k_in % 4 == 0 -- every 4-float chunk of a slab is either whole or past the end, so the loads
// are branch-free 16-byte loads from clamped addresses; otherwise (the 13-feature input layers) element-wise guarded loads.
template <int NW, int CB, bool VEC>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2))) void truss_gcn_layer_kernel(const GcnLayerDev P) {
  constexpr int MT = 32 * NW, NT = 64 * NW, WROWS = 32 * CB;
  extern __shared__ __attribute__((aligned(16))) char tg_smem[];
  float *sXraw = (float *)tg_smem;                         // [MT][TG_LD]       raw input rows of a slab
  float *sXa = sXraw + MT * TG_LD;                         // [2][MT][TG_LD]    aggregated rows (MFMA A operand)
  float *sW = sXa + 2 * MT * TG_LD;                        // [2][WROWS][TG_LD] W slab, [col][k]
  float *sCoef = sW + 2 * WROWS * TG_LD;                   // [MT][Kn]          (only read from LDS when Kn > TG_KREG)
  int16_t *sIdx = (int16_t *)(sCoef + MT * P.Kn);          // [N][Kn]           source node of term t (-1: none)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = P.N, Kn = P.Kn, K = P.K, C = P.C;
  const int g0 = blockIdx.x * P.GB;
  const int ng = (P.B - g0 < P.GB) ? P.B - g0 : P.GB;
  const int rows = ng * N;                                 // live rows of this tile (<= MT)
  const long row0 = (long)g0 * N;

  // ---- aggregation item of this thread: row ar, eight k's from ak (MT * 2 items, NT = 2 MT threads) ----
  const int ar = tid >> 1, ak = (tid & 1) * 8;
  const int ag = ar / N, an = ar - ag * N;                 // graph in the tile, node
  const bool alive = ar < rows;
  const bool kreg = Kn <= TG_KREG;                         // uniform
  float cf[TG_KREG];
  uint32_t cj[(TG_KREG + 3) / 4] = {};                     // source nodes of the terms, one byte each (n_nodes <= 256)
  const int abase = ((alive ? ag * N : 0)) * TG_LD + ak;   // float offset of the graph's first row (this thread's eight k's) in sXraw
  if (kreg) {
    // branch-free, so that the nine table reads and then the nine coefficient reads are in flight together (a dead row / a term
    // past Kn reads entry 0 and is masked afterwards)
    int jj[TG_KREG];
#pragma unroll
    for (int t = 0; t < TG_KREG; ++t) {
      const bool use = alive && t < Kn;
      jj[t] = P.nbr ? (int)P.nbr[use ? an * Kn + t : 0] : t;
      jj[t] = use ? jj[t] : -1;
    }
#pragma unroll
    for (int t = 0; t < TG_KREG; ++t) {
      const int j = jj[t];
      const float c = P.adj[j < 0 ? 0 : (long)(g0 + ag) * P.a_stride + (long)an * N + j];
      cf[t] = j < 0 ? 0.0f : c;
      cj[t >> 2] |= (uint32_t)(j < 0 ? 0 : j) << (8 * (t & 3));          // a missing term re-reads a live row with coefficient 0
    }
  } else {
    for (int i = tid; i < N * Kn; i += NT) sIdx[i] = P.nbr ? P.nbr[i] : (int16_t)(i % Kn);
    for (int i = tid; i < rows * Kn; i += NT) {
      const int r = i / Kn, t = i - r * Kn, g = r / N, n = r - g * N;
      const int j = P.nbr ? (int)P.nbr[n * Kn + t] : t;
      sCoef[i] = j < 0 ? 0.0f : P.adj[(long)(g0 + g) * P.a_stride + (long)n * N + j];
    }
  }

  // ---- staging: X slab = MT * 4 chunks of 4 floats (chunk q -> row q / 4, k (q % 4) * 4), two per thread; W slab = WROWS * 4 chunks ----
  constexpr int XC = MT * 4 / NT;                          // = 2
  constexpr int WC = (WROWS * 4 + NT - 1) / NT;
  tg_f4 rx[XC], rw[WC];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      const int q = tid + c * NT, r = q >> 2, kk = k0 + (q & 3) * 4;
      tg_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (VEC) {
        const bool ok = r < rows && kk < K;
        const tg_f4 ld = *(const tg_f4 *)(ok ? P.x + (row0 + r) * P.x_stride + kk : P.x);    // always a valid address, no branch
        v = ok ? ld : v;
      } else if (r < rows) {
        const float *src = P.x + (row0 + r) * P.x_stride + kk;
        if (kk + 0 < K) v[0] = src[0];
        if (kk + 1 < K) v[1] = src[1];
        if (kk + 2 < K) v[2] = src[2];
        if (kk + 3 < K) v[3] = src[3];
      }
      rx[c] = v;
    }
#pragma unroll
    for (int c = 0; c < WC; ++c) {
      const int q = tid + c * NT, col = q >> 2, kk = k0 + (q & 3) * 4;
      tg_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (VEC) {
        const bool ok = col < C && kk < K;                 // (q >= WROWS * 4 implies col >= 32 CB >= C)
        const tg_f4 ld = *(const tg_f4 *)(ok ? P.w + (long)col * K + kk : P.w);
        v = ok ? ld : v;
      } else if (col < C) {
        const float *src = P.w + (long)col * K + kk;
        if (kk + 0 < K) v[0] = src[0];
        if (kk + 1 < K) v[1] = src[1];
        if (kk + 2 < K) v[2] = src[2];
        if (kk + 3 < K) v[3] = src[3];
      }
      rw[c] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      const int q = tid + c * NT;
      *(tg_f4 *)(sXraw + (q >> 2) * TG_LD + (q & 3) * 4) = rx[c];
    }
#pragma unroll
    for (int c = 0; c < WC; ++c) {
      const int q = tid + c * NT;
      if (q < WROWS * 4) *(tg_f4 *)(sW + (buf * WROWS + (q >> 2)) * TG_LD + (q & 3) * 4) = rw[c];
    }
  };
  // X'[ar][ak .. ak + 7] of the slab in sXraw -> sXa[buf]
  auto aggregate = [&](int buf) {
    tg_f4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0;
    if (kreg) {
      // all TG_KREG terms, no branch (terms past Kn have coefficient 0 and re-read a live row): the loop body below must stay ONE
      // basic block so that the scheduler can spread these reads and multiply-adds between the MFMAs of the previous slab
#pragma unroll
      for (int t0 = 0; t0 < TG_KREG; t0 += 3) {
        tg_f4 v0[3], v1[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int tt = t0 + t;
          const float *src = sXraw + abase + (int)((cj[tt >> 2] >> (8 * (tt & 3))) & 255u) * TG_LD;
          v0[t] = *(const tg_f4 *)src;
          v1[t] = *(const tg_f4 *)(src + 4);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          a0 += cf[t0 + t] * v0[t];
          a1 += cf[t0 + t] * v1[t];
        }
      }
    } else if (alive) {
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
  };

  tg_f16 acc[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cb][i] = 0.0f;
  const int mrow = wave * 32 + (lane & 31), mh = lane >> 5;
  // the 8 MFMA steps of a slab in two halves of 4 (B operands of a half: CB x 4 registers); accumulators round-robin over the column
  // blocks, so that consecutive MFMAs are independent
  auto mfma_slab = [&](int buf) {
    const float *pa = sXa + (buf * MT + mrow) * TG_LD + mh * 8;
    const float *pb = sW + (buf * WROWS + (lane & 31)) * TG_LD + mh * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const tg_f4 xa = *(const tg_f4 *)(pa + 4 * h);
      tg_f4 wb[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) wb[cb] = *(const tg_f4 *)(pb + cb * 32 * TG_LD + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q], wb[cb][q], acc[cb], 0, 0, 0);
    }
  };

  // ---- pipeline: slab s is multiplied while slab s + 1 is aggregated (same wave: the MFMAs run in the matrix pipe, the gathers on
  // VALU / LDS) and slab s + 2 is in flight from HBM / L2.  Two barriers per slab, both next to the (short) register -> LDS copy.
  const int nslab = (K + TG_KS - 1) / TG_KS;
  fetch(0);
  stash(0);
  tg_lds_barrier();
  if (nslab > 1) fetch(TG_KS);
  aggregate(0);
  tg_lds_barrier();
  if (nslab > 1) stash(1);
  if (VEC && kreg) {       // (the element-wise loader of the 13-feature input layers keeps the plain loop: one slab, nothing to overlap)
    // Straight-line body: work on slabs past the end is harmless (fetch returns zeros, the extra aggregate / stash fill buffers
    // nobody reads), so nothing in it is conditional and the whole body is one scheduling region.  Source order = the order the
    // memory model allows: the MFMA operand reads of a half, then the gathers of the NEXT slab (reads of sXraw only) between that
    // half's MFMAs, and the one LDS write of the gathered row behind the last operand read (the compiler cannot tell sXa[0] from
    // sXa[1]).  The sched_group_barrier sequence asks for one MFMA, then a few VALU / LDS / VMEM instructions, 56 times (a wave
    // issues in order: what is not placed BETWEEN the MFMAs waits behind them).
    auto gather = [&](int t_lo, int t_hi, tg_f4 &a0, tg_f4 &a1) {
#pragma unroll
      for (int tt = t_lo; tt < t_hi; ++tt) {
        const float *src = sXraw + abase + (int)((cj[tt >> 2] >> (8 * (tt & 3))) & 255u) * TG_LD;
        a0 += cf[tt] * *(const tg_f4 *)src;
        a1 += cf[tt] * *(const tg_f4 *)(src + 4);
      }
    };
#ifdef TRUSS_GCN_STAMPS
    unsigned long long acc_t[4] = {0, 0, 0, 0};
    const unsigned long long t_loop0 = clock64();
#endif
    for (int s = 0; s < nslab; ++s) {
      TG_T(ta);
      tg_lds_barrier();                                     // X'[s] (and, s > 0: raw slab s + 1, W slab s + 1) are in LDS
      TG_T(tb);
      const int buf = s & 1;
      fetch((s + 2) * TG_KS);
      const float *pa = sXa + (buf * MT + mrow) * TG_LD + mh * 8;
      const float *pb = sW + (buf * WROWS + (lane & 31)) * TG_LD + mh * 8;
      tg_f4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const tg_f4 xa = *(const tg_f4 *)(pa + 4 * h);
        tg_f4 wb[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) wb[cb] = *(const tg_f4 *)(pb + cb * 32 * TG_LD + 4 * h);
        gather(h == 0 ? 0 : (TG_KREG + 1) / 2, h == 0 ? (TG_KREG + 1) / 2 : TG_KREG, a0, a1);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q], wb[cb][q], acc[cb], 0, 0, 0);
      }
      {
        float *dst = sXa + ((buf ^ 1) * MT + ar) * TG_LD + ak;
        *(tg_f4 *)dst = a0;
        *(tg_f4 *)(dst + 4) = a1;
      }
#pragma unroll
      for (int i = 0; i < 8 * CB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                     // VALU
        if (i % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // LDS read
        if (i % 8 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
      }
      TG_T(tc);
      tg_lds_barrier();                                     // everybody is done with sXraw (slab s + 1) and with sW / sXa [s & 1]
      TG_T(td);
      stash(s & 1);                                        // raw slab s + 2 -> sXraw, W slab s + 2 -> sW[s & 1]
#ifdef TRUSS_GCN_STAMPS
      __builtin_amdgcn_s_waitcnt(0);
      const unsigned long long te = clock64();
      acc_t[0] += tb - ta; acc_t[1] += tc - tb; acc_t[2] += td - tc; acc_t[3] += te - td;
#endif
    }
#ifdef TRUSS_GCN_STAMPS
    if (blockIdx.x == 0 && tid == 0) {
      for (int i = 0; i < 4; ++i) g_gcn_stamps[i] = acc_t[i];
      g_gcn_stamps[4] = clock64() - t_loop0;
      g_gcn_stamps[5] = nslab;
    }
#endif
  } else {
    for (int s = 0; s < nslab; ++s) {
      tg_lds_barrier();
      if (s + 2 < nslab) fetch((s + 2) * TG_KS);
      if (s + 1 < nslab) aggregate((s + 1) & 1);
      mfma_slab(s & 1);
      tg_lds_barrier();
      if (s + 2 < nslab) stash(s & 1);
    }
  }

  // ---- epilogue: accumulator register i of a lane = row 8 (i / 4) + 4 (l / 32) + i % 4, column l % 32 of the 32 x 32 block.
  // Per column block: (accumulate: the 16 old values, from clamped addresses, in flight together) -> bias, activation -> 16
  // predicated stores; 32 lanes write 128 contiguous bytes of a row.  `act` / `accumulate` are uniform.
  const int act = P.act;
  const bool accum = P.accumulate != 0;
  const int rbase = wave * 32 + 4 * (lane >> 5);            // row of accumulator register 0; register i: + 8 (i / 4) + i % 4
  const long ostride = P.out_stride;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int col = cb * 32 + (lane & 31);
    const bool colok = col < C;
    const float bc = (P.bias && colok) ? P.bias[col] : 0.0f;
    float *po = P.out + (row0 + rbase) * ostride + (colok ? col : 0);
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 8) {
      float old[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) old[i] = 0.0f;
      if (accum) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int dr = 8 * ((i0 + i) >> 2) + ((i0 + i) & 3);
          old[i] = po[(rbase + dr < rows ? dr : 0 - rbase) * ostride];      // a dead row re-reads the tile's first row
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int dr = 8 * ((i0 + i) >> 2) + ((i0 + i) & 3);
        float v = acc[cb][i0 + i] + bc;
        if (act == 1) v = fmaxf(v, 0.0f);
        else if (act == 2) v = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        v += old[i];
        if (colok && rbase + dr < rows) po[dr * ostride] = v;
      }
    }
  }
}

// ============================================================================================================================
// The same layer with the product on the BF16 matrix cores at float32 accuracy ("bf16x3"): every float32 operand is split EXACTLY
// into three bfloat16 terms (x = x0 + x1 + x2, truncation: 8 + 8 + 8 significant bits), and a . b is evaluated as the six partial
// products  a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)  by v_mfma_f32_32x32x16_bf16 with float32 accumulation; the three
// dropped products are below 2^-24 |a| |b|, the size of a float32 rounding.  Six bf16 MFMAs of K = 16 replace eight fp32 MFMAs of
// K = 2 per 16 k: 42 x 32 cycles instead of 56 x 64 per slab and wave (the fp32 matrix pipe is the bound of the kernel above: it
// runs at 95 % busy while two workgroups share a CU, at a power-limited ~1.45 GHz).  W arrives already split ([3][c_out][kp] bf16,
// truss_gcn_split_w, once per weight version); X' is split by the thread that gathers it.  LDS images are [term][row][16 k] bf16 =
// 32-byte rows, the two 16-byte halves of a row swapped in every other group of 8 rows (conflict-free ds_read_b128 without padding).
// Envelope: the hidden layers (c_out 33..224, k_in % 4 == 0, x 16-byte aligned, <= 9 terms per row); everything else takes the
// float32 kernel.
typedef __bf16 tg_bf8 __attribute__((ext_vector_type(8)));
typedef uint32_t tg_u4 __attribute__((ext_vector_type(4)));

// the next bf16 term of eight floats: packs their upper halves (two per dword, element 2 i in the low half) and leaves the
// (exact) remainders in place -- term after term, so that only the eight remainders stay live
__device__ __forceinline__ tg_u4 tg_split_term(float (&r)[8], bool last) {
  tg_u4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t a = __float_as_uint(r[2 * i]), b = __float_as_uint(r[2 * i + 1]);
    out[i] = __builtin_amdgcn_perm(b, a, 0x07060302u);
    if (!last) {
      r[2 * i] = r[2 * i] - __uint_as_float(a & 0xffff0000u);           // exact
      r[2 * i + 1] = r[2 * i + 1] - __uint_as_float(b & 0xffff0000u);
    }
  }
  return out;
}

#define TG_CP 224                // rows of the split-weight image: c_out padded to the kernel's 7 column blocks (zero rows)

// w [c_out][k_in] float32 -> ws [3][TG_CP][kp] bf16 (kp = k_in rounded up to 16; rows >= c_out and columns >= k_in are zero)
__global__ __launch_bounds__(256) void truss_gcn_split_w_kernel(const float *__restrict__ w, uint16_t *__restrict__ ws, int C, int K, int KP) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= TG_CP * KP) return;
  const int c = i / KP, k = i - c * KP;
  const float x = (c < C && k < K) ? w[(long)c * K + k] : 0.0f;
  const uint32_t u0 = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(u0);
  const uint32_t u1 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(u1);
  ws[i] = (uint16_t)(u0 >> 16);
  ws[(long)TG_CP * KP + i] = (uint16_t)(u1 >> 16);
  ws[2L * TG_CP * KP + i] = (uint16_t)(__float_as_uint(r2) >> 16);
}

typedef __attribute__((address_space(3))) void tg_lds_void;
typedef __attribute__((address_space(1))) const void tg_glob_void;

// KT: neighbourhood terms per row that are gathered (6: the grid trusses -- a node joins at most five others; 9: any pattern the
// register path takes).  Terms past k_nbr have coefficient 0, but every one of them is two 16-byte LDS reads per slab and thread.
template <int NW, int KT>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2))) void truss_gcn_layer_bf3_kernel(const GcnLayerDev P, const uint16_t *__restrict__ ws, int KP) {
  constexpr int MT = 32 * NW, NT = 64 * NW, CB = 7, WROWS = 32 * CB;
  static_assert(WROWS == TG_CP, "split-weight image rows");
  extern __shared__ __attribute__((aligned(16))) char tg_smem[];
  float *sXraw = (float *)tg_smem;                                   // [MT][TG_LD] float32 raw input rows of a slab
  char *sXs = tg_smem + MT * TG_LD * 4;                              // [2][3][MT][32 B]    split aggregated rows (MFMA A operand)
  char *sWs = sXs + 2 * 3 * MT * 32;                                 // [2][3][WROWS][32 B] split W slab, [col][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = P.N, Kn = P.Kn, K = P.K, C = P.C;
  const int g0 = blockIdx.x * P.GB;
  const int ng = (P.B - g0 < P.GB) ? P.B - g0 : P.GB;
  const int rows = ng * N;
  const long row0 = (long)g0 * N;

  const int ar = tid >> 1, ah = tid & 1, ak = ah * 8;
  const int ag = ar / N, an = ar - ag * N;
  const bool alive = ar < rows;
  float cf[KT];
  uint32_t cj[(KT + 3) / 4] = {};
  const int abase = ((alive ? ag * N : 0)) * TG_LD + ak;
  {
    int jj[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const bool use = alive && t < Kn;
      jj[t] = P.nbr ? (int)P.nbr[use ? an * Kn + t : 0] : t;
      jj[t] = use ? jj[t] : -1;
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int j = jj[t];
      const float c = P.adj[j < 0 ? 0 : (long)(g0 + ag) * P.a_stride + (long)an * N + j];
      cf[t] = j < 0 ? 0.0f : c;
      cj[t >> 2] |= (uint32_t)(j < 0 ? 0 : j) << (8 * (t & 3));
    }
  }

  // staging.  X slab: two 16-byte chunks per thread through registers, as in the float32 kernel.  W slab (3 terms x 224 cols x two
  // 16-byte halves = 21 wave-instructions of 1 KB): LDS-DMA (global_load_lds_dwordx4) -- lane i of instruction j fetches the chunk
  // that belongs at LDS position 64 j + i of the (swizzled) image, no registers, no ds_write.  Every wave issues exactly WD of
  // them (instruction numbers past the last repeat it), so that "at most WD vector-memory operations outstanding" means "everything
  // older than the last W slab has landed".
  constexpr int XC = MT * 4 / NT;                                    // = 2
  constexpr int WI = 3 * WROWS * 2 / 64;                             // = 21 wave-instructions per slab
  constexpr int WD = (WI + NW - 1) / NW;                             // per wave: 6 (NW = 4) or 3 (NW = 8)
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  tg_f4 rx[XC];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      const int q = tid + c * NT, r = q >> 2, kk = k0 + (q & 3) * 4;
      const bool ok = r < rows && kk < K;
      const tg_f4 ld = *(const tg_f4 *)(ok ? P.x + (row0 + r) * P.x_stride + kk : P.x);
      const tg_f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
      rx[c] = ok ? ld : z;
    }
  };
  auto stash_x = [&]() {
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      const int q = tid + c * NT;
      *(tg_f4 *)(sXraw + (q >> 2) * TG_LD + (q & 3) * 4) = rx[c];
    }
  };
  auto dma_w = [&](int k0, int buf) {                                // slabs past the end of K read the zero padding of the last one
    const int kc = k0 < KP ? k0 : KP - TG_KS;                        // (a slab past the end is never multiplied: any valid address)
#pragma unroll
    for (int c = 0; c < WD; ++c) {
      int j = uwave + c * NW;
      j = j < WI ? j : WI - 1;
      const int t = j / (WI / 3), rem = (j % (WI / 3)) * 64 + lane;  // term; position inside the term's [224][2] image
      const int col = rem >> 1, h = (rem & 1) ^ ((col >> 3) & 1);
      const uint16_t *src = ws + ((long)t * TG_CP + col) * KP + kc + h * 8;
      __builtin_amdgcn_global_load_lds((tg_glob_void *)src, (tg_lds_void *)(sWs + (buf * 3 * WROWS * 2 + j * 64) * 16), 16, 0, 0);
    }
  };
  auto gather = [&](int t_lo, int t_hi, tg_f4 &a0, tg_f4 &a1) {
#pragma unroll
    for (int tt = t_lo; tt < t_hi; ++tt) {
      const float *src = sXraw + abase + (int)((cj[tt >> 2] >> (8 * (tt & 3))) & 255u) * TG_LD;
      a0 += cf[tt] * *(const tg_f4 *)src;
      a1 += cf[tt] * *(const tg_f4 *)(src + 4);
    }
  };
  auto put_split = [&](int buf, const tg_f4 a0, const tg_f4 a1) {    // this thread's eight k of row ar, three terms
    float r[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
    for (int t = 0; t < 3; ++t)
      *(tg_u4 *)(sXs + (((buf * 3 + t) * MT + ar) * 2 + (ah ^ ((ar >> 3) & 1))) * 16) = tg_split_term(r, t == 2);
  };

  tg_f16 acc[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cb][i] = 0.0f;
  const int mrow = wave * 32 + (lane & 31), mh = lane >> 5, mcol = lane & 31;

  // Order of the vector-memory operations of a wave (they complete in order): [W 0] [X 0] [W 1] [X 1] | loop s: [X s+2] [W s+2].
  // Waiting for the registers of X s+2 (stash_x, end of iteration s) therefore implies that W s+1 has landed; the explicit
  // "vmcnt(WD)" in front of the top barrier says the same thing for the reader of this code.
  const int nslab = (K + TG_KS - 1) / TG_KS;
  dma_w(0, 0);
  fetch(0);
  stash_x();
  dma_w(TG_KS, 1);
  if constexpr (WD == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  tg_lds_barrier();
  fetch(TG_KS);
  {
    tg_f4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0;
    gather(0, KT, a0, a1);
    put_split(0, a0, a1);
  }
  tg_lds_barrier();
  stash_x();
#ifdef TRUSS_GCN_STAMPS
  unsigned long long acc_t[4] = {0, 0, 0, 0};
  const unsigned long long t_loop0 = clock64();
#endif
  for (int s = 0; s < nslab; ++s) {
    TG_T(ta);
    if constexpr (WD == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    tg_lds_barrier();                                                // X'[s], raw slab s + 1 and W slab s are in LDS
    TG_T(tb);
    const int buf = s & 1;
    fetch((s + 2) * TG_KS);
    tg_bf8 xa[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) xa[t] = *(const tg_bf8 *)(sXs + (((buf * 3 + t) * MT + mrow) * 2 + (mh ^ ((mrow >> 3) & 1))) * 16);
    tg_f4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int col = cb * 32 + mcol;
      tg_bf8 wb[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) wb[t] = *(const tg_bf8 *)(sWs + (((buf * 3 + t) * WROWS + col) * 2 + (mh ^ ((col >> 3) & 1))) * 16);
      if (cb * 2 < KT) gather(cb * 2, cb * 2 + 2 < KT ? cb * 2 + 2 : KT, a0, a1);   // two terms between the MFMAs of each of the first column blocks
      // small partial products first
      acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[2], wb[0], acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1], wb[1], acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], wb[2], acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1], wb[0], acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], wb[1], acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], wb[0], acc[cb], 0, 0, 0);
    }
    put_split(buf ^ 1, a0, a1);                                      // behind the last operand read of this slab
    TG_T(tc);
    tg_lds_barrier();                                                // everybody is done with sXraw (slab s + 1) and with sWs / sXs [buf]
    TG_T(td);
    dma_w((s + 2) * TG_KS, buf);                                     // W slab s + 2 -> sWs[buf]
    stash_x();                                                       // raw slab s + 2 -> sXraw
#ifdef TRUSS_GCN_STAMPS
    const unsigned long long te = clock64();
    acc_t[0] += tb - ta; acc_t[1] += tc - tb; acc_t[2] += td - tc; acc_t[3] += te - td;
#endif
  }
#ifdef TRUSS_GCN_STAMPS
  if (blockIdx.x == 0 && tid == 0) {
    for (int i = 0; i < 4; ++i) g_gcn_stamps[i] = acc_t[i];
    g_gcn_stamps[4] = clock64() - t_loop0;
    g_gcn_stamps[5] = nslab;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // no LDS-DMA may still be in flight when the workgroup ends

  const int act = P.act;
  const bool accum = P.accumulate != 0;
  const int rbase = wave * 32 + 4 * (lane >> 5);
  const long ostride = P.out_stride;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int col = cb * 32 + (lane & 31);
    const bool colok = col < C;
    const float bc = (P.bias && colok) ? P.bias[col] : 0.0f;
    float *po = P.out + (row0 + rbase) * ostride + (colok ? col : 0);
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 8) {
      float old[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) old[i] = 0.0f;
      if (accum) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int dr = 8 * ((i0 + i) >> 2) + ((i0 + i) & 3);
          old[i] = po[(rbase + dr < rows ? dr : 0 - rbase) * ostride];
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int dr = 8 * ((i0 + i) >> 2) + ((i0 + i) & 3);
        float v = acc[cb][i0 + i] + bc;
        if (act == 1) v = fmaxf(v, 0.0f);
        else if (act == 2) v = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        v += old[i];
        if (colok && rbase + dr < rows) po[dr * ostride] = v;
      }
    }
  }
}

static size_t tg_lds_bytes(int NW, int CB, int N, int Kn) {
  const size_t MT = 32 * NW;
  size_t b = sizeof(float) * (MT * TG_LD + 2 * MT * TG_LD + 2 * (size_t)(32 * CB) * TG_LD + MT * (size_t)Kn);
  b += sizeof(int16_t) * (size_t)N * Kn;
  return (b + 15) & ~(size_t)15;
}

extern "C" int truss_gcn_layer(const truss_gcn_layer_args_t *a, void *stream) {
  if (!a) return tb_fail(TRUSS_EINVAL, "truss_gcn_layer: NULL argument");
  if (a->struct_size != sizeof(truss_gcn_layer_args_t)) return tb_fail(TRUSS_EINVAL, "truss_gcn_layer_args_t size mismatch (ABI)");
  if (!a->x || !a->adj || !a->w || !a->out) return tb_fail(TRUSS_EINVAL, "truss_gcn_layer: a required pointer is NULL");
  if (a->n_batch < 0 || a->n_nodes < 1 || a->k_in < 1 || a->c_out < 1 || a->act < 0 || a->act > 2)
    return tb_fail(TRUSS_EINVAL, "truss_gcn_layer: bad sizes / act");
  if (a->c_out > 224) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_layer: c_out <= 224 (the reference's hidden width is 200)");
  if (a->n_nodes > 256) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_layer: n_nodes <= 256");
  if (a->nbr ? (a->k_nbr < 1 || a->k_nbr > 16) : a->n_nodes > 64)
    return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_layer: a sparsity pattern of 1..16 terms per row, or a dense adjacency of at most 64 nodes");
  if (a->x == a->out) return tb_fail(TRUSS_EINVAL, "truss_gcn_layer: out must not alias x");
  if (a->n_batch == 0) return TRUSS_OK;
  GcnLayerDev P;
  P.x = a->x; P.adj = a->adj; P.nbr = a->nbr; P.w = a->w; P.bias = a->bias; P.out = a->out;
  P.x_stride = a->x_row_stride ? a->x_row_stride : a->k_in;
  P.out_stride = a->out_row_stride ? a->out_row_stride : a->c_out;
  P.a_stride = a->a_batch_stride;
  P.B = a->n_batch; P.N = a->n_nodes; P.K = a->k_in; P.C = a->c_out; P.act = a->act; P.accumulate = a->accumulate ? 1 : 0;
  P.Kn = a->nbr ? a->k_nbr : a->n_nodes;
  const int NW = a->n_nodes > 128 ? 8 : 4, MT = 32 * NW;
  P.GB = MT / a->n_nodes;
  P.x_vec = ((size_t)a->x % 16 == 0 && P.x_stride % 4 == 0 && a->k_in % 4 == 0) ? 1 : 0;
  P.w_vec = ((size_t)a->w % 16 == 0 && a->k_in % 4 == 0) ? 1 : 0;
  const int CB = a->c_out <= 32 ? 1 : 7;
  hipStream_t st = (hipStream_t)stream;
  if (a->w_bf16x3) {
    // product on the bf16 matrix cores at float32 accuracy (see truss_gcn_layer_bf3_kernel); shapes outside its envelope are an error,
    // not a silent change of arithmetic: the caller asked for this path by passing split weights
    if (CB != 7 || !(P.x_vec) || P.Kn > TG_KREG || ((size_t)a->w_bf16x3 & 15) != 0)
      return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_layer: the bf16x3 path takes c_out 33..224, k_in % 4 == 0, 16-byte aligned x / split weights, <= 9 terms per row");
    const int KP = (a->k_in + 15) & ~15;
    const size_t MTb = 32 * (size_t)NW;
    const size_t lds3 = MTb * TG_LD * 4 + 2 * 3 * MTb * 32 + 2 * 3 * 224 * 32;
    const unsigned grid3 = (unsigned)((a->n_batch + P.GB - 1) / P.GB);
#define TG_LAUNCH3(nw, kt)                                                                                                  \
  do {                                                                                                                      \
    static TbLdsOptIn optin;                                                                                                \
    if (int rc = optin.ensure((const void *)truss_gcn_layer_bf3_kernel<nw, kt>)) return rc;                                 \
    hipLaunchKernelGGL((truss_gcn_layer_bf3_kernel<nw, kt>), dim3(grid3), dim3(64 * nw), lds3, st, P, a->w_bf16x3, KP);      \
  } while (0)
    if (NW == 4) { if (P.Kn <= 6) TG_LAUNCH3(4, 6); else TG_LAUNCH3(4, 9); }
    else { if (P.Kn <= 6) TG_LAUNCH3(8, 6); else TG_LAUNCH3(8, 9); }
#undef TG_LAUNCH3
    hipError_t e3 = hipGetLastError();
    if (e3 != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("gcn layer (bf16x3) kernel launch failed: ") + hipGetErrorString(e3));
    return TRUSS_OK;
  }
  const size_t lds = tg_lds_bytes(NW, CB, a->n_nodes, P.Kn);
  if (lds > 160 * 1024) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_layer: tile does not fit the LDS");
  const unsigned grid = (unsigned)((a->n_batch + P.GB - 1) / P.GB);
#define TG_LAUNCH(nw, cb, vec)                                                                              \
  do {                                                                                                      \
    static TbLdsOptIn optin;                                                                                \
    if (int rc = optin.ensure((const void *)truss_gcn_layer_kernel<nw, cb, vec>)) return rc;                \
    hipLaunchKernelGGL((truss_gcn_layer_kernel<nw, cb, vec>), dim3(grid), dim3(64 * nw), lds, st, P);       \
  } while (0)
  const bool vec = P.x_vec && P.w_vec;
  if (NW == 4 && CB == 7) { if (vec) TG_LAUNCH(4, 7, true); else TG_LAUNCH(4, 7, false); }
  else if (NW == 4) { if (vec) TG_LAUNCH(4, 1, true); else TG_LAUNCH(4, 1, false); }
  else if (CB == 7) { if (vec) TG_LAUNCH(8, 7, true); else TG_LAUNCH(8, 7, false); }
  else { if (vec) TG_LAUNCH(8, 1, true); else TG_LAUNCH(8, 1, false); }
#undef TG_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("gcn layer kernel launch failed: ") + hipGetErrorString(e));
  return TRUSS_OK;
}

#ifdef TRUSS_GCN_STAMPS
extern "C" int truss_debug_gcn_stamps(unsigned long long *out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gcn_stamps), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int truss_gcn_split_w(const float *w, int32_t c_out, int32_t k_in, uint16_t *w_bf16x3, void *stream) {
  if (!w || !w_bf16x3 || c_out < 1 || k_in < 1) return tb_fail(TRUSS_EINVAL, "truss_gcn_split_w: bad argument");
  if (c_out > TG_CP) return tb_fail(TRUSS_EUNSUPPORTED, "truss_gcn_split_w: c_out <= 224");
  const int KP = (k_in + 15) & ~15;
  const int n = TG_CP * KP;
  hipLaunchKernelGGL(truss_gcn_split_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, w_bf16x3, c_out, k_in, KP);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tb_fail(TRUSS_EHIP, std::string("gcn split kernel launch failed: ") + hipGetErrorString(e));
  return TRUSS_OK;
}
