// K13: the dense contractions of the hot path on the gfx950 matrix cores.
//
// Replaces the MatMul ops of tf.layers.dense (reference layers/dnn.py:57-62, model/deepfm.py:84-104,
// layers/mmoe.py:53-58, keras Dense in layers/keras/blocks.py:84-89, Cross layers/keras/interaction.py:
// 249-286) and their two gradients.  Until now these went through torch.mm (rocBLAS picked fp32 tiles
// that reach ~14 TFLOP/s on these skinny shapes and, with atomics allowed, are not run-to-run
// deterministic).
//
// MI355X design
//   * fp32 path: v_mfma_f32_32x32x2_f32 - exact f32 (bit-equal to an fmaf chain in k order), 64 FLOP/clk
//     per SIMD = the chip's 157 TFLOP/s f32 peak; needed because the north-star bar is 1e-4 on fp32 logits.
//   * bf16 path (config 3, "bf16 dense + fp32 emb"): fp32 operands in HBM are rounded to bf16 (RNE) while
//     being staged into LDS - no separate cast pass, fp32 master weights - and contracted with
//     v_mfma_f32_32x32x16_bf16 into fp32 accumulators.
//   * one workgroup = 4 waves = a 64x64 output tile, each wave one 32x32 accumulator (16 VGPRs); operands
//     staged through LDS in k-major layout so that the per-lane fragment reads are conflict-free
//     (row strides 65 / 68 floats: MI355X_MICROARCH.md LDS banking); next tile's global loads are issued
//     before the MFMAs of the current one (register double buffer).
//   * three operand layouts without transposes in HBM: NN (forward x.W), NT (dx = dy.W^T), TN (dW = x^T.dy).
//     TN contracts over the batch (K = 4096) into a small M x N: split-K over blockIdx.z into a
//     workspace + a deterministic reduce (no atomics) keeps >= 512 workgroups in flight.
//   * epilogue: + bias[col], optional accumulate into C (gradients land directly in the flat gradient
//     buffer: no autograd add kernels), optional per-tile column statistics for BatchNorm.
#include "er_gemm_core.h"

namespace er {

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(kBlock)
gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kOpTile];  // [stage][A | B][64][SK]
  gemm_f32_block<A_KC, B_KC>(g, blockIdx.x, blockIdx.z, lds);
}

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(kBlock)
gemm_f32_bn_bwd_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kOpTile];
  gemm_f32_block<A_KC, B_KC, true>(g, blockIdx.x, 0, lds);
}

// ... with the DCN-v2 cross layer's elementwise part in the epilogue (er_gemm_f32_cross)
template <bool A_KC, bool B_KC, int XEPI>
__global__ void __launch_bounds__(kBlock)
gemm_f32_cross_kernel(GemmArgs g, er_gemm_epilogue xe) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kOpTile];
  gemm_f32_block<A_KC, B_KC, false, XEPI>(g, blockIdx.x, 0, lds, false, &xe);
}

// DIN's first attention layer on the generated [q, h, q - h, q * h] operand (DinGen): forward / weight gradient (DIN = 1),
// input gradient reduced to dh and dq partials in the epilogue (DIN = 2)
template <bool A_KC, bool B_KC, int DIN>
__global__ void __launch_bounds__(kBlock)
gemm_f32_din_kernel(GemmArgs g, DinGen d) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kOpTile];
  gemm_f32_block<A_KC, B_KC, false, 0, DIN>(g, blockIdx.x, blockIdx.z, lds, false, nullptr, &d);
}

// the A operand is a BatchNorm'd layer's z, normalised + activated while staging (BnA; er_gemm_f32_bn_a)
__global__ void __launch_bounds__(kBlock)
gemm_f32_bna_kernel(GemmArgs g, BnA b) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kOpTile];
  __shared__ __attribute__((aligned(16))) float coef[4 * kBnAMaxK];
  bna_load_coef(b, g.K, coef);
  __syncthreads();
  gemm_f32_block<true, false, false, 0, 3>(g, blockIdx.x, 0, lds, false, nullptr, nullptr, &b, coef);
}

// A TALL projection onto 1 - 4 columns (DIN's attention score layer: [B x L, 32] -> [B x L, 1], reference
// model/multi_tower_din.py:80-84 - the last tf.layers.dense of the attention DNN): 64-column MFMA tiles did 1/64 useful work
// and took the generic fetch path ([K, 1] is not 16-byte addressable): 20.7 us for 26 MB.  Here lpr lanes share a row
// (16 bytes of it each per trip: a wave reads whole contiguous rows), each lane accumulates its k's in order, the lanes'
// partial sums are added by a fixed butterfly.  b.mean != nullptr: the row is the producing layer's z and its BatchNorm +
// activation is applied on the way (BnA; y kept for the backward) - bn_act_one's arithmetic.
template <int NC>
__global__ void __launch_bounds__(kBlock)
gemv_bna_kernel(const float* __restrict__ x, int ldx, int M, int K, BnA b, const float* __restrict__ W, int ldw,
                const float* __restrict__ bias, float* __restrict__ C, int ldc, int lpr) {
  constexpr int R = 4;  // rows per thread, their loads in flight together
  const int rpb = kBlock / lpr;
  const int lr = threadIdx.x % lpr;
  const int r0 = (static_cast<int>(blockIdx.x) * R) * rpb + threadIdx.x / lpr;
  const bool bn = b.mean != nullptr;
  float acc[R][NC];
#pragma unroll
  for (int u = 0; u < R; ++u)
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[u][n] = 0.f;
  for (int c = lr; c < K / 4; c += lpr) {
    const int k = 4 * c;
    f32x4v xv[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      int row = r0 + u * rpb;
      row = row < M ? row : M - 1;  // (clamped, never stored)
      xv[u] = *reinterpret_cast<const f32x4v*>(x + static_cast<int64_t>(row) * ldx + k);
    }
    float w[4][NC];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NC; ++n) w[j][n] = W[static_cast<int64_t>(k + j) * ldw + n];
    float mu[4], is[4], ga[4], be[4];
    if (bn) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mu[j] = b.mean[k + j];
        is[j] = b.invstd[k + j];
        ga[j] = b.gamma ? b.gamma[k + j] : 1.f;
        be[j] = b.beta ? b.beta[k + j] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      f32x4v v = xv[u];
      if (bn) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = (v[j] - mu[j]) * is[j];
          t = t * ga[j] + be[j];
          if (b.act == ER_ACT_RELU) t = t > 0.f ? t : 0.f;
          v[j] = t;
        }
        const int row = r0 + u * rpb;
        if (b.y && row < M) *reinterpret_cast<f32x4v*>(b.y + static_cast<int64_t>(row) * b.ldy + k) = v;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NC; ++n) acc[u][n] = acc[u][n] + v[j] * w[j][n];
    }
  }
#pragma unroll
  for (int u = 0; u < R; ++u) {
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      float a = acc[u][n];
      for (int m = 1; m < lpr; m <<= 1) a = a + __shfl_xor(a, m, 64);
      acc[u][n] = a;
    }
    const int row = r0 + u * rpb;
    if (lr == 0 && row < M) {
#pragma unroll
      for (int n = 0; n < NC; ++n) C[static_cast<int64_t>(row) * ldc + n] = acc[u][n] + (bias ? bias[n] : 0.f);
    }
  }
}

// The weight gradient of a TALL projection onto 1 - 4 columns, dW [K][N] = x^T . dz over rows >> K (DIN's attention score
// layer: x [B x L, 32], dz [B x L, 1]): in the grouped launch this problem had a [rows, 1] operand no 16-byte load can take, so
// its ~100 k-splits of 64 k-tiles each ran the one-tile-at-a-time generic loop and the whole launch waited for them.  Here a
// workgroup owns a chunk of rows; lpr lanes share a row (16 bytes of x each), every lane keeps the products of ITS k's with
// the row's dz in registers, the lanes that hold the same k's are added through LDS in a fixed order -> partial [chunk][K][N];
// er_colsum_partials_multi-style reduction over the chunks follows (wgrad_narrow_reduce_kernel).
template <int NC>
__global__ void __launch_bounds__(kBlock)
wgrad_narrow_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dz, int lddz, int rows, int K,
                            int rows_per_chunk, int lpr, float* __restrict__ partial, int with_bias) {
  constexpr int R = 4;  // rows per lane and trip, their loads in flight together
  __shared__ float sm[kBlock * 4 * NC];
  const int rpb = kBlock / lpr;            // row lanes
  const int lr = threadIdx.x % lpr, rl = threadIdx.x / lpr;
  const int r_lo = static_cast<int>(blockIdx.x) * rows_per_chunk;
  int r_hi = r_lo + rows_per_chunk;
  r_hi = r_hi < rows ? r_hi : rows;
  for (int c0 = 0; c0 < K / 4; c0 += lpr) {  // (K / 4 > lpr: the k groups in rounds)
    const int c = c0 + lr;
    const bool live = c < K / 4;
    const int k = 4 * (live ? c : 0);
    float acc[4][NC];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NC; ++n) acc[j][n] = 0.f;
    for (int r0 = r_lo + rl; r0 < r_hi; r0 += R * rpb) {
      f32x4v xv[R];
      float dv[R][NC];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        int r = r0 + u * rpb;
        r = r < r_hi ? r : r_hi - 1;  // (clamped; its products are dropped below)
        xv[u] = *reinterpret_cast<const f32x4v*>(x + static_cast<int64_t>(r) * ldx + k);
#pragma unroll
        for (int n = 0; n < NC; ++n) dv[u][n] = dz[static_cast<int64_t>(r) * lddz + n];
      }
#pragma unroll
      for (int u = 0; u < R; ++u) {
        if (r0 + u * rpb < r_hi) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NC; ++n) acc[j][n] = acc[j][n] + xv[u][j] * dv[u][n];
        }
      }
    }
    __syncthreads();  // (the previous round's readers are done)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NC; ++n) sm[(rl * lpr + lr) * 4 * NC + j * NC + n] = acc[j][n];
    __syncthreads();
    // element e = (lr', j, n) of this round: the row lanes' values added in row-lane order
    for (int e = threadIdx.x; e < lpr * 4 * NC; e += kBlock) {
      const int l2 = e / (4 * NC), jn = e % (4 * NC);
      const int c2 = c0 + l2;
      if (c2 < K / 4) {
        float t = 0.f;
        for (int q = 0; q < rpb; ++q) t = t + sm[(q * lpr + l2) * 4 * NC + jn];
        const int kk = 4 * c2 + jn / NC, n = jn % NC;
        partial[(static_cast<int64_t>(blockIdx.x) * (K + with_bias) + kk) * NC + n] = t;
      }
    }
  }
  if (with_bias) {
    // row K of the chunk's record: the column sums of dz over the chunk (the bias gradient of the projection): every thread
    // its rows (stride kBlock, four in flight), then the threads in order
    float bs[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) bs[n] = 0.f;
    for (int r0 = r_lo + threadIdx.x; r0 < r_hi; r0 += R * kBlock) {
      float dv[R][NC];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        int r = r0 + u * kBlock;
        r = r < r_hi ? r : r_hi - 1;
#pragma unroll
        for (int n = 0; n < NC; ++n) dv[u][n] = dz[static_cast<int64_t>(r) * lddz + n];
      }
#pragma unroll
      for (int u = 0; u < R; ++u) {
        if (r0 + u * kBlock < r_hi) {
#pragma unroll
          for (int n = 0; n < NC; ++n) bs[n] = bs[n] + dv[u][n];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NC; ++n) sm[threadIdx.x * NC + n] = bs[n];
    __syncthreads();
    for (int w = kBlock / 2; w >= 1; w >>= 1) {
      if (static_cast<int>(threadIdx.x) < w) {
#pragma unroll
        for (int n = 0; n < NC; ++n) sm[threadIdx.x * NC + n] = sm[threadIdx.x * NC + n] + sm[(threadIdx.x + w) * NC + n];
      }
      __syncthreads();
    }
    if (threadIdx.x < NC) partial[(static_cast<int64_t>(blockIdx.x) * (K + 1) + K) * NC + threadIdx.x] = sm[threadIdx.x];
  }
}

// dW[e] (+)= sum over the chunks of partial[chunk][e]: one workgroup per element, lane t adds chunks t, t + 256, ..., then a
// fixed tree over the lanes (a serial loop over ~512 chunks per element was a 90 us chain of dependent loads)
__global__ void __launch_bounds__(kBlock)
wgrad_narrow_reduce_kernel(const float* __restrict__ partial, int chunks, int KN, int N, float* __restrict__ dW, int lddw,
                           int accumulate, float* __restrict__ dbias, int stride) {
  // (stride: floats per chunk record = KN, + N when the records carry the bias row; elements >= KN go to dbias)
  __shared__ float sm[kBlock];
  const int e = static_cast<int>(blockIdx.x);
  float t = 0.f;
  for (int c = threadIdx.x; c < chunks; c += kBlock) t = t + partial[static_cast<int64_t>(c) * stride + e];
  sm[threadIdx.x] = t;
  __syncthreads();
  for (int w = kBlock / 2; w >= 1; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) sm[threadIdx.x] = sm[threadIdx.x] + sm[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* p = e < KN ? dW + static_cast<int64_t>(e / N) * lddw + e % N : dbias + (e - KN);
    *p = accumulate ? *p + sm[0] : sm[0];
  }
}

// dq[b][j] = sum over the row tiles that hold rows of example b of its slot's partial (tile order: fixed)
__global__ void __launch_bounds__(kBlock)
din_dq_finish_kernel(const float* __restrict__ partial, int B, int L, int E, int slots, int64_t M, float* __restrict__ dq, int lddq) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * E) return;
  const int b = static_cast<int>(i / E), j = static_cast<int>(i % E);
  const int64_t r0 = static_cast<int64_t>(b) * L, r1 = r0 + L - 1;
  float s = 0.f;
  for (int64_t ty = r0 / BM; ty <= r1 / BM; ++ty) {
    const int slot = b - static_cast<int>((ty * BM) / L);
    s = s + partial[(ty * slots + slot) * E + j];
  }
  dq[static_cast<int64_t>(b) * lddq + j] = s;
}

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(kBlock)
gemm_f32_grouped_kernel(GroupedArgs ga) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kOpTile];
  const GroupedCoords c = grouped_coords(ga, blockIdx.x);
  if (c.split < 0) return;
  gemm_f32_block<A_KC, B_KC>(ga.p[c.p], c.tile, c.split, lds, c.plain);
}

// ... forward (NN) with bias + BatchNorm on the moving statistics + activation in the epilogue of the problems that ask for
// it (GemmArgs.fz_y): the experts of a multi-task model write z and y from one launch
__global__ void __launch_bounds__(kBlock)
gemm_f32_grouped_fz_kernel(GroupedArgs ga) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kOpTile];
  const GroupedCoords c = grouped_coords(ga, blockIdx.x);
  if (c.split < 0) return;
  gemm_f32_block<true, false, false, kEpiFrozenBn>(ga.p[c.p], c.tile, c.split, lds, c.plain);
}

// ... with the BatchNorm-backward column sums of each problem's producing layer in the epilogue (BnBwdEpi per problem)
template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(kBlock)
gemm_f32_grouped_bn_bwd_kernel(GroupedArgs ga) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kOpTile];
  const GroupedCoords c = grouped_coords(ga, blockIdx.x);
  if (c.split < 0) return;
  gemm_f32_block<A_KC, B_KC, true>(ga.p[c.p], c.tile, c.split, lds, c.plain);
}

// ... input gradients (NT) of which some also store the producing layer's dz (BnBwdEpi.dz_out: frozen statistics)
__global__ void __launch_bounds__(kBlock)
gemm_f32_grouped_bn_bwd_dz_kernel(GroupedArgs ga) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kOpTile];
  const GroupedCoords c = grouped_coords(ga, blockIdx.x);
  if (c.split < 0) return;
  gemm_f32_block<true, true, true, kEpiFrozenDz>(ga.p[c.p], c.tile, c.split, lds, c.plain);
}

// ------------------------------------------------------------------------------------------------
// bf16 inputs (rounded from fp32 while staging), fp32 accumulate.  LDS: As[m][k], Bs[n][k] in bf16, row
// stride 40 halves (80 B): the 16-byte fragment reads of a 16-lane group hit 16 distinct bank quads.
// Fragment of v_mfma_f32_32x32x16_bf16: lane l holds A[i = l & 31][k = 8 * (l >> 5) + 0..7].
// ------------------------------------------------------------------------------------------------
constexpr int kBf16SH = BK16 + 8;  // halves per LDS row
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_bf16_block(const GemmArgs& g, int bx, int bz, short* __restrict__ As, short* __restrict__ Bs,
                                                bool plain_tiles = false) {
  constexpr int SH = kBf16SH;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tx, ty;
  tile_coords(bx, static_cast<int>(ceil_div(g.N, BN)), static_cast<int>(ceil_div(g.M, BM)), tx, ty, plain_tiles);
  const int m0 = ty * BM, n0 = tx * BN;
  const int kbeg = bz * g.k_per_split;
  int kend = kbeg + g.k_per_split;
  if (kend > g.K) kend = g.K;
  const bool a_vec = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
  const bool b_vec = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
  f32x4v ra[2], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + i * kBlock;
      if (A_KC) ra[i] = load4<true>(g.A, g.lda, m0 + (u >> 3), k0 + (u & 7) * 4, g.M, kend, a_vec);
      else ra[i] = load4<false>(g.A, g.lda, m0 + (u & 15) * 4, k0 + (u >> 4), g.M, kend, a_vec);
      if (B_KC) rb[i] = load4<true>(g.B, g.ldb, n0 + (u >> 3), k0 + (u & 7) * 4, g.N, kend, b_vec);
      else rb[i] = load4<false>(g.B, g.ldb, n0 + (u & 15) * 4, k0 + (u >> 4), g.N, kend, b_vec);
    }
  };
  auto put = [&](short* S, const f32x4v& v, int u, bool kc) {
    if (kc) {  // 4 consecutive k of one row: one 8-byte store
      bf16x4 h = {f32_to_bf16_rne(v[0]), f32_to_bf16_rne(v[1]), f32_to_bf16_rne(v[2]), f32_to_bf16_rne(v[3])};
      *reinterpret_cast<bf16x4*>(&S[(u >> 3) * SH + (u & 7) * 4]) = h;
    } else {   // one k of 4 consecutive rows
      const int k = u >> 4, mn4 = (u & 15) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) S[(mn4 + j) * SH + k] = f32_to_bf16_rne(v[j]);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int a_row = wm * 32 + (lane & 31);
  const int b_row = wn * 32 + (lane & 31);
  const int kgrp = (lane >> 5) * 8;
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK16) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      put(As, ra[i], tid + i * kBlock, A_KC);
      put(Bs, rb[i], tid + i * kBlock, B_KC);
    }
    __syncthreads();
    if (k0 + BK16 < kend) fetch(k0 + BK16);
#pragma unroll
    for (int kk = 0; kk < BK16; kk += 16) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(&As[a_row * SH + kk + kgrp]);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bs[b_row * SH + kk + kgrp]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
  }
  const int col = n0 + wn * 32 + (lane & 31);
  const float bv = (g.bias && g.splits == 1 && col < g.N) ? g.bias[col] : 0.f;
  const int khalf = lane >> 5;
  if (g.col_stats)
    tile_col_stats(acc, bv, m0 + wm * 32, g.M, col, g.N, wm, wn, lane, reinterpret_cast<float*>(As),
                   g.col_stats + static_cast<int64_t>(ty) * g.N * 3);
  if (col >= g.N) return;
  float* Cz = g.C + (g.splits > 1 ? static_cast<int64_t>(bz) * g.M * g.N : 0);
  const int ldc = g.splits > 1 ? g.N : g.ldc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    if (row < g.M) {
      float* p = Cz + static_cast<int64_t>(row) * ldc + col;
      float v = acc[r] + bv;
      if (g.accumulate && g.splits == 1) v = *p + v;
      *p = v;
    }
  }
}

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(kBlock)
gemm_bf16_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) short As[BM * kBf16SH];
  __shared__ __attribute__((aligned(16))) short Bs[BN * kBf16SH];
  gemm_bf16_block<A_KC, B_KC>(g, blockIdx.x, blockIdx.z, As, Bs);
}

// several bf16 problems in ONE launch (the weight gradients of a bf16 step: er_gemm_grouped_bf16)
template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(kBlock)
gemm_bf16_grouped_kernel(GroupedArgs ga) {
  __shared__ __attribute__((aligned(16))) short As[BM * kBf16SH];
  __shared__ __attribute__((aligned(16))) short Bs[BN * kBf16SH];
  const GroupedCoords c = grouped_coords(ga, blockIdx.x);
  if (c.split < 0) return;
  gemm_bf16_block<A_KC, B_KC>(ga.p[c.p], c.tile, c.split, As, Bs, c.plain);
}


template <int VEC>
__global__ void __launch_bounds__(kBlock)
gemm_splitk_reduce_kernel(const float* __restrict__ ws, int64_t mn, int N, int splits, const float* __restrict__ bias,
                          float* __restrict__ C, int ldc, int accumulate) {
  splitk_reduce_elems<VEC>(ws, mn, N, splits, bias, C, ldc, accumulate,
                           (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * VEC);
}


template <int VEC>
__global__ void __launch_bounds__(kBlock)
gemm_splitk_reduce_grouped_kernel(GroupedReduceArgs ra) {
  splitk_reduce_grouped_block<VEC>(ra, blockIdx.x);
}

}  // namespace er

namespace {

float* g_gemm_ws = nullptr;
size_t g_gemm_ws_floats = 0;

int ensure_ws(size_t floats, float** out) {
  if (floats > g_gemm_ws_floats) {
    // growing is a hipMalloc: not capturable.  er_gemm_reserve() pre-sizes it before graph capture.
    if (g_gemm_ws) (void)hipFree(g_gemm_ws);
    g_gemm_ws = nullptr;
    g_gemm_ws_floats = 0;
    ER_CHECK_HIP(hipMalloc(&g_gemm_ws, floats * sizeof(float)));
    g_gemm_ws_floats = floats;
  }
  *out = g_gemm_ws;
  return 0;
}

template <bool BF16>
int launch_gemm(int layout, er::GemmArgs& a, hipStream_t s) {
  const int64_t n_tiles = er::ceil_div(a.N, er::BN) * er::ceil_div(a.M, er::BM);
  dim3 grid(static_cast<unsigned>(n_tiles), 1, static_cast<unsigned>(a.splits));  // (x fastest: split z's tiles are consecutive)
  dim3 block(er::kBlock);
#define ER_LAUNCH_GEMM(KERNEL)                                                   \
  switch (layout) {                                                              \
    case ER_GEMM_NN: hipLaunchKernelGGL((KERNEL<true, false>), grid, block, 0, s, a); break;  \
    case ER_GEMM_NT: hipLaunchKernelGGL((KERNEL<true, true>), grid, block, 0, s, a); break;   \
    case ER_GEMM_TN: hipLaunchKernelGGL((KERNEL<false, false>), grid, block, 0, s, a); break; \
    default: er::set_error("er_gemm: unknown layout %d", layout); return 2;     \
  }
  if (BF16) {
    ER_LAUNCH_GEMM(er::gemm_bf16_kernel)
  } else if (a.bn.partial) {
    ER_LAUNCH_GEMM(er::gemm_f32_bn_bwd_kernel)
  } else {
    ER_LAUNCH_GEMM(er::gemm_f32_kernel)
  }
#undef ER_LAUNCH_GEMM
  ER_LAUNCH_CHECK();
  return 0;
}

int choose_splits(int M, int N, int K, int ktile) {
  const int64_t tiles = er::ceil_div(M, er::BM) * er::ceil_div(N, er::BN);
  if (tiles >= 256 || K < 1024) return 1;  // split only the batch-long contractions (dW = x^T.dy)
  int64_t s = 512 / tiles;
  const int64_t max_by_k = K / (4 * ktile);  // at least 4 k-tiles per split
  if (s > max_by_k) s = max_by_k;
  if (s > 64) s = 64;
  return s < 1 ? 1 : static_cast<int>(s);
}

template <bool BF16>
int gemm_entry(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
               const float* bias, int accumulate, float* col_stats, er_stream_t stream, const char* who,
               const er::BnBwdEpi* bn = nullptr) {
  ER_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "%s: bad arguments", who);
  ER_REQUIRE(layout >= ER_GEMM_NN && layout <= ER_GEMM_TN, "%s: unknown layout %d", who, layout);
  const int min_lda = (layout == ER_GEMM_TN) ? M : K;
  const int min_ldb = (layout == ER_GEMM_NT) ? K : N;
  ER_REQUIRE(lda >= min_lda && ldb >= min_ldb && ldc >= N, "%s: leading dimension too small", who);
  hipStream_t s = er::as_stream(stream);
  constexpr int ktile = BF16 ? er::BK16 : er::BK32;
  er::GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.accumulate = accumulate;
  a.col_stats = col_stats;
  if (bn) {
    a.bn = *bn;
    if (a.bn.n_src == 0) { a.bn.col0 = 0; a.bn.n_src = N; }
  }
  a.splits = (col_stats || bn) ? 1 : choose_splits(M, N, K, ktile);
  ER_REQUIRE(!(col_stats && accumulate), "%s: column statistics need a plain (non-accumulating) output", who);
  a.k_per_split = static_cast<int>(er::ceil_div(er::ceil_div(K, a.splits), ktile)) * ktile;
  a.splits = static_cast<int>(er::ceil_div(K, a.k_per_split));
  if (a.splits > 1) {
    float* ws;
    if (int rc = ensure_ws(static_cast<size_t>(a.splits) * M * N, &ws)) return rc;
    a.C = ws;
    if (int rc = launch_gemm<BF16>(layout, a, s)) return rc;
    const int64_t mn = static_cast<int64_t>(M) * N;
    if (N % 4 == 0 && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
      hipLaunchKernelGGL(er::gemm_splitk_reduce_kernel<4>, dim3(static_cast<unsigned>(er::ceil_div(mn / 4, er::kBlock))),
                         dim3(er::kBlock), 0, s, ws, mn, N, a.splits, bias, C, ldc, accumulate);
    } else {
      hipLaunchKernelGGL(er::gemm_splitk_reduce_kernel<1>, dim3(static_cast<unsigned>(er::ceil_div(mn, er::kBlock))),
                         dim3(er::kBlock), 0, s, ws, mn, N, a.splits, bias, C, ldc, accumulate);
    }
    ER_LAUNCH_CHECK();
    return 0;
  }
  return launch_gemm<BF16>(layout, a, s);
}

}  // namespace

// The grouped launch's records for `n` (<= kMaxGroup) problems of one layout: k-splits, the XCD and legacy regions of the
// grid (GroupedArgs), the split-K workspace (grown here: not capturable - er_gemm_reserve) and the reduce launch's items.
int er::plan_grouped(int layout, const er_gemm_problem* pr, int n, bool bf16, er::GroupedPlan* plan, int64_t target_override) {
  er::GroupedArgs& ga = plan->ga;
  er::GroupedReduceArgs& ra = plan->ra;
  bool& any_bn = plan->any_bn;
  bool& any_fz = plan->any_fz;
  int64_t total_tiles = 0;
  for (int i = 0; i < n; ++i) {
    const er_gemm_problem& q = pr[i];
    ER_REQUIRE(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K > 0, "er_gemm_grouped_f32: problem %d: bad arguments", i);
    const int min_lda = (layout == ER_GEMM_TN) ? q.M : q.K;
    const int min_ldb = (layout == ER_GEMM_NT) ? q.K : q.N;
    ER_REQUIRE(q.lda >= min_lda && q.ldb >= min_ldb && q.ldc >= q.N,
               "er_gemm_grouped_f32: problem %d: leading dimension too small", i);
    ER_REQUIRE(!bf16 || !q.bn_partial, "er_gemm_grouped_bf16: problem %d: the BatchNorm-backward epilogue is fp32 only", i);
    total_tiles += er::ceil_div(q.M, er::BM) * er::ceil_div(q.N, er::BN);
  }
  // k-splits: enough workgroups for ~2 per CU over the whole group (A/B: 512 beat 1024 and 2048), >= 4 k-tiles per split
  static const int64_t target_blocks = [] {  // (A/B knob)
    const char* e = getenv("ER_WGRAD_TARGET_BLOCKS");
    const int64_t v = e ? atoll(e) : 0;
    return v >= 1 ? v : 512;
  }();
  const int64_t target = target_override > 0 ? target_override : target_blocks;
  int64_t want = total_tiles >= target ? 1 : er::ceil_div(target, total_tiles);
  // whole k-splits per XCD (GroupedArgs): the wanted split count moves to the nearest of 4, 8, 16, 24, ...
  static const bool by_xcd = [] {  // (A/B knob)
    const char* e = getenv("ER_WGRAD_XCD");
    return !(e && e[0] == '0');
  }();
  if (by_xcd && want >= 3) want = want <= 5 ? 4 : (want <= 11 ? 8 : ((want + 4) / 8) * 8);
  if (want > 64) want = 64;
  ga.n = 0;
  ga.start[0] = 0;
  ga.xstart[0] = 0;
  er::GemmArgs* slot[er::kMaxGroup];  // where problem i's arguments live (the workspace base is patched in below)
  ra.n = 0;
  ra.start[0] = 0;
  size_t ws_floats = 0;
  any_bn = false;
  any_fz = false;
  plan->any_dz = false;
  bool& any_dz = plan->any_dz;
  int n_splits[er::kMaxGroup], xcd_ok[er::kMaxGroup];
  for (int i = 0; i < n; ++i) {
    const er_gemm_problem& q = pr[i];
    er::GroupedArgs& grp = ga;
    er::GemmArgs& a = grp.p[grp.n];
    slot[i] = &a;
    a = er::GemmArgs();
    a.A = q.A; a.B = q.B; a.C = q.C; a.bias = q.bias;
    a.M = q.M; a.N = q.N; a.K = q.K; a.lda = q.lda; a.ldb = q.ldb; a.ldc = q.ldc;
    a.accumulate = q.accumulate;
    a.col_stats = q.col_stats;
    ER_REQUIRE(!(q.col_stats && q.accumulate), "er_gemm_grouped_f32: problem %d: column statistics need a plain output", i);
    if (q.bn_partial) {
      ER_REQUIRE(q.bn_z && q.bn_y && q.bn_ld >= q.N && !q.accumulate && (!q.bn_use_bn || (q.bn_mean && q.bn_invstd)),
                 "er_gemm_grouped_f32: problem %d: bad BatchNorm-backward epilogue arguments", i);
      a.bn.z = q.bn_z; a.bn.zbias = q.bn_zbias; a.bn.y = q.bn_y; a.bn.mean = q.bn_mean; a.bn.invstd = q.bn_invstd;
      a.bn.ld = q.bn_ld; a.bn.use_bn = q.bn_use_bn; a.bn.act = q.bn_act;
      a.bn.partial = q.bn_partial;
      a.bn.col0 = 0; a.bn.n_src = q.N;
      if (q.bn_dz_out) {
        ER_REQUIRE(q.bn_use_bn && q.bn_invstd, "er_gemm_grouped_f32: problem %d: bn_dz_out needs the layer's statistics", i);
        ER_REQUIRE(layout == ER_GEMM_NT, "er_gemm_grouped_f32: problem %d: bn_dz_out is an input-gradient (NT) epilogue", i);
        a.bn.gamma = q.bn_gamma;
        a.bn.dz_out = 1;
        any_dz = true;
      }
      any_bn = true;
    }
    if (q.fz_y) {
      ER_REQUIRE(layout == ER_GEMM_NN && !q.bn_partial && !q.col_stats && !q.accumulate && q.fz_mean && q.fz_var && q.fz_save &&
                     (q.fz_act == ER_ACT_NONE || q.fz_act == ER_ACT_RELU),
                 "er_gemm_grouped_f32: problem %d: bad frozen-BatchNorm epilogue arguments", i);
      a.bn.zbias = q.fz_bias; a.bn.z = q.fz_gamma; a.bn.y = q.fz_beta; a.bn.mean = q.fz_mean; a.bn.invstd = q.fz_var;
      a.bn.act = q.fz_act;
      a.fz_y = q.fz_y; a.fz_save = q.fz_save; a.fz_eps = q.fz_eps;
      any_fz = true;
    }
    int64_t sp = want;
    // a long contraction (DIN's attention MLP contracts over B x L = 204,800 rows into an 80-column output) gets
    // splits of at most 2048 rows whatever the group's tile count asks for: 13 splits of 15,753 rows took 0.52 ms
    static const int64_t split_rows = [] {  // (A/B knob for tools/gpu_round2_cc.sh)
      const char* e = getenv("ER_WGRAD_SPLIT_ROWS");
      const int64_t v = e ? atoll(e) : 0;
      return v >= 256 ? v : 2048;
    }();
    const int64_t by_len = er::ceil_div(q.K, split_rows);
    if (by_len > sp) sp = by_len;
    static const int64_t max_splits = [] {  // (A/B knob)
      const char* e = getenv("ER_WGRAD_MAX_SPLITS");
      const int64_t v = e ? atoll(e) : 0;
      return v >= 1 ? v : 128;
    }();
    if (sp > max_splits) sp = max_splits;
    const int64_t max_by_k = q.K / (4 * er::BK32);
    if (sp > max_by_k) sp = max_by_k;
    if (sp < 1 || q.col_stats || q.bn_partial || q.fz_y) sp = 1;
    a.k_per_split = static_cast<int>(er::ceil_div(er::ceil_div(q.K, sp), er::BK32)) * er::BK32;
    a.splits = static_cast<int>(er::ceil_div(q.K, a.k_per_split));
    const int64_t tiles = er::ceil_div(q.M, er::BM) * er::ceil_div(q.N, er::BN);
    grp.tiles[grp.n] = static_cast<int>(tiles);
    n_splits[grp.n] = a.splits;
    xcd_ok[grp.n] = (by_xcd && by_len <= want) ? 1 : 0;  // (a length-driven split count: legacy placement, GroupedArgs)
    ++grp.n;
    if (a.splits > 1) {
      const int64_t mn = static_cast<int64_t>(q.M) * q.N;
      er::ReduceItem& r = ra.r[ra.n];
      r.ws = reinterpret_cast<const float*>(ws_floats);  // offset for now: the base is known after ensure_ws
      r.mn = mn; r.N = q.N; r.splits = a.splits; r.bias = q.bias; r.C = q.C; r.ldc = q.ldc; r.accumulate = q.accumulate;
      r.vec = (q.N % 4 == 0 && q.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(q.C) & 15) == 0) ? 1 : 0;
      a.C = reinterpret_cast<float*>(ws_floats);
      ws_floats += static_cast<size_t>(a.splits) * mn;
      ws_floats = (ws_floats + 3) & ~static_cast<size_t>(3);
      ++ra.n;
    }
  }
  er::grouped_layout(ga.tiles, n_splits, ga.n, xcd_ok, ga.start, ga.xstart, ga.xsplits, ga.xper);
  if (ra.n > 0) {
    float* ws;
    if (int rc = ensure_ws(ws_floats, &ws)) return rc;
    int k = 0;
    for (int i = 0; i < n; ++i) {
      if (slot[i]->splits > 1) {
        const size_t off = reinterpret_cast<size_t>(slot[i]->C);
        slot[i]->C = ws + off;
        ra.r[k].ws = ws + off;
        ++k;
      }
    }
    for (int j = 0; j < ra.n; ++j) {
      const int64_t units = ra.r[j].vec ? ra.r[j].mn / 4 : ra.r[j].mn;
      ra.start[j + 1] = ra.start[j] + static_cast<int>(er::ceil_div(units, er::kBlock));
    }
  }
  return 0;
}

int er::launch_grouped_reduce(const er::GroupedReduceArgs& ra, hipStream_t s) {
  if (ra.n > 0) {
    dim3 rgrid(static_cast<unsigned>(ra.start[ra.n])), block(er::kBlock);
    hipLaunchKernelGGL(er::gemm_splitk_reduce_grouped_kernel<4>, rgrid, block, 0, s, ra);  // (16-byte lanes per problem: r.vec)
    ER_LAUNCH_CHECK();
  }
  return 0;
}

namespace {

int gemm_grouped_f32(int layout, const er_gemm_problem* pr, int n, er_stream_t stream, bool bf16 = false) {
  hipStream_t s = er::as_stream(stream);
  er::GroupedPlan plan;
  if (int rc = er::plan_grouped(layout, pr, n, bf16, &plan)) return rc;
  const er::GroupedArgs& ga = plan.ga;
  const bool any_bn = plan.any_bn;
  dim3 grid(static_cast<unsigned>(er::grouped_grid(ga))), block(er::kBlock);
  ER_REQUIRE(!plan.any_fz || (!bf16 && !any_bn && layout == ER_GEMM_NN), "er_gemm_grouped: the frozen-BatchNorm epilogue is fp32 NN only");
  ER_REQUIRE(!plan.any_dz || !bf16, "er_gemm_grouped: bn_dz_out is fp32 only");
  if (plan.any_fz) {
    hipLaunchKernelGGL(er::gemm_f32_grouped_fz_kernel, grid, block, 0, s, ga);
  } else if (bf16) {
    switch (layout) {
      case ER_GEMM_NN: hipLaunchKernelGGL((er::gemm_bf16_grouped_kernel<true, false>), grid, block, 0, s, ga); break;
      case ER_GEMM_NT: hipLaunchKernelGGL((er::gemm_bf16_grouped_kernel<true, true>), grid, block, 0, s, ga); break;
      case ER_GEMM_TN: hipLaunchKernelGGL((er::gemm_bf16_grouped_kernel<false, false>), grid, block, 0, s, ga); break;
      default: er::set_error("er_gemm_grouped_bf16: unknown layout %d", layout); return 2;
    }
  } else if (plan.any_dz) {
    ER_REQUIRE(!bf16 && layout == ER_GEMM_NT, "er_gemm_grouped: bn_dz_out is fp32 NT only");
    hipLaunchKernelGGL(er::gemm_f32_grouped_bn_bwd_dz_kernel, grid, block, 0, s, ga);
  } else if (any_bn) {
    switch (layout) {
      case ER_GEMM_NN: hipLaunchKernelGGL((er::gemm_f32_grouped_bn_bwd_kernel<true, false>), grid, block, 0, s, ga); break;
      case ER_GEMM_NT: hipLaunchKernelGGL((er::gemm_f32_grouped_bn_bwd_kernel<true, true>), grid, block, 0, s, ga); break;
      case ER_GEMM_TN: hipLaunchKernelGGL((er::gemm_f32_grouped_bn_bwd_kernel<false, false>), grid, block, 0, s, ga); break;
      default: er::set_error("er_gemm_grouped_f32: unknown layout %d", layout); return 2;
    }
  } else {
    switch (layout) {
      case ER_GEMM_NN: hipLaunchKernelGGL((er::gemm_f32_grouped_kernel<true, false>), grid, block, 0, s, ga); break;
      case ER_GEMM_NT: hipLaunchKernelGGL((er::gemm_f32_grouped_kernel<true, true>), grid, block, 0, s, ga); break;
      case ER_GEMM_TN: hipLaunchKernelGGL((er::gemm_f32_grouped_kernel<false, false>), grid, block, 0, s, ga); break;
      default: er::set_error("er_gemm_grouped_f32: unknown layout %d", layout); return 2;
    }
  }
  ER_LAUNCH_CHECK();
  return er::launch_grouped_reduce(plan.ra, s);
}

}  // namespace

extern "C" {

int er_gemm_reserve(int64_t floats) {
  ER_REQUIRE(floats >= 0, "er_gemm_reserve: negative size");
  float* p;
  return ensure_ws(static_cast<size_t>(floats), &p);
}

int er_gemm_f32(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                float* C, int32_t ldc, const float* bias, int accumulate, float* col_stats, er_stream_t stream) {
  return gemm_entry<false>(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, col_stats, stream, "er_gemm_f32");
}

int er_gemm_bf16(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                 float* C, int32_t ldc, const float* bias, int accumulate, float* col_stats, er_stream_t stream) {
  return gemm_entry<true>(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, col_stats, stream, "er_gemm_bf16");
}

int er_gemm_f32_bn_bwd(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                       int32_t ldb, float* C, int32_t ldc, const float* z, const float* z_bias, const float* y,
                       const float* save_mean, const float* save_invstd, int32_t ld_zy, int use_bn, int act,
                       float* partial, er_stream_t stream) {
  ER_REQUIRE(z && y && partial && ld_zy >= N, "er_gemm_f32_bn_bwd: bad epilogue arguments");
  ER_REQUIRE(!use_bn || (save_mean && save_invstd), "er_gemm_f32_bn_bwd: BatchNorm statistics missing");
  er::BnBwdEpi e;
  e.z = z; e.zbias = z_bias; e.y = y; e.mean = save_mean; e.invstd = save_invstd;
  e.ld = ld_zy; e.use_bn = use_bn; e.act = act; e.partial = partial;
  return gemm_entry<false>(layout, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, nullptr, stream, "er_gemm_f32_bn_bwd", &e);
}

int er_gemm_f32_bn_bwd_cols(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                            int32_t ldb, float* C, int32_t ldc, const float* z, const float* z_bias, const float* y,
                            const float* save_mean, const float* save_invstd, int32_t ld_zy, int use_bn, int act,
                            float* partial, int32_t col0, int32_t n_src, er_stream_t stream) {
  ER_REQUIRE(z && y && partial && col0 >= 0 && n_src >= 1 && col0 + n_src <= N && ld_zy >= n_src,
             "er_gemm_f32_bn_bwd_cols: bad epilogue arguments (source columns [%d, %d) of %d outputs)", col0, col0 + n_src, N);
  ER_REQUIRE(!use_bn || (save_mean && save_invstd), "er_gemm_f32_bn_bwd_cols: BatchNorm statistics missing");
  er::BnBwdEpi e;
  e.z = z; e.zbias = z_bias; e.y = y; e.mean = save_mean; e.invstd = save_invstd;
  e.ld = ld_zy; e.use_bn = use_bn; e.act = act; e.partial = partial;
  e.col0 = col0; e.n_src = n_src;
  return gemm_entry<false>(layout, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, nullptr, stream, "er_gemm_f32_bn_bwd_cols", &e);
}

int er_gemm_f32_cross(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                      float* C, int32_t ldc, const float* bias, int accumulate, const er_gemm_epilogue* epi,
                      er_stream_t stream) {
  ER_REQUIRE(A && B && C && epi && M > 0 && N > 0 && K > 0, "er_gemm_f32_cross: bad arguments");
  ER_REQUIRE(layout == ER_GEMM_NN || layout == ER_GEMM_NT, "er_gemm_f32_cross: layout %d (NN forward, NT backward)", layout);
  ER_REQUIRE(lda >= K && ldb >= (layout == ER_GEMM_NT ? K : N) && ldc >= N, "er_gemm_f32_cross: leading dimension too small");
  er_gemm_epilogue e = *epi;
  if (e.kind == ER_EPI_CROSS_FWD) {
    ER_REQUIRE(e.x0 && e.xl && e.ld_x0 >= N && e.ld_xl >= N && (!e.u || e.ld_u >= N) && !accumulate,
               "er_gemm_f32_cross: bad forward epilogue arguments");
  } else if (e.kind == ER_EPI_CROSS_BWD) {
    ER_REQUIRE(e.dout && e.ld_dout >= N && (e.diag == 0.f || (e.du_in && e.ld_du_in >= N)),
               "er_gemm_f32_cross: bad backward epilogue arguments (dout / du_in)");
    ER_REQUIRE(!e.prev_u || (e.x0 && e.dx0 && e.ld_x0 >= N && e.ld_prev_u >= N && e.ld_dx0 >= N &&
                             (!e.du_out || e.ld_du_out >= N) && (!e.du_out_bf16 || e.ld_du_out_bf16 >= N) &&
                             (e.diag == 0.f || (e.xl && e.ld_xl >= N))),
               "er_gemm_f32_cross: bad backward epilogue arguments (the lower layer's part)");
  } else {
    ER_REQUIRE(false, "er_gemm_f32_cross: epilogue kind %d", e.kind);
  }
  er::GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.accumulate = accumulate;
  a.col_stats = nullptr;
  a.splits = 1;
  a.k_per_split = static_cast<int>(er::ceil_div(K, er::BK32)) * er::BK32;
  const int64_t n_tiles = er::ceil_div(N, er::BN) * er::ceil_div(M, er::BM);
  dim3 grid(static_cast<unsigned>(n_tiles)), block(er::kBlock);
  hipStream_t s = er::as_stream(stream);
  if (e.kind == ER_EPI_CROSS_FWD) {
    if (layout == ER_GEMM_NN) hipLaunchKernelGGL((er::gemm_f32_cross_kernel<true, false, ER_EPI_CROSS_FWD>), grid, block, 0, s, a, e);
    else hipLaunchKernelGGL((er::gemm_f32_cross_kernel<true, true, ER_EPI_CROSS_FWD>), grid, block, 0, s, a, e);
  } else {
    if (layout == ER_GEMM_NN) hipLaunchKernelGGL((er::gemm_f32_cross_kernel<true, false, ER_EPI_CROSS_BWD>), grid, block, 0, s, a, e);
    else hipLaunchKernelGGL((er::gemm_f32_cross_kernel<true, true, ER_EPI_CROSS_BWD>), grid, block, 0, s, a, e);
  }
  ER_LAUNCH_CHECK();
  return 0;
}

int er_gemm_f32_bn_a(int32_t M, int32_t N, int32_t K, const float* z, int32_t ldz, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int act, float* y, int32_t ldy, const float* W, int32_t ldw, float* C,
                     int32_t ldc, const float* bias, float* col_stats, er_stream_t stream) {
  ER_REQUIRE(z && mean && invstd && W && C && M > 0 && N > 0 && K > 0, "er_gemm_f32_bn_a: bad arguments");
  ER_REQUIRE(K % 4 == 0 && K <= er::kBnAMaxK, "er_gemm_f32_bn_a: K = %d (a multiple of 4, at most %d)", K, er::kBnAMaxK);
  ER_REQUIRE(ldz >= K && ldw >= N && ldc >= N && (!y || ldy >= K), "er_gemm_f32_bn_a: leading dimension too small");
  ER_REQUIRE(ldz % 4 == 0 && (!y || ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
             "er_gemm_f32_bn_a: z / y must be 16-byte aligned with leading dimensions that are multiples of 4");
  ER_REQUIRE(act == ER_ACT_NONE || act == ER_ACT_RELU, "er_gemm_f32_bn_a: activation %d", act);
  er::GemmArgs a;
  a.A = z; a.B = W; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = ldz; a.ldb = ldw; a.ldc = ldc;
  a.accumulate = 0;
  a.col_stats = col_stats;
  a.splits = 1;
  a.k_per_split = static_cast<int>(er::ceil_div(K, er::BK32)) * er::BK32;
  er::BnA b{mean, invstd, gamma, beta, act, y, ldy};
  const int64_t n_tiles = er::ceil_div(N, er::BN) * er::ceil_div(M, er::BM);
  hipLaunchKernelGGL(er::gemm_f32_bna_kernel, dim3(static_cast<unsigned>(n_tiles)), dim3(er::kBlock), 0, er::as_stream(stream), a, b);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_gemv_f32_bn_a(int32_t M, int32_t N, int32_t K, const float* x, int32_t ldx, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int act, float* y, int32_t ldy, const float* W, int32_t ldw, float* C,
                     int32_t ldc, const float* bias, er_stream_t stream) {
  ER_REQUIRE(x && W && C && M > 0 && N >= 1 && N <= 4 && K > 0, "er_gemv_f32_bn_a: bad arguments (1 <= N <= 4)");
  ER_REQUIRE(K % 4 == 0 && ldx >= K && ldx % 4 == 0 && ldw >= N && ldc >= N && (reinterpret_cast<uintptr_t>(x) & 15) == 0,
             "er_gemv_f32_bn_a: K and ldx multiples of 4, x 16-byte aligned");
  ER_REQUIRE(!mean || (invstd && (!y || (ldy >= K && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0))),
             "er_gemv_f32_bn_a: bad BatchNorm arguments");
  ER_REQUIRE(act == ER_ACT_NONE || act == ER_ACT_RELU, "er_gemv_f32_bn_a: activation %d", act);
  int lpr = 1;
  while (lpr * 2 <= 8 && lpr * 2 <= K / 4) lpr *= 2;
  er::BnA b{mean, invstd, gamma, beta, act, y, ldy};
  const int rows_per_block = (er::kBlock / lpr) * 4;
  dim3 grid(static_cast<unsigned>(er::ceil_div(M, rows_per_block))), block(er::kBlock);
  hipStream_t s = er::as_stream(stream);
  switch (N) {
    case 1: hipLaunchKernelGGL(er::gemv_bna_kernel<1>, grid, block, 0, s, x, ldx, M, K, b, W, ldw, bias, C, ldc, lpr); break;
    case 2: hipLaunchKernelGGL(er::gemv_bna_kernel<2>, grid, block, 0, s, x, ldx, M, K, b, W, ldw, bias, C, ldc, lpr); break;
    case 3: hipLaunchKernelGGL(er::gemv_bna_kernel<3>, grid, block, 0, s, x, ldx, M, K, b, W, ldw, bias, C, ldc, lpr); break;
    default: hipLaunchKernelGGL(er::gemv_bna_kernel<4>, grid, block, 0, s, x, ldx, M, K, b, W, ldw, bias, C, ldc, lpr); break;
  }
  ER_LAUNCH_CHECK();
  return 0;
}

int er_wgrad_tall_narrow(int32_t rows, int32_t K, int32_t N, const float* x, int32_t ldx, const float* dz, int32_t lddz,
                         float* dW, int32_t lddw, float* dbias, int accumulate, float* scratch, int64_t scratch_floats,
                         er_stream_t stream) {
  ER_REQUIRE(x && dz && dW && scratch && rows > 0 && K > 0 && N >= 1 && N <= 4, "er_wgrad_tall_narrow: bad arguments (1 <= N <= 4)");
  ER_REQUIRE(K % 4 == 0 && ldx >= K && ldx % 4 == 0 && lddz >= N && lddw >= N && (reinterpret_cast<uintptr_t>(x) & 15) == 0,
             "er_wgrad_tall_narrow: K and ldx multiples of 4, x 16-byte aligned");
  int lpr = 1;
  while (lpr * 2 <= 8 && lpr * 2 <= K / 4) lpr *= 2;
  // ~2 workgroups per compute unit, whole trips of (row lanes x 4 rows) per chunk
  const int trip = (er::kBlock / lpr) * 4;
  int64_t rpc = er::ceil_div(rows, 512);
  rpc = er::ceil_div(rpc, trip) * trip;
  const int chunks = static_cast<int>(er::ceil_div(rows, rpc));
  const int wb = dbias ? 1 : 0;
  ER_REQUIRE(static_cast<int64_t>(chunks) * (K + wb) * N <= scratch_floats, "er_wgrad_tall_narrow: scratch of %lld floats, need %lld",
             (long long)scratch_floats, (long long)chunks * (K + wb) * N);
  hipStream_t s = er::as_stream(stream);
  dim3 grid(static_cast<unsigned>(chunks)), block(er::kBlock);
  const int irpc = static_cast<int>(rpc);
  switch (N) {
    case 1: hipLaunchKernelGGL(er::wgrad_narrow_partial_kernel<1>, grid, block, 0, s, x, ldx, dz, lddz, rows, K, irpc, lpr, scratch, wb); break;
    case 2: hipLaunchKernelGGL(er::wgrad_narrow_partial_kernel<2>, grid, block, 0, s, x, ldx, dz, lddz, rows, K, irpc, lpr, scratch, wb); break;
    case 3: hipLaunchKernelGGL(er::wgrad_narrow_partial_kernel<3>, grid, block, 0, s, x, ldx, dz, lddz, rows, K, irpc, lpr, scratch, wb); break;
    default: hipLaunchKernelGGL(er::wgrad_narrow_partial_kernel<4>, grid, block, 0, s, x, ldx, dz, lddz, rows, K, irpc, lpr, scratch, wb); break;
  }
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::wgrad_narrow_reduce_kernel, dim3(static_cast<unsigned>((K + wb) * N)), block, 0, s, scratch, chunks, K * N, N,
                     dW, lddw, accumulate, dbias, (K + wb) * N);
  ER_LAUNCH_CHECK();
  return 0;
}

namespace {
int din_gen(const char* who, const float* q, int32_t ldq, const float* h, int32_t ldh, int32_t B, int32_t L, int32_t E,
            er::DinGen* d) {
  ER_REQUIRE(q && h && B > 0 && L > 0 && E > 0 && E % 4 == 0 && ldq >= E && ldh >= E && ldq % 4 == 0 && ldh % 4 == 0 &&
                 ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(h)) & 15) == 0,
             "%s: q [B][ldq] / h [B * L][ldh] must be 16-byte aligned rows, E a multiple of 4", who);
  ER_REQUIRE(static_cast<int64_t>(B) * L < (1LL << 31) / (L > 0 ? L : 1) * L && static_cast<int64_t>(B) * L < (1LL << 32) / L,
             "%s: B * L too large for the row / L shortcut", who);
  *d = er::DinGen();
  d->q = q; d->h = h; d->ldq = ldq; d->ldh = ldh; d->L = L; d->E = E;
  d->inv_L = static_cast<uint32_t>(((1ULL << 32) + L - 1) / L);
  d->inv_E = static_cast<uint32_t>(((1ULL << 32) + E - 1) / E);
  if (L == 1) d->inv_L = 0xFFFFFFFFu;  // (2^32 does not fit: row * (2^32 - 1) >> 32 == row - 1 for row >= 1 - handled below)
  return 0;
}
}  // namespace

int er_din_gemm_fwd(const float* q, int32_t ldq, const float* h, int32_t ldh, int32_t B, int32_t L, int32_t E, const float* W,
                    int32_t ldw, int32_t N, const float* bias, float* z, int32_t ldz, float* col_stats, er_stream_t stream) {
  er::DinGen d;
  if (int rc = din_gen("er_din_gemm_fwd", q, ldq, h, ldh, B, L, E, &d)) return rc;
  ER_REQUIRE(L >= 2, "er_din_gemm_fwd: L >= 2");
  ER_REQUIRE(W && z && N > 0 && ldw >= N && ldz >= N && ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0,
             "er_din_gemm_fwd: W [4E][ldw] must have 16-byte aligned rows");
  er::GemmArgs a;
  a.A = nullptr; a.B = W; a.C = z; a.bias = bias;
  a.M = B * L; a.N = N; a.K = 4 * E; a.lda = 4 * E; a.ldb = ldw; a.ldc = ldz;
  a.accumulate = 0; a.col_stats = col_stats; a.splits = 1;
  a.k_per_split = static_cast<int>(er::ceil_div(a.K, er::BK32)) * er::BK32;
  const int64_t n_tiles = er::ceil_div(a.N, er::BN) * er::ceil_div(a.M, er::BM);
  hipLaunchKernelGGL((er::gemm_f32_din_kernel<true, false, 1>), dim3(static_cast<unsigned>(n_tiles)), dim3(er::kBlock), 0,
                     er::as_stream(stream), a, d);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_din_gemm_wgrad(const float* q, int32_t ldq, const float* h, int32_t ldh, int32_t B, int32_t L, int32_t E, const float* dz,
                      int32_t lddz, int32_t N, float* dW, int32_t lddw, int accumulate, er_stream_t stream) {
  er::DinGen d;
  if (int rc = din_gen("er_din_gemm_wgrad", q, ldq, h, ldh, B, L, E, &d)) return rc;
  ER_REQUIRE(L >= 2, "er_din_gemm_wgrad: L >= 2");
  ER_REQUIRE(dz && dW && N > 0 && lddz >= N && lddw >= N && lddz % 4 == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0,
             "er_din_gemm_wgrad: dz [B * L][lddz] must have 16-byte aligned rows");
  hipStream_t s = er::as_stream(stream);
  er::GemmArgs a;
  a.A = nullptr; a.B = dz; a.bias = nullptr;
  a.M = 4 * E; a.N = N; a.K = B * L; a.lda = 4 * E; a.ldb = lddz; a.ldc = lddw;
  a.accumulate = accumulate; a.col_stats = nullptr;
  // a batch-long contraction into a small output, alone in its launch: enough k-splits for ~4 workgroups per CU (the
  // launch's 4 - 8 output tiles x 100 splits of 2048 rows - er_gemm_grouped_f32's rule - left 112 of 256 CUs with one
  // workgroup and 144 with two: 92.7 us against ~55 us for the same problem inside a grouped launch)
  static const int64_t din_wgrad_blocks = [] {  // (A/B knob)
    const char* e = getenv("ER_DIN_WGRAD_BLOCKS");
    const int64_t v = e ? atoll(e) : 0;
    return v >= 1 ? v : 1024;
  }();
  const int64_t out_tiles = er::ceil_div(a.N, er::BN) * er::ceil_div(a.M, er::BM);
  int64_t splits = er::ceil_div(din_wgrad_blocks, out_tiles);
  const int64_t max_by_k = a.K / (8 * er::BK32);
  if (splits > 256) splits = 256;
  if (splits > max_by_k) splits = max_by_k;
  if (splits < 1) splits = 1;
  a.k_per_split = static_cast<int>(er::ceil_div(er::ceil_div(a.K, splits), er::BK32)) * er::BK32;
  a.splits = static_cast<int>(er::ceil_div(a.K, a.k_per_split));
  const int64_t n_tiles = er::ceil_div(a.N, er::BN) * er::ceil_div(a.M, er::BM);
  dim3 grid(static_cast<unsigned>(n_tiles), 1, static_cast<unsigned>(a.splits));
  if (a.splits > 1) {
    float* ws;
    if (int rc = ensure_ws(static_cast<size_t>(a.splits) * a.M * a.N, &ws)) return rc;
    a.C = ws;
    hipLaunchKernelGGL((er::gemm_f32_din_kernel<false, false, 1>), grid, dim3(er::kBlock), 0, s, a, d);
    ER_LAUNCH_CHECK();
    const int64_t mn = static_cast<int64_t>(a.M) * a.N;
    if (N % 4 == 0 && lddw % 4 == 0 && (reinterpret_cast<uintptr_t>(dW) & 15) == 0) {
      hipLaunchKernelGGL(er::gemm_splitk_reduce_kernel<4>, dim3(static_cast<unsigned>(er::ceil_div(mn / 4, er::kBlock))),
                         dim3(er::kBlock), 0, s, ws, mn, N, a.splits, static_cast<const float*>(nullptr), dW, lddw, accumulate);
    } else {
      hipLaunchKernelGGL(er::gemm_splitk_reduce_kernel<1>, dim3(static_cast<unsigned>(er::ceil_div(mn, er::kBlock))),
                         dim3(er::kBlock), 0, s, ws, mn, N, a.splits, static_cast<const float*>(nullptr), dW, lddw, accumulate);
    }
  } else {
    a.C = dW;
    hipLaunchKernelGGL((er::gemm_f32_din_kernel<false, false, 1>), grid, dim3(er::kBlock), 0, s, a, d);
  }
  ER_LAUNCH_CHECK();
  return 0;
}

int64_t er_din_dq_partial_floats(int32_t B, int32_t L, int32_t E) {
  if (B <= 0 || L <= 0 || E <= 0) return -1;
  return er::ceil_div(static_cast<int64_t>(B) * L, er::BM) * ((er::BM - 1) / L + 2) * E;
}

int er_din_gemm_dgrad(const float* dz, int32_t lddz, int32_t N, const float* W, int32_t ldw, const float* q, int32_t ldq,
                      const float* h, int32_t ldh, int32_t B, int32_t L, int32_t E, float* dq, int32_t lddq, float* dh,
                      int32_t lddh, int accumulate_dh, float* dq_partial, er_stream_t stream) {
  er::DinGen d;
  if (int rc = din_gen("er_din_gemm_dgrad", q, ldq, h, ldh, B, L, E, &d)) return rc;
  ER_REQUIRE(L >= 2 && E % 16 == 0, "er_din_gemm_dgrad: E must be a multiple of 16 (one 64-column tile holds the four segments of 16 positions), L >= 2");
  ER_REQUIRE(dz && W && dq && dh && dq_partial && N > 0 && lddz >= N && ldw >= N && lddq >= E && lddh >= E && lddz % 4 == 0 &&
                 ldw % 4 == 0 && ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(W)) & 15) == 0,
             "er_din_gemm_dgrad: dz [B * L][lddz] / W [4E][ldw] must have 16-byte aligned rows");
  d.dh = dh; d.lddh = lddh; d.accumulate_dh = accumulate_dh;
  d.dq_partial = dq_partial;
  d.slots = (er::BM - 1) / L + 2;
  hipStream_t s = er::as_stream(stream);
  er::GemmArgs a;
  a.A = dz; a.B = W; a.C = nullptr; a.bias = nullptr;
  a.M = B * L; a.N = 4 * E; a.K = N; a.lda = lddz; a.ldb = ldw; a.ldc = 4 * E;
  a.accumulate = 0; a.col_stats = nullptr; a.splits = 1;
  a.k_per_split = static_cast<int>(er::ceil_div(a.K, er::BK32)) * er::BK32;
  const int64_t n_tiles = er::ceil_div(a.N, er::BN) * er::ceil_div(a.M, er::BM);
  hipLaunchKernelGGL((er::gemm_f32_din_kernel<true, true, 2>), dim3(static_cast<unsigned>(n_tiles)), dim3(er::kBlock), 0, s, a, d);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::din_dq_finish_kernel, dim3(static_cast<unsigned>(er::ceil_div(static_cast<int64_t>(B) * E, er::kBlock))),
                     dim3(er::kBlock), 0, s, dq_partial, B, L, E, d.slots, static_cast<int64_t>(B) * L, dq, lddq);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_gemm_grouped_f32(int layout, const er_gemm_problem* problems, int n, er_stream_t stream) {
  ER_REQUIRE(problems && n > 0, "er_gemm_grouped_f32: bad arguments");
  ER_REQUIRE(layout >= ER_GEMM_NN && layout <= ER_GEMM_TN, "er_gemm_grouped_f32: unknown layout %d", layout);
  for (int i = 0; i < n; i += er::kMaxGroup) {
    const int m = n - i < er::kMaxGroup ? n - i : er::kMaxGroup;
    if (int rc = gemm_grouped_f32(layout, problems + i, m, stream)) return rc;
  }
  return 0;
}

int er_gemm_grouped_bf16(int layout, const er_gemm_problem* problems, int n, er_stream_t stream) {
  ER_REQUIRE(problems && n > 0, "er_gemm_grouped_bf16: bad arguments");
  ER_REQUIRE(layout >= ER_GEMM_NN && layout <= ER_GEMM_TN, "er_gemm_grouped_bf16: unknown layout %d", layout);
  for (int i = 0; i < n; i += er::kMaxGroup) {
    const int m = n - i < er::kMaxGroup ? n - i : er::kMaxGroup;
    if (int rc = gemm_grouped_f32(layout, problems + i, m, stream, true)) return rc;
  }
  return 0;
}

int er_gemm_grouped_layout(const int32_t* tiles, const int32_t* splits, int n, const int32_t* by_xcd, int32_t* start,
                           int32_t* xstart, int32_t* xsplits, int32_t* xper) {
  if (!(tiles && splits && start && xstart && xsplits && xper && n >= 1 && n <= er::kMaxGroup)) return -1;
  er::grouped_layout(tiles, splits, n, by_xcd, start, xstart, xsplits, xper);
  return er::grouped_grid(start, xstart, n);
}

int er_gemm_grouped_coords(const int32_t* tiles, const int32_t* start, const int32_t* xstart, const int32_t* xsplits,
                           const int32_t* xper, int n, int32_t block, int32_t* problem, int32_t* tile, int32_t* split,
                           int32_t* plain) {
  ER_REQUIRE(tiles && start && xstart && xsplits && xper && problem && tile && split && plain && n >= 1 && n <= er::kMaxGroup,
             "er_gemm_grouped_coords: bad arguments");
  ER_REQUIRE(block >= 0 && block < er::grouped_grid(start, xstart, n), "er_gemm_grouped_coords: block %d outside the grid", block);
  const er::GroupedCoords c = er::grouped_coords(start, tiles, xstart, xsplits, xper, n, block);
  *problem = c.p; *tile = c.tile; *split = c.split; *plain = c.plain ? 1 : 0;
  return 0;
}

int er_gemm_row_tiles(int32_t M) { return static_cast<int>(er::ceil_div(M, er::BM)); }

}  // extern "C"
