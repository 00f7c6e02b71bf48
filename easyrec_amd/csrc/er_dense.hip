// K9/K10 + dense optimizer: the elementwise / column-reduction work around the MLP GEMMs
// (bias + BatchNorm(train) + ReLU, Dice), sigmoid cross-entropy, L2 terms, ApplyAdam over the
// flat dense-parameter buffer.  HBM/L2-bound; fuses per layer what the reference issues as
// BiasAdd + FusedBatchNorm/moments + Relu (layers/dnn.py:57-79).
//
// Reference call sites: layers/dnn.py:50-87, layers/keras/blocks.py:84-110,
// layers/keras/activation.py:47-70, builders/loss_builder.py:35-39, compat/regularizers.py:76-108,
// compat/optimizers.py:412-416 (apply_gradients -> tf.train.AdamOptimizer._apply_dense).
#include <mutex>

#include "er_common.h"
#include "er_dense_tail.h"
#include "er_decay.h"
#include "er_farmhash.h"
#include "er_fm_bodies.h"

namespace er {

// ------------------------------------------------------------------------------------------------
// column statistics in row chunks.  block = 64 columns x 4 row lanes; grid = (col_blocks, chunks)
// ------------------------------------------------------------------------------------------------
constexpr int kColsPerBlock = 64;
constexpr int kRowLanes = kBlock / kColsPerBlock;  // 4
constexpr int kMaxChunks = 1024;  // (B = 4096 layers stay at <= 128: rows / 32; tall activations get more workgroups)

inline int choose_chunks(int B, int N) {
  const int col_blocks = static_cast<int>(ceil_div(N, kColsPerBlock));
  int chunks = 1024 / col_blocks;
  const int max_by_rows = static_cast<int>(ceil_div(B, 32));
  if (chunks > max_by_rows) chunks = max_by_rows;
  if (chunks > kMaxChunks) chunks = kMaxChunks;
  if (chunks < 1) chunks = 1;
  return chunks;
}

struct Welford {
  float n, mean, m2;
};
__device__ __forceinline__ Welford wf_merge(const Welford& a, const Welford& b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  const float n = a.n + b.n;
  const float delta = b.mean - a.mean;
  Welford r;
  r.n = n;
  r.mean = a.mean + delta * (b.n / n);
  r.m2 = a.m2 + b.m2 + delta * delta * (a.n * b.n / n);
  return r;
}

// partial[(chunk*N + c)*3 + {0,1,2}] = (count, mean, M2) of z = x + bias over the chunk's rows
__global__ void __launch_bounds__(kBlock)
bn_stats_partial_kernel(const float* __restrict__ x, const float* __restrict__ bias, int B, int N, int chunks,
                        float* __restrict__ partial) {
  __shared__ Welford sm[kRowLanes][kColsPerBlock];
  const int cl = threadIdx.x % kColsPerBlock;
  const int rl = threadIdx.x / kColsPerBlock;
  const int c = blockIdx.x * kColsPerBlock + cl;
  const int chunk = blockIdx.y;
  const int rows_per_chunk = static_cast<int>(ceil_div(B, chunks));
  const int r0 = chunk * rows_per_chunk;
  const int r1 = (r0 + rows_per_chunk < B) ? r0 + rows_per_chunk : B;
  Welford w{0.f, 0.f, 0.f};
  if (c < N) {
    const float bv = bias ? bias[c] : 0.f;
    for (int r = r0 + rl; r < r1; r += kRowLanes) {
      const float v = x[static_cast<int64_t>(r) * N + c] + bv;
      w.n += 1.f;
      const float d = v - w.mean;
      w.mean += d / w.n;
      w.m2 += d * (v - w.mean);
    }
  }
  sm[rl][cl] = w;
  __syncthreads();
  if (rl == 0 && c < N) {
    Welford t = wf_merge(wf_merge(sm[0][cl], sm[1][cl]), wf_merge(sm[2][cl], sm[3][cl]));
    float* p = partial + (static_cast<int64_t>(chunk) * N + c) * 3;
    p[0] = t.n; p[1] = t.mean; p[2] = t.m2;
  }
}

// one wave per column: lanes take chunks k, k+64, ..; then a fixed butterfly of Welford merges
__global__ void __launch_bounds__(kBlock)
bn_finalize_kernel(const float* __restrict__ partial, int B, int N, int chunks, float eps, float momentum,
                   float* __restrict__ moving_mean, float* __restrict__ moving_var, float* __restrict__ save_mean,
                   float* __restrict__ save_invstd) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (c >= N) return;
  Welford t{0.f, 0.f, 0.f};
  for (int k = lane; k < chunks; k += 64) {
    const float* p = partial + (static_cast<int64_t>(k) * N + c) * 3;
    t = wf_merge(t, Welford{p[0], p[1], p[2]});
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    Welford o{__shfl_xor(t.n, off, 64), __shfl_xor(t.mean, off, 64), __shfl_xor(t.m2, off, 64)};
    // merge in a lane-independent order (lower lane first) so that every lane holds the same bits
    t = (lane & off) ? wf_merge(o, t) : wf_merge(t, o);
  }
  if (lane != 0) return;
  const float mean = t.mean;
  const float var = t.m2 / static_cast<float>(B);  // biased, as tf.nn.moments
  save_mean[c] = mean;
  save_invstd[c] = 1.f / sqrtf(var + eps);
  if (moving_mean) {
    // assign_moving_average: v -= (v - value) * (1 - momentum)
    const float om = 1.f - momentum;
    moving_mean[c] = moving_mean[c] - (moving_mean[c] - mean) * om;
    moving_var[c] = moving_var[c] - (moving_var[c] - var) * om;
  }
}

__global__ void __launch_bounds__(kBlock)
bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ gamma,
                const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
                int64_t n, int N, int use_bn, int act, float* __restrict__ y) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int c = static_cast<int>(i % N);
  float v = x[i] + (bias ? bias[c] : 0.f);
  if (use_bn) {
    v = (v - mean[c]) * invstd[c];
    v = v * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
  }
  if (act == ER_ACT_RELU) v = v > 0.f ? v : 0.f;
  y[i] = v;
}

typedef float f32x4d __attribute__((ext_vector_type(4)));

// one element of bias + BatchNorm + activation
__device__ __forceinline__ float bn_act_one(float x, float bv, float mu, float is, float ga, float be, int act) {
  float v = ((x + bv) - mu) * is;
  v = v * ga + be;
  if (act == ER_ACT_RELU) v = v > 0.f ? v : 0.f;
  return v;
}

// use_bn == ER_BN_FROZEN: batch_normalization(training=False) inside a training graph.  The reference's MMoE and
// DBMTL models build their experts that way (model/mmoe.py:37-47, model/dbmtl.py:66-70 call layers/mmoe.py MMOE
// without is_training, whose default is False): the MOVING statistics normalise, nothing updates them, gamma / beta /
// the bias and the input still receive gradients.  save_mean / save_invstd are written for the backward.
__device__ __forceinline__ void bn_frozen_apply_body(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ gamma,
                       const float* __restrict__ beta, const float* __restrict__ moving_mean,
                       const float* __restrict__ moving_var, int64_t n, int N, float eps, int act,
                       float* __restrict__ y, float* __restrict__ save_mean, float* __restrict__ save_invstd, int bx, int by) {
  const int64_t i = static_cast<int64_t>(bx) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int c = static_cast<int>(i % N);
  const float mu = moving_mean[c];
  const float is = 1.f / sqrtf(moving_var[c] + eps);
  if (i < N) {  // (row 0: one writer per column)
    save_mean[c] = mu;
    save_invstd[c] = is;
  }
  float v = ((x[i] + (bias ? bias[c] : 0.f)) - mu) * is;
  v = v * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
  if (act == ER_ACT_RELU) v = v > 0.f ? v : 0.f;
  y[i] = v;
}

__global__ void __launch_bounds__(kBlock)
bn_frozen_apply_kernel(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ gamma,
                       const float* __restrict__ beta, const float* __restrict__ moving_mean,
                       const float* __restrict__ moving_var, int64_t n, int N, float eps, int act,
                       float* __restrict__ y, float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  bn_frozen_apply_body(x, bias, gamma, beta, moving_mean, moving_var, n, N, eps, act, y, save_mean, save_invstd, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y));
}


// Fused finalize + apply.  grid = (column blocks of 64, row blocks of kApplyRows); every workgroup first
// merges the `chunks` Welford partials of ITS 64 columns (4 row-lanes x chunks/4 each, then a fixed merge:
// identical bits in every workgroup), keeps mean / invstd in LDS, then normalises its [rows x 64] tile.
// Row block 0 also writes save_mean / save_invstd and the moving statistics.  One launch instead of two.
// 16 rows per workgroup: the kernels are latency-bound (dependent loads: partials -> coefficients -> tile), so what
// matters is workgroups in flight - measured on one box, whole DeepFM step: 64 rows 0.591 ms, 32 0.562, 16 0.553,
// 8 0.569, 4 0.605 (the redundant partial merges start to cost).
constexpr int kApplyRows = 16;
inline int apply_tiles_per_block(int B) {  // (workgroups in flight ~ constant)
  static const int mid = [] {  // (A/B knob: row tiles per workgroup for 8192 <= B <= 32768)
    const char* e = getenv("ER_BN_TILES_MID");
    const int v = e ? atoi(e) : 0;
    return v >= 1 ? v : 4;  // (round 6, under the pooled merge: MMoE 25 M 2.042 ms at 2, 1.998 at 4, 1.993 at 8)
  }();
  return B > 32768 ? 16 : (B >= 8192 ? mid : 1);
}

// Tall activations (DIN's attention MLP: B x L = 204,800 rows -> 3,200 row-tile partials per column): the fused
// finalize + apply kernels below re-merge ALL partials in every 16-row workgroup (cheap for the 64 partials of a
// B = 4096 layer, 2.4 MB per workgroup here: bn_finalize_apply took 0.58 ms on average in the DIN step).  Above
// kInlineChunks partials they are merged ONCE per column by these kernels (same lane layout and merge order as the
// fused kernels' first phase) into kMergeSlices records by as many workgroups per 64 columns, and the fused kernel merges
// those.
constexpr int kInlineChunks = 256;
constexpr int kMergeSlices = 64;

__global__ void __launch_bounds__(kBlock)
bn_stats_merge_kernel(const float* __restrict__ partial, int N, int chunks, int per_slice, float* __restrict__ merged) {
  // workgroup (x, y): columns x * 64 .., partials [y * per_slice, (y + 1) * per_slice) -> merged[y][column]
  __shared__ Welford sm[kRowLanes][kColsPerBlock];
  const int cl = threadIdx.x % kColsPerBlock;
  const int rl = threadIdx.x / kColsPerBlock;
  const int c = blockIdx.x * kColsPerBlock + cl;
  const int k_end = min(chunks, static_cast<int>(blockIdx.y + 1) * per_slice);
  Welford t{0.f, 0.f, 0.f};
  if (c < N) {
    // four partials per trip, requested together from clamped addresses and merged in order (one load per trip in a loop of
    // data-dependent length was ~13 dependent round trips per lane for DIN's 3,200 row tiles: 7 - 8 us per launch)
    for (int k0 = blockIdx.y * per_slice + rl; k0 < k_end; k0 += 4 * kRowLanes) {
      Welford w4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j * kRowLanes;
        const float* p = partial + (static_cast<int64_t>(k < k_end ? k : k_end - 1) * N + c) * 3;
        w4[j] = Welford{p[0], p[1], p[2]};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (k0 + j * kRowLanes < k_end) t = wf_merge(t, w4[j]);
      }
    }
  }
  sm[rl][cl] = t;
  __syncthreads();
  if (rl == 0 && c < N) {
    const Welford w = wf_merge(wf_merge(sm[0][cl], sm[1][cl]), wf_merge(sm[2][cl], sm[3][cl]));
    float* o = merged + (static_cast<int64_t>(blockIdx.y) * N + c) * 3;
    o[0] = w.n; o[1] = w.mean; o[2] = w.m2;
  }
}

__global__ void __launch_bounds__(kBlock)
bn_bwd_merge_kernel(const float* __restrict__ partial, int N, int chunks, int per_slice, float* __restrict__ merged) {
  __shared__ float sm[2][kRowLanes][kColsPerBlock];
  const int cl = threadIdx.x % kColsPerBlock;
  const int rl = threadIdx.x / kColsPerBlock;
  const int c = blockIdx.x * kColsPerBlock + cl;
  const int k_end = min(chunks, static_cast<int>(blockIdx.y + 1) * per_slice);
  float a = 0.f, b = 0.f;
  if (c < N) {
    for (int k0 = blockIdx.y * per_slice + rl; k0 < k_end; k0 += 4 * kRowLanes) {  // (four per trip: see bn_stats_merge_kernel)
      float pa[4], pb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j * kRowLanes;
        const float* p = partial + (static_cast<int64_t>(k < k_end ? k : k_end - 1) * N + c) * 2;
        pa[j] = p[0];
        pb[j] = p[1];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (k0 + j * kRowLanes < k_end) {
          a = a + pa[j];
          b = b + pb[j];
        }
      }
    }
  }
  sm[0][rl][cl] = a;
  sm[1][rl][cl] = b;
  __syncthreads();
  if (rl == 0 && c < N) {
    float* o = merged + (static_cast<int64_t>(blockIdx.y) * N + c) * 2;
    o[0] = (sm[0][0][cl] + sm[0][1][cl]) + (sm[0][2][cl] + sm[0][3][cl]);
    o[1] = (sm[1][0][cl] + sm[1][1][cl]) + (sm[1][2][cl] + sm[1][3][cl]);
  }
}

__device__ __forceinline__ void bn_finalize_apply_body(const float* __restrict__ partial, const float* __restrict__ x, const float* __restrict__ bias,
                         const float* __restrict__ gamma, const float* __restrict__ beta, int B, int N, int chunks,
                         float eps, float momentum, float* __restrict__ moving_mean, float* __restrict__ moving_var,
                         int act, float* __restrict__ y, float* __restrict__ save_mean,
                         float* __restrict__ save_invstd, int tiles_per_block, int bx, int by,
                         uint16_t* __restrict__ yb = nullptr, int ldyb = 0, float* __restrict__ y2 = nullptr, int ldy2 = 0,
                         bool apply = true) {
  // apply == false (bn_finalize_stats_kernel): the statistics only - mean / invstd / moving statistics; x and y are not touched
  // yb: a bf16 copy of y (row stride ldyb) for the contraction that reads it next (dense_dtype 'bf16': no cast launch)
  // y2: a second fp32 copy of y at row stride ldy2 - the layer's column block of a concat (er_bn_apply_wide_fm)
  __shared__ Welford sm[kRowLanes][kColsPerBlock];
  __shared__ float s_mean[kColsPerBlock], s_inv[kColsPerBlock];
  const int cl = threadIdx.x % kColsPerBlock;
  const int rl = threadIdx.x / kColsPerBlock;
  const int c = bx * kColsPerBlock + cl;
  // batch-sized layers (one row tile per workgroup): the tile's activations are requested BEFORE the partials are merged -
  // the two round trips overlap instead of following each other (the kernel is a chain of dependent loads, not bytes)
  constexpr int kPre = kApplyRows / kRowLanes;
  float xpre[kPre];
  const bool pre = apply && tiles_per_block == 1 && c < N;
  // (requesting the column's bias / gamma / beta here too, instead of behind the merge's barrier, measured SLOWER: BatchNorm
  // family 64.0 -> 65.5 us on DeepFM, 439 -> 455 on MMoE, profiles/r06_s11_bn_coefficients_first_rejected_ab_lines.txt)
  if (pre) {
    // branch-free (rows past the end read the last row and are never stored): loads inside per-row branches are waited
    // for inside them
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
      int r = by * kApplyRows + rl + k * kRowLanes;
      r = r < B ? r : B - 1;
      xpre[k] = x[static_cast<int64_t>(r) * N + c];
    }
  }
  // The partials (n_i, mean_i, M2_i) of a lane group are pooled around a pivot p (the group's first mean) without a division
  // per partial:  S0 = sum n_i, S1 = sum n_i (mean_i - p), S2 = sum M2_i + n_i (mean_i - p)^2;
  //               mean = p + S1 / S0, M2 = S2 - S1^2 / S0
  // (exact algebra of the pooled variance; the chunk means lie within a chunk-mean standard error of p, so the subtraction
  // cancels ~1/64 of S2).  Chan's pairwise update - two IEEE divisions per partial - cost 800 VALU instructions per wave on a
  // B = 4096 layer, twice that at B = 8192, in EVERY 16-row workgroup (profiles/r06_s6_sq_counters_by_kernel.json).
  // Fixed order: lane group rl pools partials rl, rl + 4, ...; the four groups are merged (Chan) as (0 + 1) + (2 + 3).
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, pivot = 0.f;
  constexpr int kGroup = 16;
  if (c < N) {
    // groups of 16 partials (a B = 4096 layer's 64 row tiles: ONE round trip per lane): the 48 loads of a group are issued together, then pooled in order
    for (int k0 = rl; k0 < chunks; k0 += kGroup * kRowLanes) {
      Welford w8[kGroup];
#pragma unroll
      for (int j = 0; j < kGroup; ++j) {
        // branch-free: a partial past the end is loaded from the last one's address and its count / M2 zeroed afterwards
        // (an `if (k < chunks)` around the load made the compiler wait for every load inside its branch: 16 dependent
        // L2 round trips per lane on a B = 4096 layer - s_waitcnt vmcnt(0) behind each global_load_dwordx3 in the ISA)
        const int k = k0 + j * kRowLanes;
        const int kc = k < chunks ? k : chunks - 1;
        const float* p = partial + (static_cast<int64_t>(kc) * N + c) * 3;
        w8[j] = Welford{p[0], p[1], p[2]};
      }
#pragma unroll
      for (int j = 0; j < kGroup; ++j) {
        if (k0 + j * kRowLanes >= chunks) { w8[j].n = 0.f; w8[j].m2 = 0.f; }
      }
      if (k0 == rl) pivot = w8[0].mean;
#pragma unroll
      for (int j = 0; j < kGroup; ++j) {
        const float d = w8[j].mean - pivot;
        const float nd = w8[j].n * d;
        s0 = s0 + w8[j].n;
        s1 = s1 + nd;
        s2 = s2 + (w8[j].m2 + nd * d);
      }
    }
  }
  {
    const float shift = s0 > 0.f ? s1 / s0 : 0.f;
    sm[rl][cl] = Welford{s0, pivot + shift, fmaxf(s2 - s1 * shift, 0.f)};
  }
  __syncthreads();
  if (rl == 0 && c < N) {
    const Welford w = wf_merge(wf_merge(sm[0][cl], sm[1][cl]), wf_merge(sm[2][cl], sm[3][cl]));
    const float mean = w.mean;
    const float var = w.m2 / static_cast<float>(B);  // biased, as tf.nn.moments
    const float inv = 1.f / sqrtf(var + eps);
    s_mean[cl] = mean;
    s_inv[cl] = inv;
    if (by == 0) {
      save_mean[c] = mean;
      save_invstd[c] = inv;
      if (moving_mean) {
        const float om = 1.f - momentum;
        moving_mean[c] = moving_mean[c] - (moving_mean[c] - mean) * om;
        moving_var[c] = moving_var[c] - (moving_var[c] - var) * om;
      }
    }
  }
  __syncthreads();
  if (!apply) return;
  if (tiles_per_block >= 4 && N % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    // tall activations (DIN's [B x L]-row layers: 16 row tiles per workgroup): 16-byte lanes - 16 column lanes x 16 row
    // lanes, one load per lane and tile, four tiles in flight.  Measured (profiles/r03_bn_lanes.md): 250 -> 210 us per DIN
    // step for this kernel; for batch-sized layers (1 - 2 tiles per workgroup) the wider lanes were SLOWER (MMoE 205 ->
    // 233 us, DeepFM 46 -> 47) - those workgroups live on the latency of their three dependent round trips, not on
    // request width - and the backward and frozen kernels gained nothing: they keep 4-byte lanes.
    const int cl4 = (threadIdx.x & 15) * 4, rl16 = threadIdx.x >> 4;
    const int c4 = bx * kColsPerBlock + cl4;
    if (c4 >= N) return;
    float mu[4], is[4], bv[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mu[j] = s_mean[cl4 + j];
      is[j] = s_inv[cl4 + j];
      bv[j] = bias ? bias[c4 + j] : 0.f;
      ga[j] = gamma ? gamma[c4 + j] : 1.f;
      be[j] = beta ? beta[c4 + j] : 0.f;
    }
    const int row_base = by * tiles_per_block * kApplyRows + rl16;
    for (int tile0 = 0; tile0 < tiles_per_block; tile0 += 4) {
      f32x4d xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = row_base + (tile0 + u) * kApplyRows;
        const bool ok = tile0 + u < tiles_per_block && r < B;
        xv[u] = ok ? *reinterpret_cast<const f32x4d*>(x + static_cast<int64_t>(r) * N + c4) : f32x4d{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = row_base + (tile0 + u) * kApplyRows;
        if (tile0 + u < tiles_per_block && r < B) {
          f32x4d out;
#pragma unroll
          for (int j = 0; j < 4; ++j) out[j] = bn_act_one(xv[u][j], bv[j], mu[j], is[j], ga[j], be[j], act);
          *reinterpret_cast<f32x4d*>(y + static_cast<int64_t>(r) * N + c4) = out;
          if (y2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y2[static_cast<int64_t>(r) * ldy2 + c4 + j] = out[j];
          }
          if (yb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) yb[static_cast<int64_t>(r) * ldyb + c4 + j] = f32_to_bf16_bits(out[j]);
          }
        }
      }
    }
    return;
  }
  if (c >= N) return;
  const float mu = s_mean[cl], is = s_inv[cl];
  const float bv = bias ? bias[c] : 0.f;
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  // tiles_per_block row tiles per workgroup (1 for batch-sized layers; tall activations - DIN's B x L rows - amortise
  // the partial merge above over 16 tiles)
  for (int tile = 0; tile < tiles_per_block; ++tile) {
    const int r0 = (by * tiles_per_block + tile) * kApplyRows + rl;
    if (r0 - rl >= B) break;
    // fixed trip count, fully unrolled: all the tile's loads of a lane are in flight together
    float xv[kApplyRows / kRowLanes];
#pragma unroll
    for (int k = 0; k < kApplyRows / kRowLanes; ++k) {
      const int r = r0 + k * kRowLanes;
      xv[k] = pre ? xpre[k] : ((r < B) ? x[static_cast<int64_t>(r) * N + c] : 0.f);
    }
#pragma unroll
    for (int k = 0; k < kApplyRows / kRowLanes; ++k) {
      const int r = r0 + k * kRowLanes;
      if (r < B) {
        const float o = bn_act_one(xv[k], bv, mu, is, ga, be, act);
        y[static_cast<int64_t>(r) * N + c] = o;
        if (y2) y2[static_cast<int64_t>(r) * ldy2 + c] = o;
        if (yb) yb[static_cast<int64_t>(r) * ldyb + c] = f32_to_bf16_bits(o);
      }
    }
  }
}

__global__ void __launch_bounds__(kBlock)
bn_finalize_apply_kernel(const float* __restrict__ partial, const float* __restrict__ x, const float* __restrict__ bias,
                         const float* __restrict__ gamma, const float* __restrict__ beta, int B, int N, int chunks,
                         float eps, float momentum, float* __restrict__ moving_mean, float* __restrict__ moving_var,
                         int act, float* __restrict__ y, float* __restrict__ save_mean,
                         float* __restrict__ save_invstd, int tiles_per_block, uint16_t* __restrict__ yb, int ldyb) {
  bn_finalize_apply_body(partial, x, bias, gamma, beta, B, N, chunks, eps, momentum, moving_mean, moving_var, act, y, save_mean, save_invstd, tiles_per_block, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), yb, ldyb);
}

// the finalize half alone (er_bn_finalize_from_stats): the layer's apply runs inside the NEXT contraction's staging
// (er_gemm_f32_bn_a); same merge, same order, same bits as bn_finalize_apply_kernel's first phase
__global__ void __launch_bounds__(kBlock)
bn_finalize_stats_kernel(const float* __restrict__ partial, int B, int N, int chunks, float eps, float momentum,
                   float* __restrict__ moving_mean, float* __restrict__ moving_var, float* __restrict__ save_mean,
                   float* __restrict__ save_invstd) {
  bn_finalize_apply_body(partial, nullptr, nullptr, nullptr, nullptr, B, N, chunks, eps, momentum, moving_mean, moving_var, 0,
                         nullptr, save_mean, save_invstd, 1, static_cast<int>(blockIdx.x), 0, nullptr, 0, nullptr, 0, false);
}


// The LAST BatchNorm finalize + apply of DeepFM's deep tower together with what joins its output (reference
// model/deepfm.py:60-83): workgroups [0, bn_blocks) run bn_finalize_apply_body and store the activations twice - as the
// layer's own output y and as the column block [1 + D, 1 + D + N) of out = [sum(wide) | FM | deep] - the workgroups behind
// them the FM and wide row-sum bodies of er_wide_fm_concat, which depend on the embeddings only.  One launch for two
// (the concat launch and the 1.45 us boundary in front of it), the same arithmetic in the same order.
struct WideFmArgs {
  const float* wide; int n_w, ld_w;
  const float* fm_x; int F, D, ld_x;
  float* out; int ld_out;
  float* sum_out;
  int fm_blocks;
};
template <int V>
__global__ void __launch_bounds__(kBlock)
bn_apply_wide_fm_kernel(const float* __restrict__ partial, const float* __restrict__ x, const float* __restrict__ bias,
                        const float* __restrict__ gamma, const float* __restrict__ beta, int B, int N, int chunks,
                        float eps, float momentum, float* __restrict__ moving_mean, float* __restrict__ moving_var,
                        int act, float* __restrict__ y, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                        int tiles_per_block, int gx, int bn_blocks, WideFmArgs w) {
  const int bid = blockIdx.x;
  if (bid < bn_blocks) {
    bn_finalize_apply_body(partial, x, bias, gamma, beta, B, N, chunks, eps, momentum, moving_mean, moving_var, act, y,
                           save_mean, save_invstd, tiles_per_block, bid % gx, bid / gx, nullptr, 0, w.out + 1 + w.D, w.ld_out);
  } else if (bid < bn_blocks + w.fm_blocks) {
    fm_fwd_body<V>(static_cast<int64_t>(bid - bn_blocks) * kBlock + threadIdx.x, w.fm_x, B, w.F, w.D, w.ld_x, w.out + 1,
                   w.ld_out, w.sum_out);
  } else {
    rowsum_fwd_body(static_cast<int64_t>(bid - bn_blocks - w.fm_blocks) * kBlock + threadIdx.x, w.wide, B, w.n_w, w.ld_w,
                    w.out, w.ld_out);
  }
}

// backward partials: p[(chunk*N + c)*2 + {0,1}] = (sum g, sum g*xhat), g = dy * act'(y)
__device__ __forceinline__ void bn_bwd_partial_body(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ y,
                      const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ dy,
                      int B, int N, int chunks, int use_bn, int act, float* __restrict__ partial, int dy_ld, int bx, int by,
                      const float* __restrict__ gamma = nullptr, float* __restrict__ dx = nullptr) {
  // dx != nullptr (use_bn == ER_BN_FROZEN only): on the moving statistics dx = gamma * invstd * g depends on no column sum,
  // so this pass - which holds g already - writes it, and the finalize launch that follows only merges the sums into the
  // parameter gradients (er_bn_bwd_multi: the experts of a multi-task model, one pass over dy / y / z instead of two)
  __shared__ float sm[2][kRowLanes][kColsPerBlock];
  const int cl = threadIdx.x % kColsPerBlock;
  const int rl = threadIdx.x / kColsPerBlock;
  const int c = bx * kColsPerBlock + cl;
  const int chunk = by;
  const int rows_per_chunk = static_cast<int>(ceil_div(B, chunks));
  const int r0 = chunk * rows_per_chunk;
  const int r1 = (r0 + rows_per_chunk < B) ? r0 + rows_per_chunk : B;
  float sg = 0.f, sgx = 0.f;
  if (c < N) {
    const float bv = bias ? bias[c] : 0.f;
    const float mu = use_bn ? mean[c] : 0.f;
    const float is = use_bn ? invstd[c] : 0.f;
    // 8 of the lane's rows per trip (a row past the chunk's end reads the last one and is skipped), summed in row order: the
    // one-row-per-trip loop was a chain of rows_per_chunk / 4 dependent round trips with three loads in flight
    const float* xs = use_bn ? x : dy;  // (absent operands read dy: valid addresses, values dropped)
    const float* ys = act == ER_ACT_RELU ? y : dy;
    const int64_t xs_ld = use_bn ? N : dy_ld, ys_ld = act == ER_ACT_RELU ? N : dy_ld;
    constexpr int kRows = 8;
    const float ga = (dx && gamma) ? gamma[c] : 1.f;
    for (int rb = r0 + rl; rb < r1; rb += kRows * kRowLanes) {
      float gv[kRows], yv[kRows], xv[kRows];
#pragma unroll
      for (int u = 0; u < kRows; ++u) {
        const int r = rb + u * kRowLanes < r1 ? rb + u * kRowLanes : r1 - 1;
        gv[u] = dy[static_cast<int64_t>(r) * dy_ld + c];
        yv[u] = ys[static_cast<int64_t>(r) * ys_ld + c];
        xv[u] = xs[static_cast<int64_t>(r) * xs_ld + c];
      }
#pragma unroll
      for (int u = 0; u < kRows; ++u) {
        if (rb + u * kRowLanes < r1) {
          float g = gv[u];
          if (act == ER_ACT_RELU && !(yv[u] > 0.f)) g = 0.f;
          sg = sg + g;
          if (use_bn) sgx = sgx + g * ((xv[u] + bv - mu) * is);
          if (dx) dx[static_cast<int64_t>(rb + u * kRowLanes) * N + c] = ga * is * g;  // (bn_bwd_finalize_apply_body's frozen form)
        }
      }
    }
  }
  sm[0][rl][cl] = sg;
  sm[1][rl][cl] = sgx;
  __syncthreads();
  if (rl == 0 && c < N) {
    float* p = partial + (static_cast<int64_t>(chunk) * N + c) * 2;
    p[0] = (sm[0][0][cl] + sm[0][1][cl]) + (sm[0][2][cl] + sm[0][3][cl]);
    p[1] = (sm[1][0][cl] + sm[1][1][cl]) + (sm[1][2][cl] + sm[1][3][cl]);
  }
}

__global__ void __launch_bounds__(kBlock)
bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ y,
                      const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ dy,
                      int B, int N, int chunks, int use_bn, int act, float* __restrict__ partial, int dy_ld) {
  bn_bwd_partial_body(x, bias, y, mean, invstd, dy, B, N, chunks, use_bn, act, partial, dy_ld, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y));
}


// Fused finalize + apply of the backward: every workgroup reduces the partials of its 64 columns (fixed order),
// row block 0 writes / accumulates dgamma, dbeta (dbias without BatchNorm), then dx for its [rows x 64] tile.
__device__ __forceinline__ void bn_bwd_finalize_apply_body(const float* __restrict__ partial, const float* __restrict__ x,
                             const float* __restrict__ bias, const float* __restrict__ gamma,
                             const float* __restrict__ y, const float* __restrict__ mean,
                             const float* __restrict__ invstd, const float* __restrict__ dy, int B, int N,
                             int chunks, int use_bn, int act, int accumulate, float* __restrict__ dx,
                             float* __restrict__ dbias, float* __restrict__ dgamma, float* __restrict__ dbeta,
                             int dy_ld, int tiles_per_block, int bx, int by, uint16_t* __restrict__ dxb = nullptr,
                             int lddxb = 0, bool apply = true) {
  // apply == false: the parameter gradients only (row block 0) - dx was written by bn_bwd_partial_body (frozen statistics)
  // dxb: a bf16 copy of dx (row stride lddxb) for the input-gradient contraction that reads it next
  __shared__ float sm[2][kRowLanes][kColsPerBlock];
  __shared__ float s_g[kColsPerBlock], s_gx[kColsPerBlock];
  const int cl = threadIdx.x % kColsPerBlock;
  const int rl = threadIdx.x / kColsPerBlock;
  const int c = bx * kColsPerBlock + cl;
  // one row tile per workgroup: dy / x / y of the tile are requested before the partial sums are merged (see the forward)
  constexpr int kPre = kApplyRows / kRowLanes;
  float gpre[kPre], xpre[kPre], ypre[kPre];
  const bool pre = apply && tiles_per_block == 1 && c < N;
  if (pre) {
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
      const int r = by * kApplyRows + rl + k * kRowLanes;
      const int64_t i = static_cast<int64_t>(r) * N + c;
      const bool ok = r < B;
      gpre[k] = ok ? dy[static_cast<int64_t>(r) * dy_ld + c] : 0.f;
      xpre[k] = (ok && use_bn) ? x[i] : 0.f;
      ypre[k] = (ok && act == ER_ACT_RELU) ? y[i] : 1.f;
    }
  }
  float a = 0.f, b = 0.f;
  if (c < N) {
    // (16 partials per trip with clamped addresses, as in the forward, measured no faster on DeepFM and SLOWER on the taller
    // layers - MMoE's BatchNorm family 439 -> 452 us, profiles/r06_s12_bn_bwd_fm_batched_loads_ab_lines.txt: these two adds
    // per partial never waited the way the forward's Chan merge did)
#pragma unroll 8
    for (int k = rl; k < chunks; k += kRowLanes) {
      const float* p = partial + (static_cast<int64_t>(k) * N + c) * 2;
      a = a + p[0];
      b = b + p[1];
    }
  }
  sm[0][rl][cl] = a;
  sm[1][rl][cl] = b;
  __syncthreads();
  if (rl == 0 && c < N) {
    a = (sm[0][0][cl] + sm[0][1][cl]) + (sm[0][2][cl] + sm[0][3][cl]);
    b = (sm[1][0][cl] + sm[1][1][cl]) + (sm[1][2][cl] + sm[1][3][cl]);
    s_g[cl] = a;
    s_gx[cl] = b;
    if (by == 0) {
      if (use_bn) {
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + a : a;
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + b : b;
        if (use_bn == ER_BN_FROZEN) {  // fixed statistics: the bias sees the column sum of dx
          if (dbias) {
            const float d = (gamma ? gamma[c] : 1.f) * invstd[c] * a;
            dbias[c] = accumulate ? dbias[c] + d : d;
          }
        } else if (dbias && !accumulate) {
          dbias[c] = 0.f;  // BatchNorm removes the column mean: d(loss)/d(bias) == 0
        }
      } else {
        if (dbias) dbias[c] = accumulate ? dbias[c] + a : a;
      }
    }
  }
  __syncthreads();
  if (!apply) return;
  if (!pre && N <= kColsPerBlock / 2 && (N & (N - 1)) == 0) {
    // a NARROW tall layer (DIN's attention MLP ends 64 -> 32 -> 1 over B x L rows): with one lane per column, 64 columns per
    // workgroup, only N of every 64 lanes had work - the [204800, 1] layer's pass took 23.7 us for 2.4 MB.  The elementwise
    // half is remapped: kBlock / N row lanes, every lane busy, four rows in flight per lane; same formula, same bits.
    const int c2 = threadIdx.x & (N - 1);
    const int rstep = kBlock / N;
    const int row_lo = by * tiles_per_block * kApplyRows;
    int row_hi = row_lo + tiles_per_block * kApplyRows;
    row_hi = row_hi < B ? row_hi : B;
    const float sg2 = s_g[c2], sgx2 = s_gx[c2];
    const float bv2 = bias ? bias[c2] : 0.f;
    const float mu2 = use_bn ? mean[c2] : 0.f, is2 = use_bn ? invstd[c2] : 0.f;
    const float ga2 = gamma ? gamma[c2] : 1.f;
    const float invB2 = 1.f / static_cast<float>(B);
    for (int r0 = row_lo + threadIdx.x / N; r0 < row_hi; r0 += 4 * rstep) {
      float gv[4], yv[4], xv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int r = r0 + k * rstep;
        r = r < row_hi ? r : row_hi - 1;  // (clamped, never stored: a load inside a branch is waited for inside it)
        const int64_t i = static_cast<int64_t>(r) * N + c2;
        gv[k] = dy[static_cast<int64_t>(r) * dy_ld + c2];
        xv[k] = use_bn ? x[i] : 0.f;
        yv[k] = act == ER_ACT_RELU ? y[i] : 1.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = r0 + k * rstep;
        if (r < row_hi) {
          float g = gv[k];
          if (act == ER_ACT_RELU && !(yv[k] > 0.f)) g = 0.f;
          if (use_bn) {
            const float xh = (xv[k] + bv2 - mu2) * is2;
            g = (use_bn == ER_BN_FROZEN) ? ga2 * is2 * g : ga2 * is2 * (g - sg2 * invB2 - xh * (sgx2 * invB2));
          }
          dx[static_cast<int64_t>(r) * N + c2] = g;
          if (dxb) dxb[static_cast<int64_t>(r) * lddxb + c2] = f32_to_bf16_bits(g);
        }
      }
    }
    return;
  }
  if (c >= N) return;
  const float sg = s_g[cl], sgx = s_gx[cl];
  const float bv = bias ? bias[c] : 0.f;
  const float mu = use_bn ? mean[c] : 0.f, is = use_bn ? invstd[c] : 0.f;
  const float ga = gamma ? gamma[c] : 1.f;
  const float invB = 1.f / static_cast<float>(B);
  constexpr int kIter = kApplyRows / kRowLanes;
  for (int tile = 0; tile < tiles_per_block; ++tile) {
    const int r0 = (by * tiles_per_block + tile) * kApplyRows + rl;
    if (r0 - rl >= B) break;
    float gv[kIter], yv[kIter], xv[kIter];
#pragma unroll
    for (int k = 0; k < kIter; ++k) {
      const int r = r0 + k * kRowLanes;
      const int64_t i = static_cast<int64_t>(r) * N + c;
      const bool ok = r < B;
      gv[k] = pre ? gpre[k] : (ok ? dy[static_cast<int64_t>(r) * dy_ld + c] : 0.f);
      xv[k] = pre ? xpre[k] : ((ok && use_bn) ? x[i] : 0.f);
      yv[k] = (ok && act == ER_ACT_RELU) ? (pre ? ypre[k] : y[i]) : 1.f;
    }
#pragma unroll
    for (int k = 0; k < kIter; ++k) {
      const int r = r0 + k * kRowLanes;
      if (r < B) {
        float g = gv[k];
        if (act == ER_ACT_RELU && !(yv[k] > 0.f)) g = 0.f;
        if (use_bn) {
          const float xh = (xv[k] + bv - mu) * is;
          g = (use_bn == ER_BN_FROZEN) ? ga * is * g : ga * is * (g - sg * invB - xh * (sgx * invB));
        }
        dx[static_cast<int64_t>(r) * N + c] = g;
        if (dxb) dxb[static_cast<int64_t>(r) * lddxb + c] = f32_to_bf16_bits(g);
      }
    }
  }
}

__global__ void __launch_bounds__(kBlock)
bn_bwd_finalize_apply_kernel(const float* __restrict__ partial, const float* __restrict__ x,
                             const float* __restrict__ bias, const float* __restrict__ gamma,
                             const float* __restrict__ y, const float* __restrict__ mean,
                             const float* __restrict__ invstd, const float* __restrict__ dy, int B, int N,
                             int chunks, int use_bn, int act, int accumulate, float* __restrict__ dx,
                             float* __restrict__ dbias, float* __restrict__ dgamma, float* __restrict__ dbeta,
                             int dy_ld, int tiles_per_block, uint16_t* __restrict__ dxb, int lddxb) {
  bn_bwd_finalize_apply_body(partial, x, bias, gamma, y, mean, invstd, dy, B, N, chunks, use_bn, act, accumulate, dx, dbias, dgamma, dbeta, dy_ld, tiles_per_block, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), dxb, lddxb);
}


// ------------------------------------------------------------------------------------------------
// The bias / BatchNorm / activation kernels of SEVERAL layer outputs in ONE launch (er_bn_fwd_multi / er_bn_bwd_multi):
// the same-depth layers of parallel stacks (MMoE's experts and task towers) each own a few-microsecond launch that fills
// a fraction of the chip; workgroups [start[i], start[i + 1]) run layer i's body - the bodies of the single-layer
// kernels above, same arithmetic, same bits.
// ------------------------------------------------------------------------------------------------
constexpr int kBnMulti = 8;
struct BnItem {
  const float* x; const float* bias; const float* gamma; const float* beta;
  float* moving_mean; float* moving_var;
  const float* partial;   // forward: Welford column statistics [chunks][N][3]; backward: column sums [chunks][N][2]
  float* y; float* save_mean; float* save_invstd;
  const float* yin;       // backward: the layer's activation output
  const float* dy; float* dx; float* dbias; float* dgamma; float* dbeta;
  float* scratch;         // backward phase 1 output (when the sums are computed here)
  int B, N, chunks, mode, act, dy_ld, accumulate, tpb, gx;
  float eps, momentum;
  int dx_in_partial;      // backward, frozen statistics: phase 1 writes dx, phase 2 (one row block) the parameter gradients only
};
struct BnMultiArgs {
  int n;
  int start[kBnMulti + 1];
  BnItem d[kBnMulti];
};

__device__ __forceinline__ int bn_multi_find(const BnMultiArgs& a, int b) {
  int i = 0;
  while (i + 1 < a.n && b >= a.start[i + 1]) ++i;
  return i;
}

__global__ void __launch_bounds__(kBlock)
bn_fwd_multi_kernel(BnMultiArgs a) {
  const int i = bn_multi_find(a, blockIdx.x);
  const BnItem& d = a.d[i];
  const int local = blockIdx.x - a.start[i];
  if (d.mode == 1) {  // batch statistics from the GEMM's epilogue: finalise + normalise + activate
    bn_finalize_apply_body(d.partial, d.x, d.bias, d.gamma, d.beta, d.B, d.N, d.chunks, d.eps, d.momentum, d.moving_mean,
                           d.moving_var, d.act, d.y, d.save_mean, d.save_invstd, d.tpb, local % d.gx, local / d.gx);
  } else if (d.mode == ER_BN_FROZEN) {
    bn_frozen_apply_body(d.x, d.bias, d.gamma, d.beta, d.moving_mean, d.moving_var, static_cast<int64_t>(d.B) * d.N, d.N,
                         d.eps, d.act, d.y, d.save_mean, d.save_invstd, local, 0);
  } else {  // bias + activation only
    const int64_t idx = static_cast<int64_t>(local) * kBlock + threadIdx.x;
    if (idx < static_cast<int64_t>(d.B) * d.N) {
      float v = d.x[idx] + (d.bias ? d.bias[idx % d.N] : 0.f);
      if (d.act == ER_ACT_RELU) v = v > 0.f ? v : 0.f;
      d.y[idx] = v;
    }
  }
}

__global__ void __launch_bounds__(kBlock)
bn_bwd_partial_multi_kernel(BnMultiArgs a) {
  const int i = bn_multi_find(a, blockIdx.x);
  const BnItem& d = a.d[i];
  const int local = blockIdx.x - a.start[i];
  bn_bwd_partial_body(d.x, d.bias, d.yin, d.save_mean, d.save_invstd, d.dy, d.B, d.N, d.chunks, d.mode, d.act, d.scratch,
                      d.dy_ld, local % d.gx, local / d.gx, d.gamma, d.dx_in_partial ? d.dx : nullptr);
}

__global__ void __launch_bounds__(kBlock)
bn_bwd_finalize_apply_multi_kernel(BnMultiArgs a) {
  const int i = bn_multi_find(a, blockIdx.x);
  const BnItem& d = a.d[i];
  const int local = blockIdx.x - a.start[i];
  bn_bwd_finalize_apply_body(d.partial, d.x, d.bias, d.gamma, d.yin, d.save_mean, d.save_invstd, d.dy, d.B, d.N, d.chunks,
                             d.mode, d.act, d.accumulate, d.dx, d.dbias, d.dgamma, d.dbeta, d.dy_ld, d.tpb,
                             local % d.gx, local / d.gx, nullptr, 0, !d.dx_in_partial);
}

// ------------------------------------------------------------------------------------------------
// generic column sum (rows strided over chunks, deterministic)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
colsum_partial_kernel(const float* __restrict__ x, int rows, int cols, int x_stride, int chunks,
                      float* __restrict__ partial) {
  __shared__ float sm[kRowLanes][kColsPerBlock];
  const int cl = threadIdx.x % kColsPerBlock;
  const int rl = threadIdx.x / kColsPerBlock;
  const int c = blockIdx.x * kColsPerBlock + cl;
  const int chunk = blockIdx.y;
  const int rows_per_chunk = static_cast<int>(ceil_div(rows, chunks));
  const int r0 = chunk * rows_per_chunk;
  const int r1 = (r0 + rows_per_chunk < rows) ? r0 + rows_per_chunk : rows;
  float s = 0.f;
  if (c < cols)
    for (int r = r0 + rl; r < r1; r += kRowLanes) s = s + x[static_cast<int64_t>(r) * x_stride + c];
  sm[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols)
    partial[static_cast<int64_t>(chunk) * cols + c] = (sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]);
}

__global__ void __launch_bounds__(kBlock)
colsum_finalize_kernel(const float* __restrict__ partial, int cols, int chunks, float* __restrict__ out, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (c >= cols) return;
  float a = 0.f;
  for (int k = lane; k < chunks; k += 64) a = a + partial[static_cast<int64_t>(k) * cols + c];
  a = wave_sum(a);
  if (lane == 0) out[c] = accumulate ? out[c] + a : a;
}

// dst[j] (+)= sum_p partial[p * ld + j] for several jobs in one launch (er_colsum_partials_multi): one wave per column,
// colsum_finalize_kernel's order (lane k sums partials k, k + 64, ..; then the wave butterfly)
constexpr int kMaxColsumJobs = 16;
struct ColsumJobs {
  int n;
  int start[kMaxColsumJobs + 1];  // workgroups [start[i], start[i + 1]) own job i (4 columns per workgroup)
  er_tail_job j[kMaxColsumJobs];
};
__global__ void __launch_bounds__(kBlock)
colsum_partials_multi_kernel(ColsumJobs a, int accumulate) {
  int i = 0;
  while (i + 1 < a.n && static_cast<int>(blockIdx.x) >= a.start[i + 1]) ++i;
  const er_tail_job& q = a.j[i];
  const int lane = threadIdx.x & 63;
  const int c = (static_cast<int>(blockIdx.x) - a.start[i]) * (kBlock / 64) + (threadIdx.x >> 6);
  if (c >= q.n_cols) return;
  float s = 0.f;
  for (int k = lane; k < q.n_parts; k += 64) s = s + q.partial[static_cast<int64_t>(k) * q.ld + c];
  s = wave_sum(s);
  if (lane == 0) q.dst[c] = accumulate ? q.dst[c] + s : s;
}

// few columns (the bias gradient of a narrow head: dy [B, 1]): one workgroup per column, one launch
__global__ void __launch_bounds__(kBlock)
colsum_narrow_kernel(const float* __restrict__ x, int rows, int cols, int x_stride, float* __restrict__ out,
                     int accumulate) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float a = 0.f;
  // (the loads of 8 trips are issued together, the additions keep their order: 10 us -> a few for 8,192 rows)
  int r = threadIdx.x;
  for (; r + 7 * kBlock < rows; r += 8 * kBlock) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[static_cast<int64_t>(r + j * kBlock) * x_stride + c];
#pragma unroll
    for (int j = 0; j < 8; ++j) a = a + v[j];
  }
  for (; r < rows; r += kBlock) a = a + x[static_cast<int64_t>(r) * x_stride + c];
  const float s = block_sum_256(a, red);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + s : s;
}

// ... of several narrow matrices in ONE launch (er_colsum_narrow_multi: the bias gradients of a multi-task model's tower
// heads and gates, reference model/mmoe.py:56-68 - one 4 us launch each before): workgroup b sums column b - start[j] of job j
constexpr int kMaxNarrowJobs = 16;
struct NarrowJobs {
  int n;
  int start[kMaxNarrowJobs + 1];
  er_colsum_job j[kMaxNarrowJobs];
};
__global__ void __launch_bounds__(kBlock)
colsum_narrow_multi_kernel(NarrowJobs a, int accumulate) {
  __shared__ float red[4];
  int i = 0;
  while (i + 1 < a.n && static_cast<int>(blockIdx.x) >= a.start[i + 1]) ++i;
  const er_colsum_job& q = a.j[i];
  const float* __restrict__ x = q.x;
  const int rows = q.rows, x_stride = q.x_stride;
  const int c = blockIdx.x - a.start[i];
  float s = 0.f;
  int r = threadIdx.x;
  for (; r + 7 * kBlock < rows; r += 8 * kBlock) {  // (colsum_narrow_kernel's loop: the same order, the same bits)
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[static_cast<int64_t>(r + j * kBlock) * x_stride + c];
#pragma unroll
    for (int j = 0; j < 8; ++j) s = s + v[j];
  }
  for (; r < rows; r += kBlock) s = s + x[static_cast<int64_t>(r) * x_stride + c];
  const float t = block_sum_256(s, red);
  if (threadIdx.x == 0) q.out[c] = accumulate ? q.out[c] + t : t;
}

// ------------------------------------------------------------------------------------------------
// Dice
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

__global__ void __launch_bounds__(kBlock)
dice_apply_kernel(const float* __restrict__ x, const float* __restrict__ alpha, const float* __restrict__ mean,
                  const float* __restrict__ invstd, int64_t n, int N, float* __restrict__ y) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int c = static_cast<int>(i % N);
  const float v = x[i];
  const float p = sigmoidf_((v - mean[c]) * invstd[c]);
  y[i] = alpha[c] * (1.f - p) * v + p * v;
}

// partials for dice backward: (sum q, sum q*xhat, sum dalpha) with q = dy * x*(1-alpha) * p*(1-p)
__global__ void __launch_bounds__(kBlock)
dice_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ alpha, const float* __restrict__ mean,
                        const float* __restrict__ invstd, const float* __restrict__ dy, int B, int N, int chunks,
                        float* __restrict__ partial) {
  __shared__ float sm[3][kRowLanes][kColsPerBlock];
  const int cl = threadIdx.x % kColsPerBlock;
  const int rl = threadIdx.x / kColsPerBlock;
  const int c = blockIdx.x * kColsPerBlock + cl;
  const int chunk = blockIdx.y;
  const int rows_per_chunk = static_cast<int>(ceil_div(B, chunks));
  const int r0 = chunk * rows_per_chunk;
  const int r1 = (r0 + rows_per_chunk < B) ? r0 + rows_per_chunk : B;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (c < N) {
    const float al = alpha[c], mu = mean[c], is = invstd[c];
    for (int r = r0 + rl; r < r1; r += kRowLanes) {
      const int64_t i = static_cast<int64_t>(r) * N + c;
      const float v = x[i];
      const float xh = (v - mu) * is;
      const float p = sigmoidf_(xh);
      const float q = dy[i] * v * (1.f - al) * p * (1.f - p);
      s0 = s0 + q;
      s1 = s1 + q * xh;
      s2 = s2 + dy[i] * (1.f - p) * v;
    }
  }
  sm[0][rl][cl] = s0; sm[1][rl][cl] = s1; sm[2][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < N) {
    float* p = partial + (static_cast<int64_t>(chunk) * N + c) * 3;
    for (int k = 0; k < 3; ++k) p[k] = (sm[k][0][cl] + sm[k][1][cl]) + (sm[k][2][cl] + sm[k][3][cl]);
  }
}

__global__ void __launch_bounds__(kBlock)
dice_bwd_finalize_kernel(const float* __restrict__ partial, int N, int chunks, float* __restrict__ sums,
                         float* __restrict__ dalpha) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (c >= N) return;
  float a = 0.f, b = 0.f, d = 0.f;
  for (int k = lane; k < chunks; k += 64) {
    const float* p = partial + (static_cast<int64_t>(k) * N + c) * 3;
    a = a + p[0]; b = b + p[1]; d = d + p[2];
  }
  a = wave_sum(a); b = wave_sum(b); d = wave_sum(d);
  if (lane != 0) return;
  sums[c] = a;
  sums[N + c] = b;
  if (dalpha) dalpha[c] = d;
}

__global__ void __launch_bounds__(kBlock)
dice_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ alpha, const float* __restrict__ mean,
                      const float* __restrict__ invstd, const float* __restrict__ dy, const float* __restrict__ sums,
                      int64_t n, int B, int N, float* __restrict__ dx) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int c = static_cast<int>(i % N);
  const float v = x[i];
  const float xh = (v - mean[c]) * invstd[c];
  const float p = sigmoidf_(xh);
  const float al = alpha[c];
  const float g = dy[i];
  // y = v*(al + (1-al)*p): direct term + the path through p = sigmoid(BN(x))
  const float direct = g * (al + (1.f - al) * p);
  const float q = g * v * (1.f - al) * p * (1.f - p);  // d loss / d xhat
  const float invB = 1.f / static_cast<float>(B);
  const float through_bn = invstd[c] * (q - sums[c] * invB - xh * (sums[N + c] * invB));
  dx[i] = direct + through_bn;
}

// ------------------------------------------------------------------------------------------------
// loss + scalar reductions (single block, fixed tree -> deterministic)
// ------------------------------------------------------------------------------------------------
// (kCeBlock = 1024 threads, 16 waves: er_dense_tail.h) sums combine in a fixed order
__device__ __forceinline__ float block_sum_1024(float v, float* smem16) {
  v = wave_sum(v);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) smem16[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < kCeBlock / 64; ++i) r = r + smem16[i];
  }
  __syncthreads();
  return r;
}

__device__ __forceinline__ void sigmoid_ce_body(const float* __restrict__ z, const float* __restrict__ y, const float* __restrict__ w, int B,
                  float loss_scale, float* __restrict__ loss_out, float* __restrict__ dz, float* __restrict__ probs) {
  __shared__ float red[kCeBlock / 64];
  __shared__ float s_nz;
  float nz = static_cast<float>(B);
  if (w) {
    float cnt = 0.f;
#pragma unroll 8
    for (int i = threadIdx.x; i < B; i += kCeBlock) cnt += (w[i] != 0.f ? 1.f : 0.f);
    const float tot = block_sum_1024(cnt, red);
    if (threadIdx.x == 0) s_nz = tot > 0.f ? tot : 1.f;
    __syncthreads();
    nz = s_nz;
  }
  float acc = 0.f;
#pragma unroll 4
  for (int i = threadIdx.x; i < B; i += kCeBlock) {
    const float zi = z[i], yi = y[i];
    const float wi = w ? w[i] : 1.f;
    // tf.nn.sigmoid_cross_entropy_with_logits: max(z,0) - z*y + log1p(exp(-|z|))
    const float ce = fmaxf(zi, 0.f) - zi * yi + log1pf(expf(-fabsf(zi)));
    acc = acc + wi * ce;
    const float p = 1.f / (1.f + expf(-zi));
    if (probs) probs[i] = p;
    if (dz) dz[i] = loss_scale * wi * (p - yi) / nz;
  }
  const float s = block_sum_1024(acc, red);
  if (threadIdx.x == 0 && loss_out) loss_out[0] = loss_scale * s / nz;
}

__global__ void __launch_bounds__(kCeBlock)
sigmoid_ce_kernel(const float* __restrict__ z, const float* __restrict__ y, const float* __restrict__ w, int B,
                  float loss_scale, float* __restrict__ loss_out, float* __restrict__ dz, float* __restrict__ probs) {
  sigmoid_ce_body(z, y, w, B, loss_scale, loss_out, dz, probs);
}

// the losses of several heads (the towers of a multi-task model) in one launch: workgroup t runs head t's body
constexpr int kCeMulti = 8;
struct CeMultiArgs {
  const float* z[kCeMulti]; const float* y[kCeMulti]; const float* w[kCeMulti];
  float* loss[kCeMulti]; float* dz[kCeMulti]; float* probs[kCeMulti];
  int B[kCeMulti];
  float scale[kCeMulti];
};
__global__ void __launch_bounds__(kCeBlock)
sigmoid_ce_multi_kernel(CeMultiArgs a) {
  const int t = blockIdx.x;
  sigmoid_ce_body(a.z[t], a.y[t], a.w[t], a.B[t], a.scale[t], a.loss[t], a.dz[t], a.probs[t]);
}

// regularization_loss = reg_emb + reg_dense; total_loss = regularization_loss + sum_i losses[i]; copies of the
// individual losses into their report slots: the estimator's add_n over the loss dict + REGULARIZATION_LOSSES
// (model/easy_rec_estimator.py:166-184) as ONE launch instead of ~6 scalar add / copy kernels.
__global__ void total_loss_kernel(const float* __restrict__ reg_emb, const float* __restrict__ reg_dense, LossPtrs lp,
                                  int n, float* __restrict__ reg_out, float* __restrict__ total_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float reg = reg_emb[0] + reg_dense[0];
  reg_out[0] = reg;
  float total = reg;
  for (int i = 0; i < n; ++i) {
    const float v = lp.src[i][0];
    if (lp.dst[i]) lp.dst[i][0] = v;
    total = total + v;
  }
  total_out[0] = total;
}

// The scalar tail of the loss in ONE launch (one workgroup of 1024): regularization_loss = emb_scale * sum(partials)
// [the embedding-output L2: per-block sums of squares left by er_emb_fwd] + sum(dense_partials) [the kernels' L2: per-
// block sums of 0.5 * coef * w^2 that the dense optimizer leaves behind for the NEXT step while it has every weight in
// registers anyway - er_dense_opt_step_l2 / er_l2_partials; a single workgroup walking 0.28 M weights itself took
// 150 us], total_loss = regularization_loss + sum of the task losses, and the copies of the task losses into their
// report slots.  Replaces er_reduce_sum + er_l2_loss (two launches) + er_total_loss.  Fixed order: deterministic.
__global__ void __launch_bounds__(kCeBlock)
reg_total_loss_kernel(const float* __restrict__ emb_partials, int n_partials, float emb_scale,
                      const float* __restrict__ dense_partials, int n_dense, LossPtrs lp, int n_losses,
                      float* __restrict__ reg_out, float* __restrict__ total_out) {
  __shared__ float red[kCeBlock / 64];
  float a = 0.f;
  for (int i = threadIdx.x; i < n_partials; i += kCeBlock) a = a + emb_partials[i];
  const float emb = block_sum_1024(a, red);
  float b = 0.f;
  for (int i = threadIdx.x; i < n_dense; i += kCeBlock) b = b + dense_partials[i];
  const float dense = block_sum_1024(b, red);
  if (threadIdx.x != 0) return;
  const float reg = emb_scale * emb + dense;
  reg_out[0] = reg;
  float total = reg;
  for (int i = 0; i < n_losses; ++i) {
    const float v = lp.src[i][0];
    if (lp.dst[i]) lp.dst[i][0] = v;
    total = total + v;
  }
  total_out[0] = total;
}

// ------------------------------------------------------------------------------------------------
// The binary head of a rank model in ONE launch (er_head_sigmoid_ce): logits = x . w + b (the `output` projection of
// model/deepfm.py:84-88, dcn.py:66, ...: tf.layers.dense(units = 1)), tf.losses.sigmoid_cross_entropy of them
// (builders/loss_builder.py:35-39, no sample weights: mean over the batch), AND the head's own backward - d loss / d
// logits, dx = dz (x) w, per-workgroup partial sums of dW = x^T dz and db = sum dz, and (when x is the output of a dense
// + BatchNorm + ReLU layer) that layer's BatchNorm-backward column sums in the layout er_gemm_f32_bn_bwd leaves them.
// What the step issued before: a 64 x 64-tile GEMM for one output column, the loss kernel (one workgroup), a column
// sum, a K = 1 dgrad GEMM, a BatchNorm-backward partial pass and a share of the grouped weight-gradient launch.
// A workgroup owns 64 rows (= one row tile of the GEMMs: the BatchNorm partials line up); a row's K values are spread
// over G = K / 4 lanes (16-byte loads), dot products finish with a fixed xor-shuffle tree: deterministic.
// ------------------------------------------------------------------------------------------------
constexpr int kHeadRows = 64;
struct HeadArgs {
  const float* x; int ldx;
  const float* w; const float* b; const float* y;
  int B, K, G;
  float scale, nz;  // loss_scale, B as a float: dz = scale * (p - y) / nz, the loss kernel's own expression
  float* logits; float* probs; float* dz; float* dx;
  float* loss_part;   // [tiles]
  float* wb_part;     // [tiles][K + 1]: sum_rows x[r][c] * dz[r] | sum_rows dz[r]
  const float* src_z; const float* src_mean; const float* src_invstd; int src_ld, src_act, src_bn;
  float* bn_part;     // [tiles][K][2] or nullptr
};

template <int PASSES>
__device__ __forceinline__ void head_body(const HeadArgs& a, float* smem) {
  const int G = a.G, K = a.K;
  const int rpp = kBlock / G;                 // rows per pass (PASSES * rpp == kHeadRows unless G < 4)
  const int sub = threadIdx.x % G, rl = threadIdx.x / G;
  const int c = sub * 4;
  const bool col_ok = c < K;
  const int r0 = blockIdx.x * kHeadRows;
  const bool bn = a.bn_part != nullptr;
  __shared__ float s_z[kHeadRows], s_dzr[kHeadRows];
  // every load of the workgroup is issued before the first use: one round trip to memory, not one per pass
  float4 xv[PASSES], zv[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int r = r0 + ps * rpp + rl;
    const bool ok = r < a.B && col_ok && ps * rpp + rl < kHeadRows;  // (G < 4: more row lanes than the tile has rows)
    xv[ps] = ok ? *reinterpret_cast<const float4*>(a.x + static_cast<int64_t>(r) * a.ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    zv[ps] = (ok && bn) ? *reinterpret_cast<const float4*>(a.src_z + static_cast<int64_t>(r) * a.src_ld + c)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int myrow = r0 + static_cast<int>(threadIdx.x);  // phase 2: thread t < 64 owns row t of the tile
  const float yrow = (threadIdx.x < kHeadRows && myrow < a.B) ? a.y[myrow] : 0.f;
  float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_ok) wv = *reinterpret_cast<const float4*>(a.w + c);
  const float bias = a.b ? a.b[0] : 0.f;
  float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bn && a.src_bn && col_ok) {
    mu = *reinterpret_cast<const float4*>(a.src_mean + c);
    is = *reinterpret_cast<const float4*>(a.src_invstd + c);
  }
  const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, iss[4] = {is.x, is.y, is.z, is.w};
  // phase 1: the rows' dot products (G lanes per row, fixed xor-shuffle tree) -> LDS
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    float dot = (xv[ps].x * wv.x + xv[ps].y * wv.y) + (xv[ps].z * wv.z + xv[ps].w * wv.w);
    // G = 4 * PASSES lanes per row (one pass: G = 1, 2 or 4)
    dot = PASSES > 1 ? group_sum<(kBlock / kHeadRows) * PASSES>(dot) : group_sum_rt(dot, G);
    if (sub == 0 && ps * rpp + rl < kHeadRows) s_z[ps * rpp + rl] = dot + bias;
  }
  __syncthreads();
  // phase 2: ONE wavefront evaluates the loss of the tile's 64 rows (exp / log1p / two divisions per row are ~350 wave
  // instructions: every lane group repeating them for its own row was 2 us per workgroup)
  if (threadIdx.x < kHeadRows) {
    const bool ok = myrow < a.B;
    const float z = s_z[threadIdx.x];
    // tf.nn.sigmoid_cross_entropy_with_logits: max(z,0) - z*y + log1p(exp(-|z|))
    float ce = fmaxf(z, 0.f) - z * yrow + log1pf(expf(-fabsf(z)));
    const float p = 1.f / (1.f + expf(-z));
    float dzr = a.scale * (p - yrow) / a.nz;
    if (!ok) { ce = 0.f; dzr = 0.f; }
    s_dzr[threadIdx.x] = dzr;
    if (ok) {
      a.logits[myrow] = z;
      if (a.probs) a.probs[myrow] = p;
      if (a.dz) a.dz[myrow] = dzr;
    }
    const float cs = wave_sum(ce), ds = wave_sum(dzr);
    if (threadIdx.x == 0) {
      a.loss_part[blockIdx.x] = cs;
      a.wb_part[static_cast<int64_t>(blockIdx.x) * (K + 1) + K] = ds;
    }
  }
  __syncthreads();
  // phase 3: dx = dz (x) w, and the per-column partial sums of dW and of the producing layer's BatchNorm backward
  float dw[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int lr = ps * rpp + rl;
    const int r = r0 + lr;
    if (!(r < a.B && lr < kHeadRows && col_ok)) continue;
    const float dzr = s_dzr[lr];
    const float g[4] = {dzr * wv.x, dzr * wv.y, dzr * wv.z, dzr * wv.w};
    *reinterpret_cast<float4*>(a.dx + static_cast<int64_t>(r) * K + c) = make_float4(g[0], g[1], g[2], g[3]);
    const float xs[4] = {xv[ps].x, xv[ps].y, xv[ps].z, xv[ps].w};
    const float zs[4] = {zv[ps].x, zv[ps].y, zv[ps].z, zv[ps].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dw[j] = dw[j] + xs[j] * dzr;
      if (bn) {
        float gj = g[j];
        if (a.src_act == ER_ACT_RELU && !(xs[j] > 0.f)) gj = 0.f;  // (x IS the layer's activation output y)
        sg[j] = sg[j] + gj;
        if (a.src_bn) sgx[j] = sgx[j] + gj * ((zs[j] - mus[j]) * iss[j]);
      }
    }
  }
  // combine the rpp row lanes of every column in a fixed order
  float* s_dw = smem;                         // [rpp][K]
  float* s_g = smem + rpp * K;                // [rpp][K]
  float* s_gx = smem + 2 * rpp * K;           // [rpp][K]
  if (col_ok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s_dw[rl * K + c + j] = dw[j];
      s_g[rl * K + c + j] = sg[j];
      s_gx[rl * K + c + j] = sgx[j];
    }
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < K) {
    float d = 0.f, g1 = 0.f, g2 = 0.f;
    for (int q = 0; q < rpp; ++q) {
      d = d + s_dw[q * K + t];
      g1 = g1 + s_g[q * K + t];
      g2 = g2 + s_gx[q * K + t];
    }
    a.wb_part[static_cast<int64_t>(blockIdx.x) * (K + 1) + t] = d;
    if (bn) {
      float* pp = a.bn_part + (static_cast<int64_t>(blockIdx.x) * K + t) * 2;
      pp[0] = g1;
      pp[1] = g2;
    }
  }
}

__global__ void __launch_bounds__(kBlock)
head_sigmoid_ce_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [3][rows_per_pass][K] + 2 * [rows_per_pass]
  switch (a.G) {  // passes = 64 rows / (256 / G) rows per pass
    case 64: head_body<16>(a, smem); break;
    case 32: head_body<8>(a, smem); break;
    case 16: head_body<4>(a, smem); break;
    case 8: head_body<2>(a, smem); break;
    default: head_body<1>(a, smem); break;  // G <= 4: one pass covers the 64 rows (the surplus row lanes idle)
  }
}

// (the loss tail's records and body: er_dense_tail.h - the fused tail of er_embedding.hip runs the same body)
__global__ void __launch_bounds__(kCeBlock)
loss_tail_kernel(LossTailArgs a) {
  __shared__ float lds[kLossTailLdsFloats];
  loss_tail_body<kCeBlock>(a, lds);
}

__global__ void __launch_bounds__(kBlock)
reduce_sum_kernel(const float* __restrict__ p, int n, float scale, float* __restrict__ out, int accumulate) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += kBlock) acc = acc + p[i];
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = accumulate ? (out[0] + scale * s) : scale * s;
}

__global__ void __launch_bounds__(kBlock)
l2_loss_partial_kernel(const float* __restrict__ w, const float* __restrict__ coef, int64_t n,
                       float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const float c = coef[i];
    if (c != 0.f) acc = acc + c * (0.5f * (w[i] * w[i]));
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// ------------------------------------------------------------------------------------------------
// dense-variable optimizer over the flat buffer
// ------------------------------------------------------------------------------------------------
// (dense_opt_elem: er_dense_tail.h)
__global__ void __launch_bounds__(kBlock)
dense_opt_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ grad,
                 const float* __restrict__ l2coef, int64_t n, int opt_kind, const er_opt_hyper* __restrict__ hyper,
                 float* __restrict__ l2_partial) {
  __shared__ float red[4];
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  float l2 = 0.f;
  if (i < n) l2 = dense_opt_elem(w, m, v, grad, l2coef, i, opt_kind, *hyper);
  if (l2_partial) {  // (uniform) sum over the block of 0.5 * coef * w_new^2: the next step's kernel-L2 loss term
    const float sum = block_sum_256(l2, red);
    if (threadIdx.x == 0) l2_partial[blockIdx.x] = sum;
  }
}

// partial[b] = sum over block b's 256 weights of 0.5 * coef * w^2, the block mapping and order of dense_opt_kernel
__global__ void __launch_bounds__(kBlock)
l2_partials_kernel(const float* __restrict__ w, const float* __restrict__ coef, int64_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  float l2 = 0.f;
  if (i < n) {
    const float c = coef[i];
    if (c != 0.f) l2 = c * (0.5f * (w[i] * w[i]));
  }
  const float sum = block_sum_256(l2, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = sum;
}



// per-step scalars: out[:] = table[counter % n_slots][:]; counter += 1   (block 0)
// + the step's other input-independent front work, side by side in the same launch (each was a ~5 us launch of its own):
//   blocks [0, zero_blocks)            zero `zero_n` floats at `zero` (the flat gradient buffer of the dense variables)
//   blocks [.., + hash_blocks)         string_to_hash_bucket_fast of the batch's id strings (er_farmhash.h)
// (the closed-form replay's per-launch table needs the step index, which block 0 is incrementing while the other blocks
// run: it rides with the id sort instead, er_embedding.hip)
struct PrologueHash {
  const uint8_t* bytes;
  const int64_t* offsets;
  int64_t n, n_per_col;
  const uint64_t* num_buckets;
  int drop_empty;
  int64_t* out;
};
__global__ void __launch_bounds__(256)
hyper_select_kernel(const float* __restrict__ table, int64_t* __restrict__ counter, int n_slots,
                                    int floats_per_slot, float* __restrict__ out, float* __restrict__ hist,
                                    int64_t hist_capacity, int hist_index, float* __restrict__ zero, int64_t zero_n,
                                    DecayTabDev tabs, int zero_blocks, PrologueHash hs, int hash_blocks) {
  const int bid = blockIdx.x;
  if (bid >= zero_blocks + hash_blocks) {  // this step's lag-1 replay table, from the counter the last lookup launch left
    decay_build_lag1(tabs, hist, tabs.lag[0], (bid - zero_blocks - hash_blocks) * static_cast<int>(blockDim.x >> 6) + static_cast<int>(threadIdx.x >> 6));
    return;
  }
  if (bid >= zero_blocks) {  // hash
    const int64_t stride = static_cast<int64_t>(hash_blocks) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(bid - zero_blocks) * blockDim.x + threadIdx.x; i < hs.n; i += stride)
      hs.out[i] = hash_bucket_one(hs.bytes, hs.offsets, i, hs.n_per_col, hs.num_buckets, hs.drop_empty);
    return;
  }
  if (zero) {
    const int64_t stride = static_cast<int64_t>(zero_blocks) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(bid) * blockDim.x + threadIdx.x; i < zero_n; i += stride) zero[i] = 0.f;
  }
  if (bid != 0) return;
  const int64_t c = *counter;
  const int64_t slot = c % n_slots;
  for (int i = threadIdx.x; i < floats_per_slot; i += blockDim.x) out[i] = table[slot * floats_per_slot + i];
  __syncthreads();
  if (threadIdx.x == 0) {
    // per-step history of one scalar (Adam's lr_t): the lazy dense-decay catch-up replays past steps from it
    // hist holds 2 * hist_capacity floats: [value per step | running maximum of the values up to that step] (the
    // absorbed regime of the replay bounds every later update with the largest lr_t so far)
    if (hist && c < hist_capacity) {
      const float val = table[slot * floats_per_slot + hist_index];
      hist[c] = val;
      const float before = c > 0 ? hist[hist_capacity + c - 1] : 0.f;
      hist[hist_capacity + c] = val > before ? val : before;
    }
    *counter = c + 1;
  }
  // closed-form replay tables (er_decay.h): the sums of the row-idle interval that starts after step t0 = c - K are
  // complete now that lr_t(c) exists: C[t0 + 1][:] = T(t0, K)
  if (tabs.coef != nullptr && hist && c < hist_capacity && c - tabs.K + 1 >= 0) {
    __syncthreads();  // hist[c] written by thread 0
    if (threadIdx.x < 64) {
      const float mine = decay_sum_wave(tabs, hist, c - tabs.K, tabs.K);
      if (threadIdx.x < kDecayLd) tabs.C[(c - tabs.K + 1) * kDecayLd + threadIdx.x] = mine;
    }
  }
}

inline int blocks_for(int64_t n) { return static_cast<int>(ceil_div(n, kBlock)); }

// scratch for column-reduction partials: one buffer per stream would be the general answer; the
// training loop is single-stream per process, so one lazily grown device buffer is used.
static float* g_scratch = nullptr;
static size_t g_scratch_floats = 0;
// Host functions that launch a producer and a consumer of the shared scratch take this lock: ranks simulated as
// threads (tests) call concurrently with the GIL released, and a launch of another thread between the two would
// overwrite the scratch in stream order.
static std::mutex g_scratch_mu;
static std::mutex g_merge_mu;  // the merged-partials buffer (taken after g_scratch_mu where both are held)

// merged BatchNorm partials of tall activations (bn_stats_merge_kernel / bn_bwd_merge_kernel): 3 floats x 64 K columns
static float* g_merged = nullptr;
constexpr size_t kMergedFloats = (3u << 12) * 64;  // kMergeSlices records x 3 floats x 4 K columns
int get_merged(float** out) {
  if (!g_merged) {
    hipError_t e = hipMalloc(&g_merged, kMergedFloats * sizeof(float));
    if (e != hipSuccess) {
      g_merged = nullptr;
      set_error("merge buffer allocation failed: %s", hipGetErrorString(e));
      return 1;
    }
  }
  *out = g_merged;
  return 0;
}

int get_scratch(size_t floats, float** out) {
  if (floats > g_scratch_floats) {
    if (g_scratch) (void)hipFree(g_scratch);
    size_t want = floats < (1u << 20) ? (1u << 20) : floats;
    hipError_t e = hipMalloc(&g_scratch, want * sizeof(float));
    if (e != hipSuccess) {
      g_scratch = nullptr;
      g_scratch_floats = 0;
      set_error("scratch allocation of %zu floats failed: %s", want, hipGetErrorString(e));
      return 1;
    }
    g_scratch_floats = want;
  }
  *out = g_scratch;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// K16 streaming AUC (tf.metrics.auc, used by RankModel.build_metric_graph, model/rank_model.py:358-373).
// TF keeps, per threshold t, tp/fn/tn/fp = counts of (label, prediction > t).  With sorted thresholds that is a
// histogram over bucket(p) = #{t : p > t}: counts[label][bucket], integer atomics (order-independent, exact).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
auc_hist_kernel(const float* __restrict__ probs, const float* __restrict__ labels, const float* __restrict__ weights,
                int64_t n, const float* __restrict__ thresholds, int T, unsigned long long* __restrict__ counts) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  if (weights && !(weights[i] > 0.f)) return;
  const float p = probs[i];
  int lo = 0, hi = T;  // first index with !(p > thresholds[idx]) == number of thresholds below p
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (p > thresholds[mid]) lo = mid + 1; else hi = mid;
  }
  const int pos = labels[i] != 0.f ? 1 : 0;  // tf.cast(labels, bool)
  atomicAdd(&counts[static_cast<size_t>(pos) * (T + 1) + lo], 1ull);
}

// ------------------------------------------------------------------------------------------------
// K16b grouped AUC (gAUC / session AUC).  The rows arrive sorted by (key, prediction): thread i finds its key's
// segment [gs, ge) and, inside it, the run [rs, re) of predictions equal to its own (binary searches), so its average
// 1-based rank in the segment is (rs - gs + 1 + re - gs) / 2 - a half-integer, exact in a double, so the atomic sums
// below do not depend on their order.  Accumulators live at the segment's first index.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
grouped_auc_rank_kernel(const int64_t* __restrict__ keys, const float* __restrict__ preds, const float* __restrict__ labels,
                        int64_t n, double* __restrict__ pos_rank_sum, double* __restrict__ n_pos, double* __restrict__ n_all) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t key = keys[i];
  int64_t lo = 0, hi = i;          // gs: first index with keys[idx] >= key
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
  const int64_t gs = lo;
  lo = i + 1; hi = n;              // ge: first index with keys[idx] > key
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] <= key) lo = mid + 1; else hi = mid; }
  const int64_t ge = lo;
  if (i == gs) n_all[gs] = static_cast<double>(ge - gs);
  if (labels[i] == 0.f) return;
  const float p = preds[i];
  lo = gs; hi = i;                 // rs: first index of the segment with preds[idx] >= p
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (preds[mid] < p) lo = mid + 1; else hi = mid; }
  const int64_t rs = lo;
  lo = i + 1; hi = ge;             // re: first index of the segment with preds[idx] > p
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (preds[mid] <= p) lo = mid + 1; else hi = mid; }
  const int64_t re = lo;
  atomicAdd(&pos_rank_sum[gs], 0.5 * static_cast<double>((rs - gs + 1) + (re - gs)));
  atomicAdd(&n_pos[gs], 1.0);
}

// per segment: the Mann-Whitney AUC when both classes are present, weighted by the reduction
// (0 'mean': 1, 1 'mean_by_sample_num': rows, 2 'mean_by_positive_num': positives; reference core/metrics.py:59-108);
// out[0] += w * auc, out[1] += w, out[2] += 1
__global__ void __launch_bounds__(kBlock)
grouped_auc_reduce_kernel(const double* __restrict__ pos_rank_sum, const double* __restrict__ n_pos,
                          const double* __restrict__ n_all, int64_t n, int reduction, double* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const double cnt = n_all[i];
  if (cnt <= 0.0) return;
  const double np = n_pos[i], nn = cnt - np;
  if (np <= 0.0 || nn <= 0.0) return;  // (keys with one class are skipped, metrics.py:93-94)
  const double auc = (pos_rank_sum[i] - np * (np + 1.0) * 0.5) / (np * nn);
  const double w = reduction == 0 ? 1.0 : (reduction == 1 ? cnt : np);
  atomicAdd(&out[0], w * auc);
  atomicAdd(&out[1], w);
  atomicAdd(&out[2], 1.0);
}

}  // namespace er

// er_loss_tail's / er_dense_opt_step_l2's argument checks and device records (also used by er_emb_bwd_fused_tail)
int er::make_loss_tail_args(const er_loss_tail_job* job, er::LossTailArgs* out) {
  ER_REQUIRE(job && out, "er_loss_tail: null job");
  const er_loss_tail_job& q = *job;
  ER_REQUIRE(q.reg_out && q.total_out && q.n_losses >= 0 && q.n_losses <= 8 && q.n_partials >= 0 && (q.emb_partials || q.n_partials == 0) &&
                 q.n_dense >= 0 && (q.dense_partials || q.n_dense == 0) && q.n_jobs >= 0 && q.n_jobs <= er::kTailJobs && (q.jobs || q.n_jobs == 0) &&
                 (q.losses || q.n_losses == 0),
             "er_loss_tail: bad arguments (at most 8 losses, %d column-sum jobs)", er::kTailJobs);
  er::LossTailArgs& a = *out;
  a.emb_partials = q.emb_partials; a.n_partials = q.n_partials; a.emb_scale = q.emb_scale;
  a.dense_partials = q.dense_partials; a.n_dense = q.n_dense; a.n_losses = q.n_losses;
  for (int i = 0; i < 8; ++i) {
    a.lp.src[i] = i < q.n_losses ? q.losses[i] : nullptr;
    a.lp.dst[i] = (i < q.n_losses && q.report) ? q.report[i] : nullptr;
    a.loss_parts[i] = (i < q.n_losses && q.loss_parts) ? q.loss_parts[i] : 0;
    a.loss_scale[i] = (i < q.n_losses && q.loss_scales) ? q.loss_scales[i] : 1.f;
    a.loss_div[i] = (i < q.n_losses && q.loss_divs) ? q.loss_divs[i] : 1.f;
    a.loss_value[i] = (i < q.n_losses && q.loss_values) ? q.loss_values[i] : nullptr;
    ER_REQUIRE(i >= q.n_losses || q.losses[i], "er_loss_tail: loss %d is null", i);
  }
  a.reg_out = q.reg_out; a.total_out = q.total_out; a.n_jobs = q.n_jobs;
  for (int j = 0; j < q.n_jobs; ++j) {
    ER_REQUIRE(q.jobs[j].partial && q.jobs[j].dst && q.jobs[j].n_parts > 0 && q.jobs[j].n_cols > 0 && q.jobs[j].ld >= q.jobs[j].n_cols,
               "er_loss_tail: job %d: bad arguments", j);
    a.jobs[j] = er::TailJob{q.jobs[j].partial, q.jobs[j].dst, q.jobs[j].n_parts, q.jobs[j].n_cols, q.jobs[j].ld};
  }
  return 0;
}

int er::make_dense_opt_args(const er_dense_opt_job* job, er::DenseOptArgs* out) {
  ER_REQUIRE(job && out, "er_dense_opt_step: null job");
  const er_dense_opt_job& q = *job;
  ER_REQUIRE(q.w && q.grad && q.hyper && q.n > 0, "er_dense_opt_step: bad arguments");
  ER_REQUIRE(q.opt_kind >= ER_OPT_SGD && q.opt_kind <= ER_OPT_ADAGRAD, "er_dense_opt_step: unknown optimizer %d", q.opt_kind);
  if (q.opt_kind == ER_OPT_ADAM || q.opt_kind == ER_OPT_LAZY_ADAM) ER_REQUIRE(q.m && q.v, "er_dense_opt_step: Adam needs m, v");
  if (q.opt_kind == ER_OPT_ADAGRAD) ER_REQUIRE(q.v, "er_dense_opt_step: Adagrad needs the accumulator in v");
  *out = er::DenseOptArgs{q.w, q.m, q.v, q.grad, q.l2coef, q.n, q.opt_kind, q.hyper, q.l2_partials};
  return 0;
}

extern "C" {

/* Pre-size the internal scratch (call once before hipGraph capture: allocation is not capturable). */
int er_reserve_scratch(int64_t floats) {
  float* p;
  return er::get_scratch(static_cast<size_t>(floats), &p);
}

int er_bn_act_fwd(const float* x, const float* bias, const float* gamma, const float* beta, int32_t B, int32_t N,
                  int use_bn, float eps, float momentum, float* moving_mean, float* moving_var, int act, float* y,
                  float* save_mean, float* save_invstd, er_stream_t stream) {
  ER_REQUIRE(x && y && B > 0 && N > 0, "er_bn_act_fwd: bad arguments");
  hipStream_t s = er::as_stream(stream);
  const int64_t n = static_cast<int64_t>(B) * N;
  std::unique_lock<std::mutex> lock(er::g_scratch_mu, std::defer_lock);
  if (use_bn == ER_BN_FROZEN) {
    ER_REQUIRE(moving_mean && moving_var && save_mean && save_invstd,
               "er_bn_act_fwd: the moving statistics and save_mean/save_invstd are required with ER_BN_FROZEN");
    hipLaunchKernelGGL(er::bn_frozen_apply_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, s, x, bias, gamma, beta,
                       moving_mean, moving_var, n, N, eps, act, y, save_mean, save_invstd);
    ER_LAUNCH_CHECK();
    return 0;
  }
  if (use_bn) {
    lock.lock();
    ER_REQUIRE(save_mean && save_invstd, "er_bn_act_fwd: save_mean/save_invstd required with use_bn");
    const int chunks = er::choose_chunks(B, N);
    float* scratch;
    if (er::get_scratch(static_cast<size_t>(chunks) * N * 3, &scratch)) return 1;
    dim3 grid(static_cast<unsigned>(er::ceil_div(N, er::kColsPerBlock)), static_cast<unsigned>(chunks));
    hipLaunchKernelGGL(er::bn_stats_partial_kernel, grid, dim3(er::kBlock), 0, s, x, bias, B, N, chunks, scratch);
    ER_LAUNCH_CHECK();
    return er_bn_apply_from_stats(x, bias, scratch, chunks, gamma, beta, B, N, eps, momentum, moving_mean, moving_var,
                                  act, y, save_mean, save_invstd, stream);
  }
  hipLaunchKernelGGL(er::bn_apply_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, s, x, bias, gamma, beta,
                     save_mean, save_invstd, n, N, use_bn, act, y);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_bn_apply_from_stats(const float* x, const float* bias, const float* col_stats, int32_t chunks,
                           const float* gamma, const float* beta, int32_t B, int32_t N, float eps, float momentum,
                           float* moving_mean, float* moving_var, int act, float* y, float* save_mean,
                           float* save_invstd, er_stream_t stream) {
  return er_bn_apply_from_stats_b16(x, bias, col_stats, chunks, gamma, beta, B, N, eps, momentum, moving_mean, moving_var, act, y,
                                    save_mean, save_invstd, nullptr, 0, stream);
}

int er_bn_apply_wide_fm(const float* x, const float* col_stats, int32_t chunks, const float* gamma, const float* beta,
                        int32_t B, int32_t N, float eps, float momentum, float* moving_mean, float* moving_var, int act,
                        float* y, float* save_mean, float* save_invstd, const float* wide, int32_t n_w, int32_t ld_w,
                        const float* fm_x, int32_t F, int32_t D, int32_t ld_x, float* out, int32_t ld_out, float* sum_out,
                        er_stream_t stream) {
  ER_REQUIRE(x && y && col_stats && save_mean && save_invstd && B > 0 && N > 0 && chunks > 0 && wide && fm_x && out &&
                 sum_out && n_w > 0 && F > 0 && D > 0 && ld_out >= 1 + D + N && ld_w >= n_w && ld_x >= F * D,
             "er_bn_apply_wide_fm: bad arguments");
  if (chunks > er::kInlineChunks) {
    er::set_error("er_bn_apply_wide_fm: %d row tiles need the merge launch: use er_bn_apply_from_stats + er_wide_fm_concat", chunks);
    return 3;
  }
  const int tpb = er::apply_tiles_per_block(B);
  const int gx = static_cast<int>(er::ceil_div(N, er::kColsPerBlock));
  const int bn_blocks = gx * static_cast<int>(er::ceil_div(B, er::kApplyRows * tpb));
  const bool vec = (D % 4 == 0) && (ld_x % 4 == 0) && ((reinterpret_cast<uintptr_t>(fm_x) & 15) == 0);
  er::WideFmArgs w{wide, n_w, ld_w, fm_x, F, D, ld_x, out, ld_out, sum_out, er::blocks_for(static_cast<int64_t>(B) * (vec ? D / 4 : D))};
  const int rs_blocks = er::blocks_for(static_cast<int64_t>(B) * 4);
  dim3 grid(static_cast<unsigned>(bn_blocks + w.fm_blocks + rs_blocks));
  if (vec) {
    hipLaunchKernelGGL(er::bn_apply_wide_fm_kernel<4>, grid, dim3(er::kBlock), 0, er::as_stream(stream), col_stats, x, nullptr,
                       gamma, beta, B, N, chunks, eps, momentum, moving_mean, moving_var, act, y, save_mean, save_invstd, tpb, gx,
                       bn_blocks, w);
  } else {
    hipLaunchKernelGGL(er::bn_apply_wide_fm_kernel<1>, grid, dim3(er::kBlock), 0, er::as_stream(stream), col_stats, x, nullptr,
                       gamma, beta, B, N, chunks, eps, momentum, moving_mean, moving_var, act, y, save_mean, save_invstd, tpb, gx,
                       bn_blocks, w);
  }
  ER_LAUNCH_CHECK();
  return 0;
}

int er_bn_apply_from_stats_b16(const float* x, const float* bias, const float* col_stats, int32_t chunks,
                               const float* gamma, const float* beta, int32_t B, int32_t N, float eps, float momentum,
                               float* moving_mean, float* moving_var, int act, float* y, float* save_mean,
                               float* save_invstd, uint16_t* y_bf16, int32_t ld_bf16, er_stream_t stream) {
  ER_REQUIRE(x && y && col_stats && save_mean && save_invstd && B > 0 && N > 0 && chunks > 0 && (!y_bf16 || ld_bf16 >= N),
             "er_bn_apply_from_stats: bad arguments");
  std::unique_lock<std::mutex> merge_lock(er::g_merge_mu, std::defer_lock);
  if (chunks > er::kInlineChunks && static_cast<size_t>(N) * 3 * er::kMergeSlices <= er::kMergedFloats) {
    float* merged;
    if (er::get_merged(&merged)) return 1;
    merge_lock.lock();  // (held until the consumer below is launched)
    const int per_slice = static_cast<int>(er::ceil_div(chunks, er::kMergeSlices));
    const int slices = static_cast<int>(er::ceil_div(chunks, per_slice));
    hipLaunchKernelGGL(er::bn_stats_merge_kernel,
                       dim3(static_cast<unsigned>(er::ceil_div(N, er::kColsPerBlock)), static_cast<unsigned>(slices)),
                       dim3(er::kBlock), 0, er::as_stream(stream), col_stats, N, chunks, per_slice, merged);
    ER_LAUNCH_CHECK();
    col_stats = merged;
    chunks = slices;
  }
  const int tpb = er::apply_tiles_per_block(B);
  dim3 grid(static_cast<unsigned>(er::ceil_div(N, er::kColsPerBlock)),
            static_cast<unsigned>(er::ceil_div(B, er::kApplyRows * tpb)));
  hipLaunchKernelGGL(er::bn_finalize_apply_kernel, grid, dim3(er::kBlock), 0, er::as_stream(stream), col_stats, x, bias,
                     gamma, beta, B, N, chunks, eps, momentum, moving_mean, moving_var, act, y, save_mean, save_invstd, tpb,
                     y_bf16, ld_bf16);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_bn_finalize_from_stats(const float* col_stats, int32_t chunks, int32_t B, int32_t N, float eps, float momentum,
                              float* moving_mean, float* moving_var, float* save_mean, float* save_invstd,
                              er_stream_t stream) {
  ER_REQUIRE(col_stats && save_mean && save_invstd && B > 0 && N > 0 && chunks > 0, "er_bn_finalize_from_stats: bad arguments");
  std::unique_lock<std::mutex> merge_lock(er::g_merge_mu, std::defer_lock);
  if (chunks > er::kInlineChunks && static_cast<size_t>(N) * 3 * er::kMergeSlices <= er::kMergedFloats) {
    float* merged;
    if (er::get_merged(&merged)) return 1;
    merge_lock.lock();  // (held until the consumer below is launched)
    const int per_slice = static_cast<int>(er::ceil_div(chunks, er::kMergeSlices));
    const int slices = static_cast<int>(er::ceil_div(chunks, per_slice));
    hipLaunchKernelGGL(er::bn_stats_merge_kernel,
                       dim3(static_cast<unsigned>(er::ceil_div(N, er::kColsPerBlock)), static_cast<unsigned>(slices)),
                       dim3(er::kBlock), 0, er::as_stream(stream), col_stats, N, chunks, per_slice, merged);
    ER_LAUNCH_CHECK();
    col_stats = merged;
    chunks = slices;
  }
  hipLaunchKernelGGL(er::bn_finalize_stats_kernel, dim3(static_cast<unsigned>(er::ceil_div(N, er::kColsPerBlock))), dim3(er::kBlock), 0,
                     er::as_stream(stream), col_stats, B, N, chunks, eps, momentum, moving_mean, moving_var, save_mean,
                     save_invstd);
  ER_LAUNCH_CHECK();
  return 0;
}


namespace {
// > kInlineChunks partial sums per column: merged into kMergeSlices records first (the lock stays held until the
// caller has launched the consumer)
int merge_bwd_partials(const float** partial, int* chunks, int N, hipStream_t s, std::unique_lock<std::mutex>* lock) {
  if (*chunks <= er::kInlineChunks || static_cast<size_t>(N) * 2 * er::kMergeSlices > er::kMergedFloats) return 0;
  float* merged;
  if (er::get_merged(&merged)) return 1;
  lock->lock();
  const int per_slice = static_cast<int>(er::ceil_div(*chunks, er::kMergeSlices));
  const int slices = static_cast<int>(er::ceil_div(*chunks, per_slice));
  hipLaunchKernelGGL(er::bn_bwd_merge_kernel,
                     dim3(static_cast<unsigned>(er::ceil_div(N, er::kColsPerBlock)), static_cast<unsigned>(slices)),
                     dim3(er::kBlock), 0, s, *partial, N, *chunks, per_slice, merged);
  ER_LAUNCH_CHECK();
  *partial = merged;
  *chunks = slices;
  return 0;
}
}  // namespace

int er_bn_act_bwd(const float* x, const float* bias, const float* gamma, const float* y, const float* save_mean,
                  const float* save_invstd, const float* dy, int32_t B, int32_t N, int use_bn, int act, float* dx,
                  float* dbias, float* dgamma, float* dbeta, int accumulate, er_stream_t stream) {
  return er_bn_act_bwd_ld(x, bias, gamma, y, save_mean, save_invstd, dy, N, B, N, use_bn, act, dx, dbias, dgamma, dbeta,
                          accumulate, stream);
}

int er_bn_act_bwd_ld(const float* x, const float* bias, const float* gamma, const float* y, const float* save_mean,
                     const float* save_invstd, const float* dy, int32_t dy_ld, int32_t B, int32_t N, int use_bn, int act,
                     float* dx, float* dbias, float* dgamma, float* dbeta, int accumulate, er_stream_t stream) {
  return er_bn_act_bwd_ld_b16(x, bias, gamma, y, save_mean, save_invstd, dy, dy_ld, B, N, use_bn, act, dx, dbias, dgamma, dbeta,
                              accumulate, nullptr, 0, stream);
}

int er_bn_act_bwd_ld_b16(const float* x, const float* bias, const float* gamma, const float* y, const float* save_mean,
                         const float* save_invstd, const float* dy, int32_t dy_ld, int32_t B, int32_t N, int use_bn, int act,
                         float* dx, float* dbias, float* dgamma, float* dbeta, int accumulate, uint16_t* dx_bf16,
                         int32_t ld_bf16, er_stream_t stream) {
  ER_REQUIRE(x && y && dy && dx && B > 0 && N > 0 && dy_ld >= N && (!dx_bf16 || ld_bf16 >= N), "er_bn_act_bwd: bad arguments");
  hipStream_t s = er::as_stream(stream);
  const int chunks = er::choose_chunks(B, N);
  std::lock_guard<std::mutex> lock(er::g_scratch_mu);
  float* scratch;
  if (er::get_scratch(static_cast<size_t>(chunks) * N * 2, &scratch)) return 1;
  dim3 grid(static_cast<unsigned>(er::ceil_div(N, er::kColsPerBlock)), static_cast<unsigned>(chunks));
  hipLaunchKernelGGL(er::bn_bwd_partial_kernel, grid, dim3(er::kBlock), 0, s, x, bias, y, save_mean, save_invstd, dy, B,
                     N, chunks, use_bn, act, scratch, dy_ld);
  ER_LAUNCH_CHECK();
  const float* partial = scratch;
  int n_partial = chunks;
  std::unique_lock<std::mutex> merge_lock(er::g_merge_mu, std::defer_lock);
  if (int rc = merge_bwd_partials(&partial, &n_partial, N, s, &merge_lock)) return rc;
  const int tpb = er::apply_tiles_per_block(B);
  dim3 grid2(static_cast<unsigned>(er::ceil_div(N, er::kColsPerBlock)),
             static_cast<unsigned>(er::ceil_div(B, er::kApplyRows * tpb)));
  hipLaunchKernelGGL(er::bn_bwd_finalize_apply_kernel, grid2, dim3(er::kBlock), 0, s, partial, x, bias, gamma, y,
                     save_mean, save_invstd, dy, B, N, n_partial, use_bn, act, accumulate, dx, dbias, dgamma, dbeta, dy_ld, tpb,
                     dx_bf16, ld_bf16);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_bn_act_bwd_from_partials(const float* x, const float* bias, const float* gamma, const float* y,
                                const float* save_mean, const float* save_invstd, const float* dy, int32_t B, int32_t N,
                                int use_bn, int act, const float* partial, int32_t chunks, float* dx, float* dbias,
                                float* dgamma, float* dbeta, int accumulate, er_stream_t stream) {
  return er_bn_act_bwd_from_partials_ld(x, bias, gamma, y, save_mean, save_invstd, dy, N, B, N, use_bn, act, partial, chunks, dx,
                                        dbias, dgamma, dbeta, accumulate, stream);
}

int er_bn_act_bwd_from_partials_ld(const float* x, const float* bias, const float* gamma, const float* y,
                                   const float* save_mean, const float* save_invstd, const float* dy, int32_t dy_ld, int32_t B,
                                   int32_t N, int use_bn, int act, const float* partial, int32_t chunks, float* dx, float* dbias,
                                   float* dgamma, float* dbeta, int accumulate, er_stream_t stream) {
  return er_bn_act_bwd_from_partials_ld_b16(x, bias, gamma, y, save_mean, save_invstd, dy, dy_ld, B, N, use_bn, act, partial, chunks,
                                            dx, dbias, dgamma, dbeta, accumulate, nullptr, 0, stream);
}

int er_bn_act_bwd_from_partials_ld_b16(const float* x, const float* bias, const float* gamma, const float* y,
                                       const float* save_mean, const float* save_invstd, const float* dy, int32_t dy_ld, int32_t B,
                                       int32_t N, int use_bn, int act, const float* partial, int32_t chunks, float* dx,
                                       float* dbias, float* dgamma, float* dbeta, int accumulate, uint16_t* dx_bf16,
                                       int32_t ld_bf16, er_stream_t stream) {
  ER_REQUIRE(x && y && dy && dx && partial && B > 0 && N > 0 && chunks > 0 && dy_ld >= N && (!dx_bf16 || ld_bf16 >= N),
             "er_bn_act_bwd_from_partials: bad arguments");
  std::unique_lock<std::mutex> merge_lock(er::g_merge_mu, std::defer_lock);
  if (int rc = merge_bwd_partials(&partial, &chunks, N, er::as_stream(stream), &merge_lock)) return rc;
  const int tpb = er::apply_tiles_per_block(B);
  dim3 grid2(static_cast<unsigned>(er::ceil_div(N, er::kColsPerBlock)),
             static_cast<unsigned>(er::ceil_div(B, er::kApplyRows * tpb)));
  hipLaunchKernelGGL(er::bn_bwd_finalize_apply_kernel, grid2, dim3(er::kBlock), 0, er::as_stream(stream), partial, x, bias,
                     gamma, y, save_mean, save_invstd, dy, B, N, chunks, use_bn, act, accumulate, dx, dbias, dgamma, dbeta, dy_ld, tpb,
                     dx_bf16, ld_bf16);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_bn_fwd_multi(const er_bn_layer* layers, int n, er_stream_t stream) {
  ER_REQUIRE(layers && n >= 1, "er_bn_fwd_multi: bad arguments");
  for (int base = 0; base < n; base += er::kBnMulti) {
    er::BnMultiArgs a;
    a.n = 0;
    a.start[0] = 0;
    for (int i = base; i < n && i < base + er::kBnMulti; ++i) {
      const er_bn_layer& q = layers[i];
      ER_REQUIRE(q.x && q.y && q.B > 0 && q.N > 0, "er_bn_fwd_multi: layer %d: bad arguments", i);
      er::BnItem& d = a.d[a.n];
      d = er::BnItem();
      d.x = q.x; d.bias = q.bias; d.gamma = q.gamma; d.beta = q.beta; d.moving_mean = q.moving_mean; d.moving_var = q.moving_var;
      d.y = q.y; d.save_mean = q.save_mean; d.save_invstd = q.save_invstd;
      d.B = q.B; d.N = q.N; d.mode = q.use_bn; d.act = q.act; d.eps = q.eps; d.momentum = q.momentum;
      int blocks;
      if (q.use_bn == 1) {
        ER_REQUIRE(q.col_stats && q.chunks > 0 && q.chunks <= er::kInlineChunks && q.save_mean && q.save_invstd,
                   "er_bn_fwd_multi: layer %d: batch statistics need col_stats of at most %d row tiles, save_mean, save_invstd",
                   i, er::kInlineChunks);
        d.partial = q.col_stats; d.chunks = q.chunks;
        d.tpb = er::apply_tiles_per_block(q.B);
        d.gx = static_cast<int>(er::ceil_div(q.N, er::kColsPerBlock));
        blocks = d.gx * static_cast<int>(er::ceil_div(q.B, er::kApplyRows * d.tpb));
      } else {
        ER_REQUIRE(q.use_bn != ER_BN_FROZEN || (q.moving_mean && q.moving_var && q.save_mean && q.save_invstd),
                   "er_bn_fwd_multi: layer %d: ER_BN_FROZEN needs the moving statistics and save_mean / save_invstd", i);
        d.gx = 1;
        blocks = er::blocks_for(static_cast<int64_t>(q.B) * q.N);
      }
      a.start[a.n + 1] = a.start[a.n] + blocks;
      ++a.n;
    }
    hipLaunchKernelGGL(er::bn_fwd_multi_kernel, dim3(static_cast<unsigned>(a.start[a.n])), dim3(er::kBlock), 0,
                       er::as_stream(stream), a);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

static bool frozen_one_pass() {  // (A/B knob: '0' = column sums, then finalize + apply - two passes over dy / y / z)
  static const bool on = [] {
    const char* e = getenv("ER_BN_FROZEN_ONE_PASS");
    return !(e && e[0] == '0');
  }();
  return on;
}

int er_bn_bwd_multi(const er_bn_layer* layers, int n, er_stream_t stream) {
  ER_REQUIRE(layers && n >= 1, "er_bn_bwd_multi: bad arguments");
  hipStream_t s = er::as_stream(stream);
  std::lock_guard<std::mutex> lock(er::g_scratch_mu);
  for (int base = 0; base < n; base += er::kBnMulti) {
    er::BnMultiArgs p1, p2;
    p1.n = p2.n = 0;
    p1.start[0] = p2.start[0] = 0;
    size_t need = 0;
    const int m = (n - base < er::kBnMulti) ? n - base : er::kBnMulti;
    for (int i = base; i < base + m; ++i) {  // scratch for the layers whose column sums are computed here
      const er_bn_layer& q = layers[i];
      if (!q.partial) need += static_cast<size_t>(er::choose_chunks(q.B, q.N)) * q.N * 2;
    }
    float* scratch = nullptr;
    if (need && er::get_scratch(need, &scratch)) return 1;
    size_t off = 0;
    for (int i = base; i < base + m; ++i) {
      const er_bn_layer& q = layers[i];
      // (dx == NULL with ready-made partial sums: the parameter gradients only - dx was written by the contraction that
      // produced dy, er_gemm_problem.bn_dz_out)
      ER_REQUIRE(q.x && q.dy && (q.dx || (q.partial && q.use_bn == ER_BN_FROZEN)) && q.B > 0 && q.N > 0 && q.dy_ld >= q.N,
                 "er_bn_bwd_multi: layer %d: bad arguments", i);
      ER_REQUIRE(!q.use_bn || (q.save_mean && q.save_invstd), "er_bn_bwd_multi: layer %d: BatchNorm statistics missing", i);
      er::BnItem d = er::BnItem();
      d.x = q.x; d.bias = q.bias; d.gamma = q.gamma; d.beta = q.beta; d.yin = q.y_in; d.save_mean = q.save_mean;
      d.save_invstd = q.save_invstd; d.dy = q.dy; d.dy_ld = q.dy_ld; d.dx = q.dx; d.dbias = q.dbias; d.dgamma = q.dgamma;
      d.dbeta = q.dbeta; d.accumulate = q.accumulate; d.B = q.B; d.N = q.N; d.mode = q.use_bn; d.act = q.act;
      d.gx = static_cast<int>(er::ceil_div(q.N, er::kColsPerBlock));
      d.tpb = er::apply_tiles_per_block(q.B);
      if (q.partial) {
        ER_REQUIRE(q.chunks > 0 && q.chunks <= er::kInlineChunks, "er_bn_bwd_multi: layer %d: at most %d partials", i,
                   er::kInlineChunks);
        d.partial = q.partial; d.chunks = q.chunks;
        d.dx_in_partial = q.dx ? 0 : 1;
      } else {
        d.chunks = er::choose_chunks(q.B, q.N);
        ER_REQUIRE(d.chunks <= er::kInlineChunks, "er_bn_bwd_multi: layer %d: too tall for the grouped form", i);
        d.scratch = scratch + off;
        d.partial = d.scratch;
        off += static_cast<size_t>(d.chunks) * q.N * 2;
        d.dx_in_partial = (q.use_bn == ER_BN_FROZEN && frozen_one_pass()) ? 1 : 0;
        p1.d[p1.n] = d;
        p1.start[p1.n + 1] = p1.start[p1.n] + d.gx * d.chunks;
        ++p1.n;
      }
      p2.d[p2.n] = d;
      p2.start[p2.n + 1] = p2.start[p2.n] + d.gx * (d.dx_in_partial ? 1 : static_cast<int>(er::ceil_div(q.B, er::kApplyRows * d.tpb)));
      ++p2.n;
    }
    if (p1.n > 0) {
      hipLaunchKernelGGL(er::bn_bwd_partial_multi_kernel, dim3(static_cast<unsigned>(p1.start[p1.n])), dim3(er::kBlock), 0, s, p1);
      ER_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(er::bn_bwd_finalize_apply_multi_kernel, dim3(static_cast<unsigned>(p2.start[p2.n])), dim3(er::kBlock),
                       0, s, p2);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

int er_colsum(const float* x, int32_t rows, int32_t cols, int32_t x_stride, float* out, er_stream_t stream) {
  return er_colsum_acc(x, rows, cols, x_stride, out, 0, stream);
}

int er_colsum_acc(const float* x, int32_t rows, int32_t cols, int32_t x_stride, float* out, int accumulate,
                  er_stream_t stream) {
  ER_REQUIRE(x && out && rows > 0 && cols > 0, "er_colsum: bad arguments");
  hipStream_t s = er::as_stream(stream);
  if (cols <= 8 && static_cast<int64_t>(rows) * cols <= (1 << 17)) {
    hipLaunchKernelGGL(er::colsum_narrow_kernel, dim3(static_cast<unsigned>(cols)), dim3(er::kBlock), 0, s, x, rows, cols,
                       x_stride, out, accumulate);
    ER_LAUNCH_CHECK();
    return 0;
  }
  const int chunks = er::choose_chunks(rows, cols);
  float* scratch;
  if (er::get_scratch(static_cast<size_t>(chunks) * cols, &scratch)) return 1;
  dim3 grid(static_cast<unsigned>(er::ceil_div(cols, er::kColsPerBlock)), static_cast<unsigned>(chunks));
  hipLaunchKernelGGL(er::colsum_partial_kernel, grid, dim3(er::kBlock), 0, s, x, rows, cols, x_stride, chunks, scratch);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::colsum_finalize_kernel, dim3(er::blocks_for(static_cast<int64_t>(cols) * 64)), dim3(er::kBlock), 0, s, scratch, cols,
                     chunks, out, accumulate);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_colsum_narrow_multi(const er_colsum_job* jobs, int32_t n_jobs, int accumulate, er_stream_t stream) {
  ER_REQUIRE(jobs && n_jobs >= 1, "er_colsum_narrow_multi: bad arguments");
  for (int base = 0; base < n_jobs; base += er::kMaxNarrowJobs) {
    er::NarrowJobs a;
    a.n = 0;
    a.start[0] = 0;
    for (int i = base; i < n_jobs && i < base + er::kMaxNarrowJobs; ++i) {
      const er_colsum_job& q = jobs[i];
      ER_REQUIRE(q.x && q.out && q.rows > 0 && q.cols > 0 && q.cols <= 64 && q.x_stride >= q.cols,
                 "er_colsum_narrow_multi: job %d: bad descriptor", i);
      a.j[a.n] = q;
      a.start[a.n + 1] = a.start[a.n] + q.cols;
      ++a.n;
    }
    hipLaunchKernelGGL(er::colsum_narrow_multi_kernel, dim3(static_cast<unsigned>(a.start[a.n])), dim3(er::kBlock), 0,
                       er::as_stream(stream), a, accumulate);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

int er_colsum_partials_multi(const er_tail_job* jobs, int32_t n_jobs, int accumulate, er_stream_t stream) {
  ER_REQUIRE(jobs && n_jobs >= 1, "er_colsum_partials_multi: bad arguments");
  for (int base = 0; base < n_jobs; base += er::kMaxColsumJobs) {
    er::ColsumJobs a;
    a.n = 0;
    a.start[0] = 0;
    for (int i = base; i < n_jobs && i < base + er::kMaxColsumJobs; ++i) {
      const er_tail_job& q = jobs[i];
      ER_REQUIRE(q.partial && q.dst && q.n_parts > 0 && q.n_cols > 0 && q.ld >= q.n_cols, "er_colsum_partials_multi: job %d: bad descriptor", i);
      a.j[a.n] = q;
      a.start[a.n + 1] = a.start[a.n] + static_cast<int>(er::ceil_div(q.n_cols, er::kBlock / 64));
      ++a.n;
    }
    hipLaunchKernelGGL(er::colsum_partials_multi_kernel, dim3(static_cast<unsigned>(a.start[a.n])), dim3(er::kBlock), 0,
                       er::as_stream(stream), a, accumulate);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

int er_dice_fwd(const float* x, const float* alpha, int32_t B, int32_t N, float eps, float momentum,
                float* moving_mean, float* moving_var, float* y, float* save_mean, float* save_invstd,
                er_stream_t stream) {
  ER_REQUIRE(x && alpha && y && save_mean && save_invstd && B > 0 && N > 0, "er_dice_fwd: bad arguments");
  hipStream_t s = er::as_stream(stream);
  const int64_t n = static_cast<int64_t>(B) * N;
  const int chunks = er::choose_chunks(B, N);
  float* scratch;
  if (er::get_scratch(static_cast<size_t>(chunks) * N * 3, &scratch)) return 1;
  dim3 grid(static_cast<unsigned>(er::ceil_div(N, er::kColsPerBlock)), static_cast<unsigned>(chunks));
  hipLaunchKernelGGL(er::bn_stats_partial_kernel, grid, dim3(er::kBlock), 0, s, x, static_cast<const float*>(nullptr), B,
                     N, chunks, scratch);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::bn_finalize_kernel, dim3(er::blocks_for(static_cast<int64_t>(N) * 64)), dim3(er::kBlock), 0, s, scratch, B, N, chunks, eps,
                     momentum, moving_mean, moving_var, save_mean, save_invstd);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::dice_apply_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, s, x, alpha, save_mean,
                     save_invstd, n, N, y);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_dice_bwd(const float* x, const float* alpha, const float* save_mean, const float* save_invstd, const float* dy,
                int32_t B, int32_t N, float* dx, float* dalpha, er_stream_t stream) {
  ER_REQUIRE(x && alpha && save_mean && save_invstd && dy && dx && B > 0 && N > 0, "er_dice_bwd: bad arguments");
  hipStream_t s = er::as_stream(stream);
  const int64_t n = static_cast<int64_t>(B) * N;
  const int chunks = er::choose_chunks(B, N);
  float* scratch;
  if (er::get_scratch(static_cast<size_t>(chunks) * N * 3 + 2 * static_cast<size_t>(N), &scratch)) return 1;
  float* sums = scratch + static_cast<size_t>(chunks) * N * 3;
  dim3 grid(static_cast<unsigned>(er::ceil_div(N, er::kColsPerBlock)), static_cast<unsigned>(chunks));
  hipLaunchKernelGGL(er::dice_bwd_partial_kernel, grid, dim3(er::kBlock), 0, s, x, alpha, save_mean, save_invstd, dy, B,
                     N, chunks, scratch);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::dice_bwd_finalize_kernel, dim3(er::blocks_for(static_cast<int64_t>(N) * 64)), dim3(er::kBlock), 0, s, scratch, N, chunks,
                     sums, dalpha);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::dice_bwd_apply_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, s, x, alpha, save_mean,
                     save_invstd, dy, sums, n, B, N, dx);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_sigmoid_ce_fwd_bwd(const float* logits, const float* labels, const float* weights, int32_t B,
                          float loss_scale, float* loss_out, float* dlogits, float* probs_out, er_stream_t stream) {
  ER_REQUIRE(logits && labels && B > 0, "er_sigmoid_ce_fwd_bwd: bad arguments");
  hipLaunchKernelGGL(er::sigmoid_ce_kernel, dim3(1), dim3(er::kCeBlock), 0, er::as_stream(stream), logits, labels,
                     weights, B, loss_scale, loss_out, dlogits, probs_out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_sigmoid_ce_multi(const er_ce_head* heads, int n, er_stream_t stream) {
  ER_REQUIRE(heads && n >= 1, "er_sigmoid_ce_multi: bad arguments");
  for (int base = 0; base < n; base += er::kCeMulti) {
    er::CeMultiArgs a;
    const int m = n - base < er::kCeMulti ? n - base : er::kCeMulti;
    for (int i = 0; i < m; ++i) {
      const er_ce_head& h = heads[base + i];
      ER_REQUIRE(h.logits && h.labels && h.B > 0, "er_sigmoid_ce_multi: head %d: bad arguments", base + i);
      a.z[i] = h.logits; a.y[i] = h.labels; a.w[i] = h.weights; a.loss[i] = h.loss_out; a.dz[i] = h.dlogits;
      a.probs[i] = h.probs_out; a.B[i] = h.B; a.scale[i] = h.loss_scale;
    }
    hipLaunchKernelGGL(er::sigmoid_ce_multi_kernel, dim3(static_cast<unsigned>(m)), dim3(er::kCeBlock), 0,
                       er::as_stream(stream), a);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

int er_total_loss(const float* reg_emb, const float* reg_dense, const float* const* losses_host, float* const* report_host,
                  int32_t n, float* reg_out, float* total_out, er_stream_t stream) {
  ER_REQUIRE(reg_emb && reg_dense && reg_out && total_out && n >= 0 && n <= 8, "er_total_loss: bad arguments (n <= 8)");
  er::LossPtrs lp;
  for (int i = 0; i < 8; ++i) {
    lp.src[i] = (i < n) ? losses_host[i] : nullptr;
    lp.dst[i] = (i < n && report_host) ? report_host[i] : nullptr;
  }
  hipLaunchKernelGGL(er::total_loss_kernel, dim3(1), dim3(64), 0, er::as_stream(stream), reg_emb, reg_dense, lp, n,
                     reg_out, total_out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_reduce_sum(const float* partials, int32_t n, float scale, float* out, int accumulate, er_stream_t stream) {
  ER_REQUIRE(partials && out && n > 0, "er_reduce_sum: bad arguments");
  hipLaunchKernelGGL(er::reduce_sum_kernel, dim3(1), dim3(er::kBlock), 0, er::as_stream(stream), partials, n, scale,
                     out, accumulate);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_l2_loss(const float* w, const float* coef, int64_t n, float* out, int accumulate, er_stream_t stream) {
  ER_REQUIRE(w && coef && out && n > 0, "er_l2_loss: bad arguments");
  hipStream_t s = er::as_stream(stream);
  int64_t blocks = er::ceil_div(n, static_cast<int64_t>(er::kBlock) * 4);
  if (blocks > 1024) blocks = 1024;
  float* scratch;
  if (er::get_scratch(static_cast<size_t>(blocks), &scratch)) return 1;
  hipLaunchKernelGGL(er::l2_loss_partial_kernel, dim3(static_cast<int>(blocks)), dim3(er::kBlock), 0, s, w, coef, n,
                     scratch);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::reduce_sum_kernel, dim3(1), dim3(er::kBlock), 0, s, scratch, static_cast<int>(blocks), 1.0f,
                     out, accumulate);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_hyper_select(const float* table, int64_t* counter, int32_t n_slots, int32_t floats_per_slot, float* out,
                    float* history, int64_t history_capacity, int32_t history_index, er_stream_t stream) {
  return er_step_prologue(table, counter, n_slots, floats_per_slot, out, history, history_capacity, history_index, nullptr,
                          0, stream);
}

int er_step_prologue(const float* table, int64_t* counter, int32_t n_slots, int32_t floats_per_slot, float* out,
                     float* history, int64_t history_capacity, int32_t history_index, float* zero, int64_t zero_floats,
                     er_stream_t stream) {
  return er_step_prologue_decay(table, counter, n_slots, floats_per_slot, out, history, history_capacity, history_index,
                                zero, zero_floats, nullptr, stream);
}

int er_step_prologue_decay(const float* table, int64_t* counter, int32_t n_slots, int32_t floats_per_slot, float* out,
                           float* history, int64_t history_capacity, int32_t history_index, float* zero,
                           int64_t zero_floats, er_decay_tables* decay_tables, er_stream_t stream) {
  return er_step_prologue_hash(table, counter, n_slots, floats_per_slot, out, history, history_capacity, history_index, zero,
                               zero_floats, decay_tables, nullptr, nullptr, 0, 1, nullptr, 0, nullptr, stream);
}

int er_step_prologue_hash(const float* table, int64_t* counter, int32_t n_slots, int32_t floats_per_slot, float* out,
                          float* history, int64_t history_capacity, int32_t history_index, float* zero, int64_t zero_floats,
                          er_decay_tables* decay_tables, const uint8_t* str_bytes, const int64_t* str_offsets, int64_t n_strings,
                          int64_t n_per_col, const uint64_t* num_buckets, int drop_empty, int64_t* ids_out, er_stream_t stream) {
  ER_REQUIRE(table && counter && out && n_slots > 0 && floats_per_slot > 0, "er_step_prologue: bad arguments");
  ER_REQUIRE(!decay_tables || (decay_tables->hist == history && decay_tables->counter == counter &&
                               decay_tables->dev.capacity == history_capacity),
             "er_step_prologue: the decay tables were created for another history / counter");
  er::DecayTabDev tabs{};
  if (decay_tables) tabs = decay_tables->dev;
  ER_REQUIRE(!history || (history_index >= 0 && history_index < floats_per_slot && history_capacity > 0),
             "er_step_prologue: bad history arguments");
  ER_REQUIRE(zero_floats >= 0 && (zero || zero_floats == 0), "er_step_prologue: bad zero arguments");
  ER_REQUIRE(n_strings >= 0 && (n_strings == 0 || (str_bytes && str_offsets && num_buckets && ids_out && n_per_col > 0)),
             "er_step_prologue_hash: bad hash arguments");
  int64_t blocks = zero ? er::ceil_div(zero_floats, 256 * 8) : 1;
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  int64_t hb = er::ceil_div(n_strings, 256);
  if (hb > 4096) hb = 4096;
  er::PrologueHash hs{str_bytes, str_offsets, n_strings, n_per_col > 0 ? n_per_col : 1, num_buckets, drop_empty, ids_out};
  // (one wavefront per k of the lag-1 table: er_decay_tables_set_prologue_build)
  const int64_t tb = (decay_tables && decay_tables->prologue_build && history) ? er::ceil_div(static_cast<int64_t>(tabs.K) * 64, 256) : 0;
  hipLaunchKernelGGL(er::hyper_select_kernel, dim3(static_cast<unsigned>(blocks + hb + tb)), dim3(256), 0, er::as_stream(stream), table,
                     counter, n_slots, floats_per_slot, out, history, history_capacity, history_index, zero, zero_floats, tabs,
                     static_cast<int>(blocks), hs, static_cast<int>(hb));
  ER_LAUNCH_CHECK();
  return 0;
}

int er_reg_total_loss(const float* emb_partials, int32_t n_partials, float emb_scale, const float* dense_partials,
                      int32_t n_dense, const float* const* losses, float* const* report, int32_t n_losses, float* reg_out,
                      float* total_out, er_stream_t stream) {
  ER_REQUIRE(reg_out && total_out && n_losses >= 0 && n_losses <= 8 && n_partials >= 0 && (emb_partials || n_partials == 0) &&
                 n_dense >= 0 && (dense_partials || n_dense == 0),
             "er_reg_total_loss: bad arguments (at most 8 losses)");
  er::LossPtrs lp;
  for (int i = 0; i < 8; ++i) {
    lp.src[i] = i < n_losses ? losses[i] : nullptr;
    lp.dst[i] = (i < n_losses && report) ? report[i] : nullptr;
  }
  hipLaunchKernelGGL(er::reg_total_loss_kernel, dim3(1), dim3(er::kCeBlock), 0, er::as_stream(stream), emb_partials,
                     n_partials, emb_scale, dense_partials, n_dense, lp, n_losses, reg_out, total_out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_head_sigmoid_ce(const float* x, int32_t ldx, const float* w, const float* b, const float* labels, int32_t B, int32_t K,
                       float loss_scale, float* logits, float* probs, float* dlogits, float* dx, float* loss_partials,
                       float* wb_partials, const float* src_z, int32_t src_ld, const float* src_mean, const float* src_invstd,
                       int32_t src_act, float* bn_partials, er_stream_t stream) {
  ER_REQUIRE(x && w && labels && logits && dx && loss_partials && wb_partials && B > 0 && K > 0,
             "er_head_sigmoid_ce: bad arguments");
  ER_REQUIRE(K % 4 == 0 && K <= 256 && ldx >= K && ldx % 4 == 0, "er_head_sigmoid_ce: K must be a multiple of 4, at most 256 (K = %d, ldx = %d)", K, ldx);
  ER_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
             "er_head_sigmoid_ce: x, w and dx must be 16-byte aligned");
  ER_REQUIRE(!bn_partials || (src_z && src_ld >= K && src_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(src_z) & 15) == 0 &&
                              (!src_mean == !src_invstd)),
             "er_head_sigmoid_ce: the BatchNorm source needs z (16-byte aligned rows) and mean / invstd together");
  er::HeadArgs a;
  int G = 1;
  while (G < K / 4) G <<= 1;
  a.x = x; a.ldx = ldx; a.w = w; a.b = b; a.y = labels; a.B = B; a.K = K; a.G = G;
  a.scale = loss_scale;
  a.nz = static_cast<float>(B);
  a.logits = logits; a.probs = probs; a.dz = dlogits; a.dx = dx; a.loss_part = loss_partials; a.wb_part = wb_partials;
  a.src_z = src_z; a.src_mean = src_mean; a.src_invstd = src_invstd; a.src_ld = src_ld; a.src_act = src_act;
  a.src_bn = src_mean != nullptr; a.bn_part = bn_partials;
  const int tiles = static_cast<int>(er::ceil_div(B, er::kHeadRows));
  const int rpp = er::kBlock / G;
  const size_t smem = (static_cast<size_t>(3) * rpp * K + 2 * rpp) * sizeof(float);
  hipLaunchKernelGGL(er::head_sigmoid_ce_kernel, dim3(tiles), dim3(er::kBlock), smem, er::as_stream(stream), a);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_loss_tail(const float* emb_partials, int32_t n_partials, float emb_scale, const float* dense_partials, int32_t n_dense,
                 const float* const* losses, float* const* report, const int32_t* loss_parts, const float* loss_scales,
                 const float* loss_divs, float* const* loss_values, int32_t n_losses, const er_tail_job* jobs, int32_t n_jobs, float* reg_out,
                 float* total_out, er_stream_t stream) {
  er_loss_tail_job job;
  job.emb_partials = emb_partials; job.n_partials = n_partials; job.emb_scale = emb_scale;
  job.dense_partials = dense_partials; job.n_dense = n_dense;
  job.losses = losses; job.report = report; job.loss_parts = loss_parts; job.loss_scales = loss_scales; job.loss_divs = loss_divs;
  job.loss_values = loss_values; job.n_losses = n_losses; job.jobs = jobs; job.n_jobs = n_jobs;
  job.reg_out = reg_out; job.total_out = total_out;
  er::LossTailArgs a;
  if (int rc = er::make_loss_tail_args(&job, &a)) return rc;
  hipLaunchKernelGGL(er::loss_tail_kernel, dim3(1), dim3(er::kCeBlock), 0, er::as_stream(stream), a);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_dense_opt_step(float* w, float* m, float* v, const float* grad, const float* l2coef, int64_t n, int opt_kind,
                      const er_opt_hyper* hyper, er_stream_t stream) {
  return er_dense_opt_step_l2(w, m, v, grad, l2coef, n, opt_kind, hyper, nullptr, stream);
}

int er_l2_partials(const float* w, const float* coef, int64_t n, float* partials, er_stream_t stream) {
  ER_REQUIRE(w && coef && partials && n > 0, "er_l2_partials: bad arguments");
  hipLaunchKernelGGL(er::l2_partials_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, er::as_stream(stream), w, coef, n,
                     partials);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_dense_opt_step_l2(float* w, float* m, float* v, const float* grad, const float* l2coef, int64_t n, int opt_kind,
                         const er_opt_hyper* hyper, float* l2_partials, er_stream_t stream) {
  er_dense_opt_job job{w, m, v, const_cast<float*>(grad), l2coef, n, opt_kind, hyper, l2_partials};
  er::DenseOptArgs chk;
  if (int rc = er::make_dense_opt_args(&job, &chk)) return rc;
  hipLaunchKernelGGL(er::dense_opt_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, er::as_stream(stream), w, m, v,
                     grad, l2coef, n, opt_kind, hyper, l2_partials);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_auc_update(const float* probs, const float* labels, const float* weights, int64_t n, const float* thresholds,
                  int32_t num_thresholds, uint64_t* counts, er_stream_t stream) {
  ER_REQUIRE(probs && labels && thresholds && counts && n >= 0 && num_thresholds >= 2, "er_auc_update: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(er::auc_hist_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, er::as_stream(stream), probs, labels,
                     weights, n, thresholds, num_thresholds, reinterpret_cast<unsigned long long*>(counts));
  ER_LAUNCH_CHECK();
  return 0;
}

int er_grouped_auc(const int64_t* keys, const float* preds, const float* labels, int64_t n, int reduction, double* work,
                   double* out, er_stream_t stream) {
  ER_REQUIRE(n >= 0 && out && (n == 0 || (keys && preds && labels && work)) && reduction >= 0 && reduction <= 2,
             "er_grouped_auc: bad arguments");
  if (n == 0) return 0;
  hipStream_t s = er::as_stream(stream);
  hipLaunchKernelGGL(er::grouped_auc_rank_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, s, keys, preds, labels, n, work,
                     work + n, work + 2 * n);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::grouped_auc_reduce_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, s, work, work + n, work + 2 * n,
                     n, reduction, out);
  ER_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
