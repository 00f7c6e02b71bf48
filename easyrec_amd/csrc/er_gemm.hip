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
#include "er_common.h"

namespace er {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64;
constexpr int BK32 = 32;   // k-tile of the f32 kernel
constexpr int BK16 = 32;   // k-tile of the bf16 kernel (two 32x32x16 steps)

// Optional epilogue of the dgrad GEMM dy = dz_next . W_next^T: the per-row-tile column sums the BatchNorm
// backward of the layer that produced y needs (sum g, sum g * xhat with g = dy masked by the activation), i.e.
// the output of bn_bwd_partial_kernel without another pass over dy / y / z.
struct BnBwdEpi {
  const float* z = nullptr;       // pre-normalisation values of the producing layer [M][ld] (its GEMM output)
  const float* zbias = nullptr;   // bias added to z (nullptr: already included)
  const float* y = nullptr;       // its activation output [M][ld]
  const float* mean = nullptr;
  const float* invstd = nullptr;
  int ld = 0, use_bn = 0, act = 0;
  float* partial = nullptr;       // [row tiles][N][2]
  // y == nullptr (the producing layer's activation output was never materialised: ATransform below): the ReLU mask is
  // recomputed from z with the producing layer's affine parameters
  const float* gamma = nullptr;
  const float* beta = nullptr;
};

// y of a dense + BatchNorm + activation layer from its pre-normalisation value: the operation sequence of
// bn_finalize_apply_kernel (er_dense.hip), so a recomputed y has the bits of a materialised one
__device__ __forceinline__ float bn_act_value(float z, float mu, float is, float ga, float be, int use_bn, int act) {
  float v = z;
  if (use_bn) {
    v = (z - mu) * is;
    v = v * ga + be;
  }
  if (act == ER_ACT_RELU) v = v > 0.f ? v : 0.f;
  return v;
}

// Operand A as the output of a dense + BatchNorm(train) + activation layer that was NEVER WRITTEN: A points at that
// layer's pre-normalisation values z (bias included) and the staging applies bn_act_value with the layer's batch
// statistics and affine parameters per FEATURE - A is [batch, features] row-major both as the forward operand (NN:
// features = k) and as the weight-gradient operand (TN: features = m), so a staging unit's 4 contiguous elements are 4
// consecutive features either way.  Saves the separate normalise + activate pass and the write + re-reads of y.
struct ATransform {
  const float* mean = nullptr;  // nullptr: A is used as it is
  const float* invstd = nullptr;
  const float* gamma = nullptr;  // nullptr: 1
  const float* beta = nullptr;   // nullptr: 0
  int act = 0;
};
constexpr int kTrMaxK = 1024;  // forward operand: the whole feature axis (= K) sits in the LDS parameter table

// BatchNorm fused INTO the epilogue (forward: normalise + activation; backward: the dz of the producing layer) needs
// the column statistics of ALL row tiles before any output can be written: the workgroups of one column of tiles meet
// at a barrier (arrive counter + spin; the whole grid is co-resident: er_gemm_fused_bn_ok), each then finalises the
// statistics of its 64 columns redundantly - in exactly the order bn_finalize_apply_kernel /
// bn_bwd_finalize_apply_kernel use, so the results are bit-identical to the two-launch form - and transforms its
// accumulator tile in registers.  One launch per dense + BatchNorm + ReLU layer instead of two, and no second pass over
// the layer's output.
struct BnFused {
  int mode = 0;                    // 0: off; 1: forward (needs col_stats); 2: backward (needs bn.partial);
                                   // 3: forward statistics only, WITHOUT a barrier: the workgroup that is last to add
                                   //    its partial to a column of tiles (arrival counter) finalises mean / invstd /
                                   //    moving statistics of those 64 columns; nobody waits, z is written as usual
  const float* gamma = nullptr;
  const float* beta = nullptr;     // forward
  float eps = 0.f, momentum = 0.f;
  float* moving_mean = nullptr;    // forward, may be nullptr (build pass)
  float* moving_var = nullptr;
  float* save_mean = nullptr;      // forward: [N] outputs
  float* save_invstd = nullptr;
  float* y = nullptr;              // forward: activation output [M][ldy]
  int ldy = 0, act = 0;
  float* dgamma = nullptr;         // backward: parameter gradients (accumulated when accumulate != 0), may be nullptr
  float* dbeta = nullptr;
  float* dbias = nullptr;          // backward without BatchNorm: the bias gradient
  int accumulate = 0;
  unsigned* counters = nullptr;    // [column tiles][2], all zero between launches (the barrier resets itself)
};

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;           // output, or split-K workspace [splits][M][N] when splits > 1
  const float* bias;  // [N] or nullptr (applied by the last stage only)
  int M, N, K;
  int lda, ldb, ldc;
  int accumulate;
  int k_per_split;    // multiple of the k-tile
  int splits;
  float* col_stats;   // nullptr, or [gridDim.y][N][3] Welford (count, mean, M2) of the output columns per row tile
  BnBwdEpi bn;        // bn.partial != nullptr: emit the BatchNorm-backward column sums of the output tile
  BnFused fu;         // fu.mode != 0: finish the BatchNorm in this launch (see BnFused)
  ATransform at;      // at.mean != nullptr: A is transformed while staged (kernels instantiated with A_TR)
};

// Loads 4 consecutive elements along the CONTIGUOUS dimension of the operand tile.
//   K_CONTIG : elements (mn, k..k+3);  else: elements (mn..mn+3, k)
template <bool K_CONTIG>
__device__ __forceinline__ f32x4v load4(const float* __restrict__ P, int ld, int mn, int k, int MN, int K,
                                        bool vec_ok) {
  f32x4v v = {0.f, 0.f, 0.f, 0.f};
  if (K_CONTIG) {
    if (mn >= MN) return v;
    const float* p = P + static_cast<int64_t>(mn) * ld + k;
    if (vec_ok && k + 3 < K) {
      v = *reinterpret_cast<const f32x4v*>(p);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k + j < K) v[j] = p[j];
    }
  } else {
    if (k >= K) return v;
    const float* p = P + static_cast<int64_t>(k) * ld + mn;
    if (vec_ok && mn + 3 < MN) {
      v = *reinterpret_cast<const f32x4v*>(p);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (mn + j < MN) v[j] = p[j];
    }
  }
  return v;
}

__device__ __forceinline__ short f32_to_bf16_rne(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return static_cast<short>((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return static_cast<short>(u >> 16);
}

// XCD-aware tile order.  Workgroup b of a launch is dispatched to XCD b % 8 and every XCD has its own L2
// (MI355X_MICROARCH.md); the column tiles of one row of tiles all read the same 64 rows of A.  Tiles are therefore
// numbered so that consecutive tiles (same tile row, x fastest) go to the SAME XCD: of the first 8 * per tiles
// (per = tiles / 8) slot i takes tile (i % 8) * per + i / 8 - XCD c owns tiles [c * per, (c + 1) * per) - and the
// remaining tiles % 8 keep their slot's number.
//
// The grid holds EXACTLY tiles x splits workgroups (per problem, in a grouped launch).  Rounds 1 - 2 rounded the tile
// count up to a multiple of 8 and let the surplus workgroups return at once: within an XCD workgroups are handed to the
// CUs round robin, so the live workgroups of a few-tile split-K problem landed on every 8th / 2nd CU only - the
// 128 x 128 weight gradient of DIN's attention MLP (1 - 4 tiles x 100 splits) ran three workgroups deep on 4 CUs per
// XCD while 28 idled (tools/micro/tn_stream.hip reproduces the kernel at 2.9x the speed with the same decomposition
// and no idle workgroups; profiles/r03_wgrad_probe.md).
__device__ __forceinline__ void tile_coords(int i, int gx, int gy, int& tx, int& ty) {
  const int nt = gx * gy;
  const int per = nt / 8;
  const int t = i < 8 * per ? (i % 8) * per + i / 8 : i;
  tx = t % gx;
  ty = t / gx;
}

// Values that cross workgroups INSIDE a launch (the per-row-tile partials of the fused epilogues) are stored and
// loaded as relaxed agent-scope atomics: on gfx950 these go through to the memory side (sc1) instead of sitting in
// the issuing XCD's L2, so no cache-wide writeback / invalidate (a release / acquire FENCE at agent scope costs about
// as much as the kernel boundary the fusion is meant to save - measured: 27 us per fused launch with fences).
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Per-row-tile column statistics of the OUTPUT (value = acc + bias), for a following BatchNorm: removes the
// separate statistics pass over the GEMM output.  A lane holds 16 rows of one column; lanes l and l^32 hold the
// other 16 rows; the two waves with wm = 0 / 1 cover the tile's 64 rows.  Welford/Chan merges in a fixed order.
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
  if (nb == 0.f) return;
  if (n == 0.f) { n = nb; mean = mb; m2 = m2b; return; }
  const float tot = n + nb;
  const float delta = mb - mean;
  mean = mean + delta * (nb / tot);
  m2 = m2 + m2b + delta * delta * (n * nb / tot);
  n = tot;
}

__device__ __forceinline__ void tile_col_stats(const f32x16& acc, float bv, int row_base, int M, int col, int N, int wm,
                                               int wn, int lane, float* lds /* >= 2*32*3 floats */,
                                               float* __restrict__ out_tile /* [N][3] of this row tile */,
                                               bool coherent = false /* read by other workgroups of THIS launch */) {
  const int khalf = lane >> 5;
  float n = 0.f, s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    if (row < M) { n += 1.f; s += acc[r] + bv; }
  }
  float mean = n > 0.f ? s / n : 0.f;
  float m2 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    if (row < M) { const float d = (acc[r] + bv) - mean; m2 += d * d; }
  }
  // the other 16 rows of this column live in lane ^ 32: lower half first so both lanes compute the same bits
  const float on = __shfl_xor(n, 32, 64), om = __shfl_xor(mean, 32, 64), o2 = __shfl_xor(m2, 32, 64);
  float an = khalf ? on : n, am = khalf ? om : mean, a2 = khalf ? o2 : m2;
  chan_merge(an, am, a2, khalf ? n : on, khalf ? mean : om, khalf ? m2 : o2);
  __syncthreads();  // LDS operand tiles are dead
  float* slot = lds + (wn * 32 + (lane & 31)) * 3;
  if (wm == 1 && khalf == 0) { slot[0] = an; slot[1] = am; slot[2] = a2; }
  __syncthreads();
  if (wm == 0 && khalf == 0 && col < N) {
    chan_merge(an, am, a2, slot[0], slot[1], slot[2]);
    float* o = out_tile + static_cast<int64_t>(col) * 3;
    if (coherent) { st_agent(o, an); st_agent(o + 1, am); st_agent(o + 2, a2); }
    else { o[0] = an; o[1] = am; o[2] = a2; }
  }
}

// BatchNorm-backward column sums of a 64-row output tile (see BnBwdEpi).  Fixed order: a lane's 16 rows in
// register order, then the lane pair (l, l ^ 32), then the two waves that share the columns.
__device__ __forceinline__ void tile_bn_bwd_partial(const f32x16& acc, const BnBwdEpi& e, const float (&py)[16],
                                                    const float (&pz)[16], int row_base, int M, int col, int N, int wm,
                                                    int wn, int lane, float* lds, int ty, bool coherent = false) {
  const int khalf = lane >> 5;
  float sg = 0.f, sgx = 0.f;
  if (col < N) {
    const float bv = e.zbias ? e.zbias[col] : 0.f;
    const float mu = e.use_bn ? e.mean[col] : 0.f;
    const float is = e.use_bn ? e.invstd[col] : 0.f;
    const float ga = e.gamma ? e.gamma[col] : 1.f, be = e.beta ? e.beta[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (row < M) {
        float g = acc[r];
        if (e.act == ER_ACT_RELU) {
          const float yv = e.y ? py[r] : bn_act_value(pz[r] + bv, mu, is, ga, be, e.use_bn, ER_ACT_NONE);
          if (!(yv > 0.f)) g = 0.f;
        }
        sg = sg + g;
        if (e.use_bn) sgx = sgx + g * ((pz[r] + bv - mu) * is);
      }
    }
  }
  const float og = __shfl_xor(sg, 32, 64), ogx = __shfl_xor(sgx, 32, 64);
  const float a = (khalf ? og : sg) + (khalf ? sg : og);
  const float ax = (khalf ? ogx : sgx) + (khalf ? sgx : ogx);
  __syncthreads();  // LDS operand tiles are dead
  float* slot = lds + (wn * 32 + (lane & 31)) * 2;
  if (wm == 1 && khalf == 0) { slot[0] = a; slot[1] = ax; }
  __syncthreads();
  if (wm == 0 && khalf == 0 && col < N) {
    float* p = e.partial + (static_cast<int64_t>(ty) * N + col) * 2;
    if (coherent) { st_agent(p, a + slot[0]); st_agent(p + 1, ax + slot[1]); }
    else { p[0] = a + slot[0]; p[1] = ax + slot[1]; }
  }
}

// Barrier among the `n` workgroups that share a column of tiles.  c[0]: arrivals, c[1]: departures; the last workgroup
// to leave zeroes both (every other one has left the spin by then), so the words are zero again for the next launch.
// No fences: the data the barrier orders is written with st_agent and read with ld_agent only; a writer waits for
// its stores to be acknowledged (s_waitcnt vmcnt(0): stores count in vmcnt on gfx9) before the workgroup's arrival
// is counted, a reader issues its loads after the spin has seen every arrival (in-order issue + the barrier).
__device__ __forceinline__ void tile_column_barrier(unsigned* c, unsigned n) {
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's partial stores have completed
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (bounded: ~50 ms.  The host only launches a co-resident grid; should that ever not hold, the launch produces wrong
    // statistics - which the tests catch - instead of hanging the device)
    for (unsigned spins = 0; __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n && spins < (1u << 21); ++spins)
      __builtin_amdgcn_s_sleep(1);
    const unsigned gone = __hip_atomic_fetch_add(c + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gone == n - 1) {  // everyone has left the spin: zero the words for the next launch
      __hip_atomic_store(c + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
}

// Forward: finalise the statistics of the tile's 64 columns from the per-row-tile Welford partials (thread (cl, rl)
// merges row tiles rl, rl + 4, ... in ascending order, eight loads in flight; then ((0 + 1) + (2 + 3)): the order of
// bn_finalize_apply_kernel), leave mean / invstd in LDS; the ty == 0 workgroup records them and moves the moving
// statistics.  lds: >= 256 + 768 + 128 floats.
__device__ __forceinline__ void fused_bn_fwd_finalize(const GemmArgs& g, int n0, int ty, int gy, float* lds,
                                                      float*& s_mu, float*& s_is) {
  float* sm = lds + 256;  // [4][64][3]
  s_mu = lds + 256 + 768;
  s_is = s_mu + 64;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = n0 + cl;
  float tn = 0.f, tm = 0.f, t2 = 0.f;
  if (c < g.N) {
    for (int k0 = rl; k0 < gy; k0 += 32) {
      float w[8][3];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j * 4;
        if (k < gy) {
          const float* p = g.col_stats + (static_cast<int64_t>(k) * g.N + c) * 3;
          w[j][0] = ld_agent(p); w[j][1] = ld_agent(p + 1); w[j][2] = ld_agent(p + 2);
        } else {
          w[j][0] = 0.f; w[j][1] = 0.f; w[j][2] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) chan_merge(tn, tm, t2, w[j][0], w[j][1], w[j][2]);
    }
  }
  float* mine = sm + (rl * 64 + cl) * 3;
  mine[0] = tn; mine[1] = tm; mine[2] = t2;
  __syncthreads();
  if (rl == 0 && c < g.N) {
    const float* q0 = sm + (0 * 64 + cl) * 3;
    const float* q1 = sm + (1 * 64 + cl) * 3;
    const float* q2 = sm + (2 * 64 + cl) * 3;
    const float* q3 = sm + (3 * 64 + cl) * 3;
    float an = q0[0], am = q0[1], a2 = q0[2];
    chan_merge(an, am, a2, q1[0], q1[1], q1[2]);
    float bn = q2[0], bm = q2[1], b2 = q2[2];
    chan_merge(bn, bm, b2, q3[0], q3[1], q3[2]);
    chan_merge(an, am, a2, bn, bm, b2);
    const float mean = am;
    const float var = a2 / static_cast<float>(g.M);  // biased, as tf.nn.moments
    const float inv = 1.f / sqrtf(var + g.fu.eps);
    s_mu[cl] = mean;
    s_is[cl] = inv;
    if (ty == 0) {
      g.fu.save_mean[c] = mean;
      g.fu.save_invstd[c] = inv;
      if (g.fu.moving_mean) {
        const float om = 1.f - g.fu.momentum;
        g.fu.moving_mean[c] = g.fu.moving_mean[c] - (g.fu.moving_mean[c] - mean) * om;
        g.fu.moving_var[c] = g.fu.moving_var[c] - (g.fu.moving_var[c] - var) * om;
      }
    }
  }
  __syncthreads();
}

// Backward: the two column sums (sum g, sum g * xhat) of the tile's 64 columns from the per-row-tile partials, in the
// order of bn_bwd_finalize_apply_kernel; the ty == 0 workgroup writes / accumulates the parameter gradients.
__device__ __forceinline__ void fused_bn_bwd_finalize(const GemmArgs& g, int n0, int ty, int gy, float* lds,
                                                      float*& s_g, float*& s_gx) {
  float* sm = lds + 256;  // [2][4][64]
  s_g = lds + 256 + 512;
  s_gx = s_g + 64;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = n0 + cl;
  float a = 0.f, b = 0.f;
  if (c < g.N) {
#pragma unroll 8
    for (int k = rl; k < gy; k += 4) {
      const float* p = g.bn.partial + (static_cast<int64_t>(k) * g.N + c) * 2;
      a = a + ld_agent(p);
      b = b + ld_agent(p + 1);
    }
  }
  sm[rl * 64 + cl] = a;
  sm[256 + rl * 64 + cl] = b;
  __syncthreads();
  if (rl == 0 && c < g.N) {
    a = (sm[cl] + sm[64 + cl]) + (sm[128 + cl] + sm[192 + cl]);
    b = (sm[256 + cl] + sm[256 + 64 + cl]) + (sm[256 + 128 + cl] + sm[256 + 192 + cl]);
    s_g[cl] = a;
    s_gx[cl] = b;
    if (ty == 0) {
      const int acc = g.fu.accumulate;
      if (g.bn.use_bn) {
        if (g.fu.dbeta) g.fu.dbeta[c] = acc ? g.fu.dbeta[c] + a : a;
        if (g.fu.dgamma) g.fu.dgamma[c] = acc ? g.fu.dgamma[c] + b : b;
      } else if (g.fu.dbias) {
        g.fu.dbias[c] = acc ? g.fu.dbias[c] + a : a;
      }
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// fp32: C tile 64x64 per workgroup (4 waves, one 32x32 accumulator each), k-tile 32.
//
// LDS: both operand tiles as [mn][k] with a row stride of 36 floats, two stages.  A lane's MFMA fragment is then
// 16 CONSECUTIVE k of one row = 4 ds_read_b128 per operand and k-tile (conflict-free for the b128 lane groups with
// stride 36, and b128 reads reach the LDS rate from one wave per SIMD where the ds_read_b32 of a k-major layout get
// a fifth of it - MI355X_MICROARCH.md, LDS): all 8 reads of a tile are issued up front and the 16 MFMAs run back to
// back.  Lane (i = lane & 31, h = lane >> 5) holds k = 16 h + s for MFMA step s, for A and B alike: each step
// contracts the pair (s, 16 + s) - a fixed order, the same for every launch.
//
// Staging: a thread owns 2 units (4 consecutive k of one row) per operand and k-tile.  An operand whose k runs
// contiguously in HBM fetches a unit with one 16-byte load; the other kind (B of NN, both of TN) with 4 dword loads
// that are each coalesced along mn across the wave - the transpose happens in registers, not as scattered LDS
// stores.  Either way a unit is ONE ds_write_b128 (8 consecutive lanes cover 32 distinct banks).
//
// Pipeline: two register sets hold the global loads of k-tiles t+1 and t+2 while tile t is contracted; tile t+1
// is written to the other LDS stage during the MFMAs of tile t: one barrier per k-tile, global latency has two
// k-tiles to hide in.  Interior tiles take branch-free loads; edge tiles (uniform test) masked scalar loads.
// ------------------------------------------------------------------------------------------------
constexpr int SK = BK32 + 4;      // floats per LDS row
constexpr int kOpTile = BM * SK;  // floats per operand tile (BM == BN)

// A thread's unit i (0 / 1) of an operand tile: 4 elements that are consecutive in HBM.
//   K_CONTIG : row = u >> 3, k = (u & 7) * 4 + 0..3, u = tid + 256 i            -> one ds_write_b128 at [row][k]
//   else     : rows (tid >> 4) * 4 + 0..3, k = (tid & 15) + 16 i                -> four ds_write_b32 at [row + j][k]
//              (32 consecutive lanes cover 16 k x 2 row groups = 32 distinct banks with the row stride 36)
template <bool KC>
__device__ __forceinline__ void unit_pos(int tid, int i, int& row, int& k) {
  if (KC) {
    const int u = tid + i * kBlock;
    row = u >> 3;
    k = (u & 7) * 4;
  } else {
    row = (tid >> 4) * 4;
    k = (tid & 15) + 16 * i;
  }
}

// Branch-free 16-byte loads of a thread's 2 units: indices outside the operand are clamped to a valid address and
// the values zeroed later, in stage_tile (NOT here: a use of the loaded value would wait for the load).  Needs
// ld % 4 == 0 and a 16-byte aligned base; the extent along the contiguous dimension rounded up to 4 is <= ld.
template <bool KC>
__device__ __forceinline__ void fetch_tile(const float* __restrict__ P, int ld, int mn0, int MN, int k0, int kend,
                                           int K, int tid, f32x4v (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, k;
    unit_pos<KC>(tid, i, row, k);
    int mn = mn0 + row;
    k += k0;
    if (KC) {
      const int kpad = (K + 3) & ~3;
      mn = mn < MN ? mn : MN - 1;
      k = k < kpad - 4 ? k : kpad - 4;
      r[i] = *reinterpret_cast<const f32x4v*>(P + static_cast<int64_t>(mn) * ld + k);
    } else {
      const int mnpad = (MN + 3) & ~3;
      mn = mn < mnpad - 4 ? mn : mnpad - 4;
      k = k < kend ? k : kend - 1;
      r[i] = *reinterpret_cast<const f32x4v*>(P + static_cast<int64_t>(k) * ld + mn);
    }
  }
}

// Generic loads (any alignment): masked scalar loads, used by the non-pipelined loop only.
template <bool KC>
__device__ __forceinline__ void fetch_tile_generic(const float* __restrict__ P, int ld, int mn0, int MN, int k0,
                                                   int kend, int tid, f32x4v (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, k;
    unit_pos<KC>(tid, i, row, k);
    const int mn = mn0 + row;
    k += k0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = 0.f;
      if (KC) {
        if (mn < MN && k + j < kend) v = P[static_cast<int64_t>(mn) * ld + k + j];
      } else {
        if (mn + j < MN && k < kend) v = P[static_cast<int64_t>(k) * ld + mn + j];
      }
      r[i][j] = v;
    }
  }
}

// tr (LDS, A_TR only): [4][kTrMaxK] = mean | invstd | gamma | beta, indexed by the absolute k (KC) or by the feature's
// offset inside the tile (non-KC); tr_off = what to add to the unit's own coordinate to get that index
template <bool KC>
__device__ __forceinline__ void stage_tile(float* __restrict__ S, int tid, const f32x4v (&r)[2], bool interior, int mn0,
                                           int MN, int k0, int kend, const float* __restrict__ tr = nullptr,
                                           int tr_off = 0, int tr_act = 0) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, k;
    unit_pos<KC>(tid, i, row, k);
    f32x4v v = r[i];
    if (tr != nullptr) {
      const int f = (KC ? k : row) + tr_off;  // (a multiple of 4)
      const f32x4v mu = *reinterpret_cast<const f32x4v*>(tr + f);
      const f32x4v is = *reinterpret_cast<const f32x4v*>(tr + kTrMaxK + f);
      const f32x4v ga = *reinterpret_cast<const f32x4v*>(tr + 2 * kTrMaxK + f);
      const f32x4v be = *reinterpret_cast<const f32x4v*>(tr + 3 * kTrMaxK + f);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = bn_act_value(v[j], mu[j], is[j], ga[j], be[j], 1, tr_act);
    }
    if (!interior) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = KC ? (mn0 + row < MN && k0 + k + j < kend) : (mn0 + row + j < MN && k0 + k < kend);
        if (!ok) v[j] = 0.f;
      }
    }
    if (KC) {
      *reinterpret_cast<f32x4v*>(&S[row * SK + k]) = v;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) S[(row + j) * SK + k] = v[j];
    }
  }
}

// bx: index of the workgroup among the problem's (8-rounded) tiles, bz: its k-split.  BN_EPI: with the BnBwdEpi
// epilogue - the y / z values of the lane's 16 output positions are requested BEFORE the k loop so that their
// latency hides behind it (32 more VGPRs: a separate instantiation).
template <bool A_KC, bool B_KC, bool BN_EPI = false, bool A_TR = false>
__device__ __forceinline__ void gemm_f32_block(const GemmArgs& g, int bx, int bz, float* __restrict__ lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tx, ty;
  tile_coords(bx, static_cast<int>(ceil_div(g.N, BN)), static_cast<int>(ceil_div(g.M, BM)), tx, ty);
  const int m0 = ty * BM, n0 = tx * BN;
  const int kbeg = bz * g.k_per_split;
  int kend = kbeg + g.k_per_split;
  if (kend > g.K) kend = g.K;
  const int T = (kend - kbeg + BK32 - 1) / BK32;
  float py[16], pz[16];
  if (BN_EPI && g.bn.partial != nullptr) {  // (a grouped launch may mix problems with and without the epilogue)
    int c = n0 + wn * 32 + (lane & 31);
    c = c < g.N ? c : g.N - 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      row = row < g.M ? row : g.M - 1;
      const int64_t i = static_cast<int64_t>(row) * g.bn.ld + c;
      py[r] = g.bn.y ? g.bn.y[i] : 0.f;
      pz[r] = g.bn.z[i];
    }
  }
  // A_TR: the parameter table of the A transform, behind the operand stages
  const float* tr = nullptr;
  if (A_TR && g.at.mean != nullptr) {
    float* t = lds + 2 * 2 * kOpTile;
    const int n_feat = A_KC ? ((g.K + 3) & ~3) : BM;
    const int f0 = A_KC ? 0 : m0;
    const int f_end = A_KC ? g.K : g.M;
    for (int i = tid; i < n_feat; i += kBlock) {
      const int f = f0 + i;
      const bool ok = f < f_end;
      t[i] = ok ? g.at.mean[f] : 0.f;
      t[kTrMaxK + i] = ok ? g.at.invstd[f] : 0.f;
      t[2 * kTrMaxK + i] = (ok && g.at.gamma) ? g.at.gamma[f] : 1.f;
      t[3 * kTrMaxK + i] = (ok && g.at.beta) ? g.at.beta[f] : 0.f;
    }
    tr = t;  // (visible after the __syncthreads() that precedes the first use of a staged tile ... and the first stage
             // itself reads it: synchronise here)
    __syncthreads();
  }
  const int tr_act = g.at.act;
  const bool a_vec = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
  const bool b_vec = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
  const bool rows_full = (m0 + BM <= g.M) && (n0 + BN <= g.N);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int khalf = lane >> 5;
  const int fa = (wm * 32 + (lane & 31)) * SK + khalf * 16;
  const int fb = kOpTile + (wn * 32 + (lane & 31)) * SK + khalf * 16;
  auto contract = [&](const float* base, f32x4v (&a)[4], f32x4v (&b)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i], b[q][i], acc, 0, 0, 0);
  };
  auto read_frags = [&](const float* base, f32x4v (&a)[4], f32x4v (&b)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = *reinterpret_cast<const f32x4v*>(base + fa + 4 * q);
      b[q] = *reinterpret_cast<const f32x4v*>(base + fb + 4 * q);
    }
  };

  if (a_vec && b_vec) {
    f32x4v ra0[2], rb0[2], ra1[2], rb1[2];
    // k-tile indices past the end are clamped to the last tile: the loop body has the same loads every
    // iteration (the compiler can then wait for exactly the older register set), the duplicate tile is never used
    auto fetch = [&](f32x4v (&ra)[2], f32x4v (&rb)[2], int t) {
      const int k0 = kbeg + (t < T ? t : T - 1) * BK32;
      fetch_tile<A_KC>(g.A, g.lda, m0, g.M, k0, kend, g.K, tid, ra);
      fetch_tile<B_KC>(g.B, g.ldb, n0, g.N, k0, kend, g.K, tid, rb);
    };
    auto stage = [&](int buf, const f32x4v (&ra)[2], const f32x4v (&rb)[2], int t) {
      const int k0 = kbeg + t * BK32;  // unclamped: a tile past the end is masked to zero
      const bool interior = rows_full && (k0 + BK32 <= kend);
      stage_tile<A_KC>(lds + buf * 2 * kOpTile, tid, ra, interior, m0, g.M, k0, kend, tr, A_KC ? k0 : 0, tr_act);
      stage_tile<B_KC>(lds + buf * 2 * kOpTile + kOpTile, tid, rb, interior, n0, g.N, k0, kend);
    };
    // One step = one k-tile: read its fragments from LDS stage `buf`, issue the global loads of k-tile t + 2 into
    // the register set that was written to LDS at the end of the previous step, contract, and - after three
    // quarters of the MFMAs - write the OTHER register set (k-tile t + 1, loaded during the previous step) to the
    // other LDS stage.  A global load has almost two k-tiles of matrix work to arrive.  Two steps per loop
    // iteration so that each register set keeps its registers; an odd k-tile count is rounded up with an
    // all-zero tile (masked in stage_tile).
    // Fragments are read a quarter of the k-tile at a time, right before their four MFMAs, and the compiler places the
    // instructions (no scheduling fences): against "every fragment first, fences around the MFMA groups" the bare core
    // (tools/micro/gemm_core.hip, variants 9 -> 1) gains 10 % on 8192 x 1152 x 256, 7 % on 8192 x 256 x 1152.
    auto step = [&](int buf, f32x4v (&fa_)[2], f32x4v (&fb_)[2], f32x4v (&sa)[2], f32x4v (&sb)[2], int t) {
      const float* base = lds + buf * 2 * kOpTile;
      fetch(fa_, fb_, t + 2);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const f32x4v a = *reinterpret_cast<const f32x4v*>(base + fa + 4 * q);
        const f32x4v b = *reinterpret_cast<const f32x4v*>(base + fb + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
      }
      stage(buf ^ 1, sa, sb, t + 1);
      {
        const f32x4v a = *reinterpret_cast<const f32x4v*>(base + fa + 12);
        const f32x4v b = *reinterpret_cast<const f32x4v*>(base + fb + 12);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
      }
      __syncthreads();
    };
    fetch(ra0, rb0, 0);
    fetch(ra1, rb1, 1);
    stage(0, ra0, rb0, 0);
    __syncthreads();
    for (int t = 0; t < T; t += 2) {
      step(0, ra0, rb0, ra1, rb1, t);
      step(1, ra1, rb1, ra0, rb0, t + 1);
    }
  } else {
    // unaligned operands (e.g. lda = 81): masked scalar loads, one k-tile at a time
    f32x4v ra[2], rb[2];
    for (int t = 0; t < T; ++t) {
      const int k0 = kbeg + t * BK32;
      fetch_tile_generic<A_KC>(g.A, g.lda, m0, g.M, k0, kend, tid, ra);
      fetch_tile_generic<B_KC>(g.B, g.ldb, n0, g.N, k0, kend, tid, rb);
      stage_tile<A_KC>(lds, tid, ra, tr == nullptr, m0, g.M, k0, kend, tr, A_KC ? k0 : 0, tr_act);  // (a transformed
      // out-of-range element is not zero: masked again after the transform)
      stage_tile<B_KC>(lds + kOpTile, tid, rb, true, n0, g.N, k0, kend);
      __syncthreads();
      f32x4v a[4], b[4];
      read_frags(lds, a, b);
      contract(lds, a, b);
      __syncthreads();
    }
  }
  // epilogue.  C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int col = n0 + wn * 32 + (lane & 31);
  const float bv = (g.bias && g.splits == 1 && col < g.N) ? g.bias[col] : 0.f;
  if (g.col_stats)  // (host guarantees splits == 1) every thread takes part: it synchronises the workgroup
    tile_col_stats(acc, bv, m0 + wm * 32, g.M, col, g.N, wm, wn, lane, lds,
                   g.col_stats + static_cast<int64_t>(ty) * g.N * 3, g.fu.mode == 1 || g.fu.mode == 3);
  if (g.fu.mode == 3) {
    // No barrier: a workgroup adds itself to its column's arrival counter once its partial is at the memory side
    // (st_agent + s_waitcnt, as in tile_column_barrier); the one that completes the count - whichever it is - merges
    // all partials of the 64 columns in bn_finalize_apply_kernel's fixed order and writes mean / invstd / the moving
    // statistics.  The consumer of those is the NEXT launch.
    const int gy = static_cast<int>(ceil_div(g.M, BM));
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(lds);
    if (tid == 0) {
      unsigned* c = g.fu.counters + 2 * tx;
      const unsigned before = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = before + 1u == static_cast<unsigned>(gy);
      if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next launch
      *flag = last ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *flag != 0u;
    __syncthreads();
    if (last) {
      float *s_mu, *s_is;
      fused_bn_fwd_finalize(g, n0, 0, gy, lds, s_mu, s_is);
    }
  }
  if (BN_EPI && g.bn.partial != nullptr)
    tile_bn_bwd_partial(acc, g.bn, py, pz, m0 + wm * 32, g.M, col, g.N, wm, wn, lane, lds, ty, g.fu.mode == 2);
  if (g.fu.mode == 1 || g.fu.mode == 2) {  // (uniform over the grid; host guarantees splits == 1 and a co-resident grid)
    const int gy = static_cast<int>(ceil_div(g.M, BM));
    tile_column_barrier(g.fu.counters + 2 * tx, static_cast<unsigned>(gy));
    const int cl = wn * 32 + (lane & 31);
    if (g.fu.mode == 1) {
      float *s_mu, *s_is;
      fused_bn_fwd_finalize(g, n0, ty, gy, lds, s_mu, s_is);
      if (col >= g.N) return;
      const float mu = s_mu[cl], is = s_is[cl];
      const float ga = g.fu.gamma ? g.fu.gamma[col] : 1.f, be = g.fu.beta ? g.fu.beta[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (row < g.M) {
          const float z = acc[r] + bv;
          g.C[static_cast<int64_t>(row) * g.ldc + col] = z;
          float v = (z - mu) * is;
          v = v * ga + be;
          if (g.fu.act == ER_ACT_RELU) v = v > 0.f ? v : 0.f;
          g.fu.y[static_cast<int64_t>(row) * g.fu.ldy + col] = v;
        }
      }
      return;
    }
    if (BN_EPI) {  // mode 2: dz of the producing layer instead of dy
      float *s_g, *s_gx;
      fused_bn_bwd_finalize(g, n0, ty, gy, lds, s_g, s_gx);
      if (col >= g.N) return;
      const float sg = s_g[cl], sgx = s_gx[cl];
      const float zb = g.bn.zbias ? g.bn.zbias[col] : 0.f;
      const float mu = g.bn.use_bn ? g.bn.mean[col] : 0.f, is = g.bn.use_bn ? g.bn.invstd[col] : 0.f;
      const float ga = g.fu.gamma ? g.fu.gamma[col] : 1.f;
      const float invB = 1.f / static_cast<float>(g.M);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (row < g.M) {
          float gr = acc[r];
          if (g.bn.act == ER_ACT_RELU && !(py[r] > 0.f)) gr = 0.f;
          if (g.bn.use_bn) {
            const float xh = (pz[r] + zb - mu) * is;
            gr = ga * is * (gr - sg * invB - xh * (sgx * invB));
          }
          g.C[static_cast<int64_t>(row) * g.ldc + col] = gr;
        }
      }
      return;
    }
  }
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

// the same with the A transform: the LDS parameter table behind the operand stages
constexpr int kTrLds = 2 * 2 * kOpTile + 4 * kTrMaxK;
template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(kBlock)
gemm_f32_tr_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[kTrLds];
  gemm_f32_block<A_KC, B_KC, false, true>(g, blockIdx.x, blockIdx.z, lds);
}

// Grouped launch: up to kMaxGroup independent problems of one layout in ONE grid (the weight gradients of all
// layers of a step: each is a small M x N with K = batch, far too few tiles to fill 256 CUs on its own).
// Workgroups [start[p], start[p+1]) belong to problem p: tile slot = local % tiles, k-split = local / tiles
constexpr int kMaxGroup = 16;
struct GroupedArgs {
  int n;
  int start[kMaxGroup + 1];
  int tiles[kMaxGroup];
  GemmArgs p[kMaxGroup];
};

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(kBlock)
gemm_f32_grouped_kernel(GroupedArgs ga) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kOpTile];
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < ga.n && b >= ga.start[p + 1]) ++p;
  const int local = b - ga.start[p];
  gemm_f32_block<A_KC, B_KC>(ga.p[p], local % ga.tiles[p], local / ga.tiles[p], lds);
}

// ... with the BatchNorm-backward column sums of each problem's producing layer in the epilogue (BnBwdEpi per problem)
template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(kBlock)
gemm_f32_grouped_bn_bwd_kernel(GroupedArgs ga) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kOpTile];
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < ga.n && b >= ga.start[p + 1]) ++p;
  const int local = b - ga.start[p];
  gemm_f32_block<A_KC, B_KC, true>(ga.p[p], local % ga.tiles[p], local / ga.tiles[p], lds);
}

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(kBlock)
gemm_f32_grouped_tr_kernel(GroupedArgs ga) {
  __shared__ __attribute__((aligned(16))) float lds[kTrLds];
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < ga.n && b >= ga.start[p + 1]) ++p;
  const int local = b - ga.start[p];
  gemm_f32_block<A_KC, B_KC, false, true>(ga.p[p], local % ga.tiles[p], local / ga.tiles[p], lds);
}

// ------------------------------------------------------------------------------------------------
// bf16 inputs (rounded from fp32 while staging), fp32 accumulate.  LDS: As[m][k], Bs[n][k] in bf16, row
// stride 40 halves (80 B): the 16-byte fragment reads of a 16-lane group hit 16 distinct bank quads.
// Fragment of v_mfma_f32_32x32x16_bf16: lane l holds A[i = l & 31][k = 8 * (l >> 5) + 0..7].
// ------------------------------------------------------------------------------------------------
constexpr int kBf16SH = BK16 + 8;  // halves per LDS row
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_bf16_block(const GemmArgs& g, int bx, int bz, short* __restrict__ As, short* __restrict__ Bs) {
  constexpr int SH = kBf16SH;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tx, ty;
  tile_coords(bx, static_cast<int>(ceil_div(g.N, BN)), static_cast<int>(ceil_div(g.M, BM)), tx, ty);
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
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < ga.n && b >= ga.start[p + 1]) ++p;
  const int local = b - ga.start[p];
  gemm_bf16_block<A_KC, B_KC>(ga.p[p], local % ga.tiles[p], local / ga.tiles[p], As, Bs);
}

// C[i, j] (+)= bias[j] + sum_s ws[s, i, j]   (split order fixed: deterministic).  VEC = 4: float4 per lane, the
// `splits` loads of a lane are independent and unrolled by 4.
template <int VEC>
__device__ __forceinline__ void splitk_reduce_elems(const float* __restrict__ ws, int64_t mn, int N, int splits,
                                                    const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                    int accumulate, int64_t i) {
  if (i >= mn) return;
  float s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
#pragma unroll 4
  for (int z = 0; z < splits; ++z) {
    if (VEC == 4) {
      const f32x4v v = *reinterpret_cast<const f32x4v*>(ws + z * mn + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] = s[j] + v[j];
    } else {
      s[0] = s[0] + ws[z * mn + i];
    }
  }
  const int64_t row = i / N;
  const int col = static_cast<int>(i % N);
  float* p = C + row * ldc + col;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float v = s[j];
    if (bias) v = v + bias[col + j];
    p[j] = accumulate ? p[j] + v : v;
  }
}

template <int VEC>
__global__ void __launch_bounds__(kBlock)
gemm_splitk_reduce_kernel(const float* __restrict__ ws, int64_t mn, int N, int splits, const float* __restrict__ bias,
                          float* __restrict__ C, int ldc, int accumulate) {
  splitk_reduce_elems<VEC>(ws, mn, N, splits, bias, C, ldc, accumulate,
                           (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * VEC);
}

// the reduces of a grouped launch in one grid: workgroups [start[p], start[p+1]) own the output of item p
struct ReduceItem {
  const float* ws;
  int64_t mn;
  int N, splits;
  const float* bias;
  float* C;
  int ldc, accumulate;
  int vec;  // 16-byte lanes (N % 4 == 0, ldc % 4 == 0, C 16-byte aligned: decided per problem)
};
struct GroupedReduceArgs {
  int n;
  int start[kMaxGroup + 1];
  ReduceItem r[kMaxGroup];
};

template <int VEC>
__global__ void __launch_bounds__(kBlock)
gemm_splitk_reduce_grouped_kernel(GroupedReduceArgs ra) {
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < ra.n && b >= ra.start[p + 1]) ++p;
  const ReduceItem& r = ra.r[p];
  const int64_t lane = static_cast<int64_t>(b - ra.start[p]) * kBlock + threadIdx.x;
  if (VEC == 4 && r.vec) splitk_reduce_elems<4>(r.ws, r.mn, r.N, r.splits, r.bias, r.C, r.ldc, r.accumulate, lane * 4);
  else splitk_reduce_elems<1>(r.ws, r.mn, r.N, r.splits, r.bias, r.C, r.ldc, r.accumulate, lane);
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

// barrier words of the fused-BatchNorm epilogues: [kMaxColTiles][2], zero between launches
constexpr int kMaxColTiles = 256;
unsigned* g_bn_counters = nullptr;
int g_num_cus = 0;

int ensure_counters() {
  if (!g_bn_counters) {
    ER_CHECK_HIP(hipMalloc(&g_bn_counters, sizeof(unsigned) * 2 * kMaxColTiles));
    ER_CHECK_HIP(hipMemset(g_bn_counters, 0, sizeof(unsigned) * 2 * kMaxColTiles));
    int dev = 0;
    hipDeviceProp_t prop;
    ER_CHECK_HIP(hipGetDevice(&dev));
    ER_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    g_num_cus = prop.multiProcessorCount;
  }
  return 0;
}

// The barrier needs every workgroup of the grid resident at once: 36 KB of LDS and 256 threads per workgroup leave
// room for at least 2 per CU; stay at that.
bool fused_bn_fits(int M, int N) {
  if (!g_bn_counters || g_num_cus <= 0) return false;
  const int64_t gx = er::ceil_div(N, er::BN), gy = er::ceil_div(M, er::BM);
  return gx <= kMaxColTiles && gx * gy <= 2LL * g_num_cus;
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
  } else if (a.at.mean) {
    ER_LAUNCH_GEMM(er::gemm_f32_tr_kernel)
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
               const er::BnBwdEpi* bn = nullptr, const er::ATransform* at = nullptr, const er::BnFused* fu = nullptr) {
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
  if (bn) a.bn = *bn;
  if (at) a.at = *at;
  if (fu) a.fu = *fu;
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

int gemm_grouped_f32(int layout, const er_gemm_problem* pr, int n, er_stream_t stream, bool bf16 = false) {
  hipStream_t s = er::as_stream(stream);
  int64_t total_tiles = 0;
  for (int i = 0; i < n; ++i) {
    const er_gemm_problem& q = pr[i];
    ER_REQUIRE(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K > 0, "er_gemm_grouped_f32: problem %d: bad arguments", i);
    const int min_lda = (layout == ER_GEMM_TN) ? q.M : q.K;
    const int min_ldb = (layout == ER_GEMM_NT) ? q.K : q.N;
    ER_REQUIRE(q.lda >= min_lda && q.ldb >= min_ldb && q.ldc >= q.N,
               "er_gemm_grouped_f32: problem %d: leading dimension too small", i);
    ER_REQUIRE(!bf16 || !(q.a_mean || q.bn_partial), "er_gemm_grouped_bf16: problem %d: epilogues / transforms are fp32 only", i);
    total_tiles += er::ceil_div(q.M, er::BM) * er::ceil_div(q.N, er::BN);
  }
  // k-splits: enough workgroups for ~2 per CU over the whole group (A/B: 512 beat 1024 and 2048), >= 4 k-tiles per split
  static const int64_t target_blocks = [] {  // (A/B knob)
    const char* e = getenv("ER_WGRAD_TARGET_BLOCKS");
    const int64_t v = e ? atoll(e) : 0;
    return v >= 1 ? v : 512;
  }();
  int64_t want = total_tiles >= target_blocks ? 1 : er::ceil_div(target_blocks, total_tiles);
  if (want > 64) want = 64;
  er::GroupedArgs ga;
  er::GroupedReduceArgs ra;
  ga.n = 0;
  ga.start[0] = 0;
  er::GemmArgs* slot[er::kMaxGroup];  // where problem i's arguments live (the workspace base is patched in below)
  ra.n = 0;
  ra.start[0] = 0;
  size_t ws_floats = 0;
  bool any_tr = false, any_bn = false;
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
      ER_REQUIRE(q.bn_z && q.bn_ld >= q.N && !q.accumulate && (!q.bn_use_bn || (q.bn_mean && q.bn_invstd)),
                 "er_gemm_grouped_f32: problem %d: bad BatchNorm-backward epilogue arguments", i);
      a.bn.z = q.bn_z; a.bn.zbias = q.bn_zbias; a.bn.y = q.bn_y; a.bn.mean = q.bn_mean; a.bn.invstd = q.bn_invstd;
      a.bn.gamma = q.bn_gamma; a.bn.beta = q.bn_beta; a.bn.ld = q.bn_ld; a.bn.use_bn = q.bn_use_bn; a.bn.act = q.bn_act;
      a.bn.partial = q.bn_partial;
      any_bn = true;
    }
    if (q.a_mean) {
      ER_REQUIRE(q.a_invstd, "er_gemm_grouped_f32: problem %d: A transform without invstd", i);
      ER_REQUIRE(layout == ER_GEMM_TN || q.K <= er::kTrMaxK - 64,
                 "er_gemm_grouped_f32: problem %d: A transform over K = %d features (limit %d)", i, q.K, er::kTrMaxK - 64);
      a.at.mean = q.a_mean; a.at.invstd = q.a_invstd; a.at.gamma = q.a_gamma; a.at.beta = q.a_beta; a.at.act = q.a_act;
      any_tr = true;
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
    if (sp < 1 || q.col_stats || q.bn_partial) sp = 1;
    a.k_per_split = static_cast<int>(er::ceil_div(er::ceil_div(q.K, sp), er::BK32)) * er::BK32;
    a.splits = static_cast<int>(er::ceil_div(q.K, a.k_per_split));
    const int64_t tiles = er::ceil_div(q.M, er::BM) * er::ceil_div(q.N, er::BN);
    grp.tiles[grp.n] = static_cast<int>(tiles);
    grp.start[grp.n + 1] = grp.start[grp.n] + grp.tiles[grp.n] * a.splits;
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
  dim3 grid(static_cast<unsigned>(ga.start[ga.n])), block(er::kBlock);
  if (bf16) {
    switch (layout) {
      case ER_GEMM_NN: hipLaunchKernelGGL((er::gemm_bf16_grouped_kernel<true, false>), grid, block, 0, s, ga); break;
      case ER_GEMM_NT: hipLaunchKernelGGL((er::gemm_bf16_grouped_kernel<true, true>), grid, block, 0, s, ga); break;
      case ER_GEMM_TN: hipLaunchKernelGGL((er::gemm_bf16_grouped_kernel<false, false>), grid, block, 0, s, ga); break;
      default: er::set_error("er_gemm_grouped_bf16: unknown layout %d", layout); return 2;
    }
  } else if (any_bn) {
    ER_REQUIRE(!any_tr, "er_gemm_grouped_f32: the A transform and the BatchNorm-backward epilogue in one launch");
    switch (layout) {
      case ER_GEMM_NN: hipLaunchKernelGGL((er::gemm_f32_grouped_bn_bwd_kernel<true, false>), grid, block, 0, s, ga); break;
      case ER_GEMM_NT: hipLaunchKernelGGL((er::gemm_f32_grouped_bn_bwd_kernel<true, true>), grid, block, 0, s, ga); break;
      case ER_GEMM_TN: hipLaunchKernelGGL((er::gemm_f32_grouped_bn_bwd_kernel<false, false>), grid, block, 0, s, ga); break;
      default: er::set_error("er_gemm_grouped_f32: unknown layout %d", layout); return 2;
    }
  } else if (any_tr) {
    switch (layout) {
      case ER_GEMM_NN: hipLaunchKernelGGL((er::gemm_f32_grouped_tr_kernel<true, false>), grid, block, 0, s, ga); break;
      case ER_GEMM_NT: hipLaunchKernelGGL((er::gemm_f32_grouped_tr_kernel<true, true>), grid, block, 0, s, ga); break;
      case ER_GEMM_TN: hipLaunchKernelGGL((er::gemm_f32_grouped_tr_kernel<false, false>), grid, block, 0, s, ga); break;
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
  if (ra.n > 0) {
    dim3 rgrid(static_cast<unsigned>(ra.start[ra.n]));
    hipLaunchKernelGGL(er::gemm_splitk_reduce_grouped_kernel<4>, rgrid, block, 0, s, ra);  // (16-byte lanes per problem: r.vec)
    ER_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

extern "C" {

int er_gemm_reserve(int64_t floats) {
  ER_REQUIRE(floats >= 0, "er_gemm_reserve: negative size");
  if (int rc = ensure_counters()) return rc;  // (allocations are not capturable: both happen here)
  float* p;
  return ensure_ws(static_cast<size_t>(floats), &p);
}

int er_gemm_fused_bn_ok(int32_t M, int32_t N) { return fused_bn_fits(M, N) ? 1 : 0; }

int er_gemm_f32_bn_fwd(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                       int32_t ldb, float* Z, int32_t ldz, const float* bias, float* col_stats, const float* gamma,
                       const float* beta, float eps, float momentum, float* moving_mean, float* moving_var, int act,
                       float* Y, int32_t ldy, float* save_mean, float* save_invstd, er_stream_t stream) {
  ER_REQUIRE(A && B && Z && Y && col_stats && save_mean && save_invstd && ldy >= N, "er_gemm_f32_bn_fwd: bad arguments");
  ER_REQUIRE(fused_bn_fits(M, N), "er_gemm_f32_bn_fwd: %d x %d outputs do not fit one co-resident grid (er_gemm_fused_bn_ok); "
             "call er_gemm_reserve first", M, N);
  ER_REQUIRE(layout >= ER_GEMM_NN && layout <= ER_GEMM_TN, "er_gemm_f32_bn_fwd: unknown layout %d", layout);
  const int min_lda = (layout == ER_GEMM_TN) ? M : K;
  const int min_ldb = (layout == ER_GEMM_NT) ? K : N;
  ER_REQUIRE(lda >= min_lda && ldb >= min_ldb && ldz >= N, "er_gemm_f32_bn_fwd: leading dimension too small");
  er::GemmArgs a;
  a.A = A; a.B = B; a.C = Z; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldz;
  a.accumulate = 0;
  a.col_stats = col_stats;
  a.splits = 1;
  a.k_per_split = static_cast<int>(er::ceil_div(K, er::BK32)) * er::BK32;
  a.fu.mode = 1;
  a.fu.gamma = gamma; a.fu.beta = beta; a.fu.eps = eps; a.fu.momentum = momentum;
  a.fu.moving_mean = moving_mean; a.fu.moving_var = moving_var;
  a.fu.save_mean = save_mean; a.fu.save_invstd = save_invstd;
  a.fu.y = Y; a.fu.ldy = ldy; a.fu.act = act;
  a.fu.counters = g_bn_counters;
  return launch_gemm<false>(layout, a, er::as_stream(stream));
}

int er_gemm_f32_bn_bwd_apply(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                             int32_t ldb, float* DZ, int32_t ldc, const float* z, const float* z_bias, const float* y,
                             const float* save_mean, const float* save_invstd, int32_t ld_zy, int use_bn, int act,
                             const float* gamma, float* partial, float* dgamma, float* dbeta, float* dbias,
                             int accumulate, er_stream_t stream) {
  ER_REQUIRE(A && B && DZ && z && y && partial && ld_zy >= N && ldc >= N, "er_gemm_f32_bn_bwd_apply: bad arguments");
  ER_REQUIRE(!use_bn || (save_mean && save_invstd), "er_gemm_f32_bn_bwd_apply: BatchNorm statistics missing");
  ER_REQUIRE(fused_bn_fits(M, N), "er_gemm_f32_bn_bwd_apply: %d x %d outputs do not fit one co-resident grid", M, N);
  ER_REQUIRE(layout >= ER_GEMM_NN && layout <= ER_GEMM_TN, "er_gemm_f32_bn_bwd_apply: unknown layout %d", layout);
  const int min_lda = (layout == ER_GEMM_TN) ? M : K;
  const int min_ldb = (layout == ER_GEMM_NT) ? K : N;
  ER_REQUIRE(lda >= min_lda && ldb >= min_ldb, "er_gemm_f32_bn_bwd_apply: leading dimension too small");
  er::GemmArgs a;
  a.A = A; a.B = B; a.C = DZ; a.bias = nullptr;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.accumulate = 0;
  a.col_stats = nullptr;
  a.splits = 1;
  a.k_per_split = static_cast<int>(er::ceil_div(K, er::BK32)) * er::BK32;
  a.bn.z = z; a.bn.zbias = z_bias; a.bn.y = y; a.bn.mean = save_mean; a.bn.invstd = save_invstd;
  a.bn.ld = ld_zy; a.bn.use_bn = use_bn; a.bn.act = act; a.bn.partial = partial;
  a.fu.mode = 2;
  a.fu.gamma = gamma; a.fu.dgamma = dgamma; a.fu.dbeta = dbeta; a.fu.dbias = dbias; a.fu.accumulate = accumulate;
  a.fu.counters = g_bn_counters;
  return launch_gemm<false>(layout, a, er::as_stream(stream));
}

int er_gemm_f32_deferred(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const er_a_transform* at,
                         const float* B, int32_t ldb, float* C, int32_t ldc, const float* bias, int accumulate,
                         float* col_stats, const er_bn_finalize* fin, er_stream_t stream) {
  ER_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "er_gemm_f32_deferred: bad arguments");
  ER_REQUIRE(layout >= ER_GEMM_NN && layout <= ER_GEMM_TN, "er_gemm_f32_deferred: unknown layout %d", layout);
  const int min_lda = (layout == ER_GEMM_TN) ? M : K;
  const int min_ldb = (layout == ER_GEMM_NT) ? K : N;
  ER_REQUIRE(lda >= min_lda && ldb >= min_ldb && ldc >= N, "er_gemm_f32_deferred: leading dimension too small");
  ER_REQUIRE(!fin || (col_stats && fin->save_mean && fin->save_invstd && fin->counters && !accumulate),
             "er_gemm_f32_deferred: finalising the statistics needs col_stats, save_mean, save_invstd, counters and a plain output");
  ER_REQUIRE(!fin || er::ceil_div(N, er::BN) <= fin->n_counters / 2,
             "er_gemm_f32_deferred: %d column tiles need %d counter words", static_cast<int>(er::ceil_div(N, er::BN)),
             static_cast<int>(2 * er::ceil_div(N, er::BN)));
  er::ATransform tr;
  if (at && at->mean) {
    ER_REQUIRE(at->invstd, "er_gemm_f32_deferred: A transform without invstd");
    ER_REQUIRE(layout != ER_GEMM_NT, "er_gemm_f32_deferred: the A transform is defined for [batch, features] operands (NN, TN)");
    ER_REQUIRE(layout == ER_GEMM_TN || K <= er::kTrMaxK - 64, "er_gemm_f32_deferred: A transform over K = %d features (limit %d)",
               K, er::kTrMaxK - 64);
    tr.mean = at->mean; tr.invstd = at->invstd; tr.gamma = at->gamma; tr.beta = at->beta; tr.act = at->act;
  }
  er::BnFused fu;
  if (fin) {
    fu.mode = 3;
    fu.eps = fin->eps; fu.momentum = fin->momentum;
    fu.moving_mean = fin->moving_mean; fu.moving_var = fin->moving_var;
    fu.save_mean = fin->save_mean; fu.save_invstd = fin->save_invstd;
    fu.counters = reinterpret_cast<unsigned*>(fin->counters);
  }
  // (the k-split of a plain contraction - hence its summation order - is er_gemm_f32's)
  return gemm_entry<false>(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, col_stats, stream, "er_gemm_f32_deferred",
                           nullptr, tr.mean ? &tr : nullptr, fin ? &fu : nullptr);
}

int er_gemm_f32_bn_bwd_z(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                         int32_t ldb, float* C, int32_t ldc, const float* z, const float* z_bias, const float* gamma,
                         const float* beta, const float* save_mean, const float* save_invstd, int32_t ld_z, int use_bn,
                         int act, float* partial, er_stream_t stream) {
  ER_REQUIRE(z && partial && ld_z >= N, "er_gemm_f32_bn_bwd_z: bad epilogue arguments");
  ER_REQUIRE(!use_bn || (save_mean && save_invstd), "er_gemm_f32_bn_bwd_z: BatchNorm statistics missing");
  er::BnBwdEpi e;
  e.z = z; e.zbias = z_bias; e.y = nullptr; e.mean = save_mean; e.invstd = save_invstd;
  e.gamma = gamma; e.beta = beta;
  e.ld = ld_z; e.use_bn = use_bn; e.act = act; e.partial = partial;
  return gemm_entry<false>(layout, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, nullptr, stream, "er_gemm_f32_bn_bwd_z", &e);
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

int er_gemm_row_tiles(int32_t M) { return static_cast<int>(er::ceil_div(M, er::BM)); }

}  // extern "C"
