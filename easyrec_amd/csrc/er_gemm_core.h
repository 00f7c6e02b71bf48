// The device core of K13 (er_gemm.hip): the fp32 MFMA tile body, its epilogues and the grouped-launch records, shared
// with the launches that carry a GEMM next to other work in ONE grid (er_embedding.hip: the weight gradients of a step
// beside the embedding row update).  Design notes: er_gemm.hip.
#pragma once
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
  float* partial = nullptr;       // [row tiles][n_src][2]
  // The producing layer's output may be a COLUMN BLOCK of this GEMM's output position: output column c belongs to source
  // column c - col0 when 0 <= c - col0 < n_src (DeepFM: the deep tower's 64 columns inside d[sum(wide) | FM | deep],
  // model/deepfm.py:75-83); z / y / mean / invstd / gamma / beta / zbias / partial are the SOURCE layer's, n_src wide.
  int col0 = 0, n_src = 0;        // n_src == 0: the whole output (n_src = N, set by the host)
  // dz_out != 0 (the producing layer normalises with the MOVING statistics - the experts of a multi-task model; col0 == 0,
  // n_src == N): its whole backward is elementwise, dz = gamma * invstd * g with g the activation-masked accumulator, so the
  // launch stores THAT instead of the accumulator and the layer's own backward pass disappears (its parameter gradients
  // come from `partial`)
  const float* gamma = nullptr;
  int dz_out = 0;
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
  // fz_y != nullptr (kernels instantiated with XEPI == kEpiFrozenBn, forward launches: bn is free there): the output ALSO goes
  // through bias + BatchNorm on the MOVING statistics + activation in the epilogue (the experts of the reference's MMoE /
  // DBMTL, layers/mmoe.py:62-83: batch_normalization(training=False) inside the training graph) - C keeps the contraction z,
  // fz_y [M][ldc] gets y = act((((z + bias) - mean) * invstd) * gamma + beta), bn_frozen_apply_body's arithmetic with
  //   bn.zbias = bias, bn.z = gamma, bn.y = beta, bn.mean = moving_mean, bn.invstd = moving VARIANCE, bn.act = activation;
  // row tile 0 writes fz_save[0 .. N) = mean, fz_save[N .. 2N) = 1 / sqrt(var + fz_eps) for the backward.
  float* fz_y = nullptr;
  float* fz_save = nullptr;
  float fz_eps = 0.f;
};
constexpr int kEpiFrozenBn = 7;  // (an XEPI value beside ER_EPI_CROSS_FWD / _BWD)
constexpr int kEpiFrozenDz = 8;  // (with BN_EPI: BnBwdEpi.dz_out is honoured - its own instantiation: inside the plain BN_EPI
                                 //  kernels the extra branch kept y's 16 values alive past the column sums and cost DeepFM's five
                                 //  input-gradient launches 40.8 -> 47.5 us)

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
//
// plain: the caller has placed the workgroup itself (a grouped launch's XCD region: grouped_coords) - tile i as it is.
__device__ __forceinline__ void tile_coords(int i, int gx, int gy, int& tx, int& ty, bool plain = false) {
  const int nt = gx * gy;
  const int per = nt / 8;
  const int t = (!plain && i < 8 * per) ? (i % 8) * per + i / 8 : i;
  tx = t % gx;
  ty = t / gx;
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
                                               float* __restrict__ out_tile /* [N][3] of this row tile */) {
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
    o[0] = an; o[1] = am; o[2] = a2;
  }
}

// BatchNorm-backward column sums of a 64-row output tile (see BnBwdEpi).  Fixed order: a lane's 16 rows in
// register order, then the lane pair (l, l ^ 32), then the two waves that share the columns.
__device__ __forceinline__ void tile_bn_bwd_partial(const f32x16& acc, const BnBwdEpi& e, const float (&py)[16],
                                                    const float (&pz)[16], int row_base, int M, int col, int N, int wm,
                                                    int wn, int lane, float* lds, int ty) {
  const int khalf = lane >> 5;
  float sg = 0.f, sgx = 0.f;
  const bool col_ok = col >= 0 && col < N;  // (col, N: the SOURCE column and width - BnBwdEpi.col0)
  if (col_ok) {
    const float bv = e.zbias ? e.zbias[col] : 0.f;
    const float mu = e.use_bn ? e.mean[col] : 0.f;
    const float is = e.use_bn ? e.invstd[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (row < M) {
        float g = acc[r];
        if (e.act == ER_ACT_RELU && !(py[r] > 0.f)) g = 0.f;
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
  if (wm == 0 && khalf == 0 && col_ok) {
    float* p = e.partial + (static_cast<int64_t>(ty) * N + col) * 2;
    p[0] = a + slot[0];
    p[1] = ax + slot[1];
  }
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
    // branch-free: an element outside the operand is loaded from a clamped (valid) address and zeroed by a select - a load
    // inside `if (inside)` is waited for inside its branch (s_waitcnt vmcnt(0) behind every global_load_dword in the ISA):
    // 16 dependent round trips per thread and k-tile on e.g. DeepFM's [B, 81] -> 256 layer (lda = 81)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bool ok;
      int64_t at;
      if (KC) {
        ok = mn < MN && k + j < kend;
        at = static_cast<int64_t>(mn < MN ? mn : MN - 1) * ld + (k + j < kend ? k + j : kend - 1);
      } else {
        ok = mn + j < MN && k < kend;
        at = static_cast<int64_t>(k < kend ? k : kend - 1) * ld + (mn + j < MN ? mn + j : MN - 1);
      }
      const float v = P[at];
      r[i][j] = ok ? v : 0.f;
    }
  }
}

template <bool KC>
__device__ __forceinline__ void stage_tile(float* __restrict__ S, int tid, const f32x4v (&r)[2], bool interior, int mn0,
                                           int MN, int k0, int kend) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, k;
    unit_pos<KC>(tid, i, row, k);
    f32x4v v = r[i];
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

// DIN's attention input [q, h, q - h, q * h] ([B * L, 4E], reference model/multi_tower_din.py:62-80) as a GENERATED operand:
// the contraction reads q [B, E] and h [B * L, E] and forms the 4E columns while staging, so the block never exists in HBM.
//   forward  (NN): z1 = [q, h, q - h, q * h] . W1           A_KC: 4 consecutive concat columns of one row per unit
//   wgrad    (TN): dW1 = [q, h, q - h, q * h]^T . dz1        !A_KC: 4 consecutive concat columns of one row per unit too
// and, for the input gradient dcat = dz1 . W1^T (NT), an epilogue that reduces the four column segments of dcat to dh and
// to per-tile partial sums of dq (DinBwd) - with W1's rows permuted so that one 64-column tile holds all four segments of 16
// embedding positions.
struct DinGen {
  const float* q;   // [B][ldq]
  const float* h;   // [B * L][ldh]
  int ldq, ldh, L, E;
  uint32_t inv_L;   // ceil(2^32 / L): row / L = (row * inv_L) >> 32 for row < 2^32 / L
  uint32_t inv_E;   // ceil(2^32 / E): column / E likewise (columns < 4E)
  // input-gradient epilogue (DinBwd)
  float* dh;        // [B * L][lddh]: dh (+)= d1 - d2 + q * d3
  int lddh, accumulate_dh;
  float* dq_partial;  // [row tiles][slots][E]: per 64-row tile and example slot, sum over the tile's rows of d0 + d2 + h * d3
  int slots;          // (63 / L) + 2: the examples a 64-row tile can touch
};
__device__ __forceinline__ int din_div_L(const DinGen& d, int row) {
  return static_cast<int>((static_cast<uint64_t>(static_cast<uint32_t>(row)) * d.inv_L) >> 32);
}
// the generated 4 consecutive concat columns [c, c + 4) of a row from its q / h pieces (c % 4 == 0, E % 4 == 0: one segment)
__device__ __forceinline__ f32x4v din_combine(int seg, const f32x4v& qv, const f32x4v& hv) {
  // branch-free: both derived pieces are computed, two selects pick (nested ternaries compiled to exec-mask branches per
  // lane and element: 65 branches in the k loop)
  const bool odd = (seg & 1) != 0, hi = (seg & 2) != 0;
  f32x4v v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float sub = qv[j] - hv[j], mul = qv[j] * hv[j];
    const float lo_v = odd ? hv[j] : qv[j];
    const float hi_v = odd ? mul : sub;
    v[j] = hi ? hi_v : lo_v;
  }
  return v;
}
// What is fixed for a thread's 2 units of the generated A tile over the whole k loop (computed once: the per-tile work is
// one multiply-high per unit - an integer division per unit and tile made the first version of these kernels VALU-bound,
// 146 us for the weight gradient of a [204800 x 128] layer against 52 us over the built block).
//   KC  (forward):          the unit's row r and example b are fixed, its concat column moves with the k-tile
//   !KC (weight gradient):  the unit's concat column (segment, position j) is fixed, its row moves with the k-tile
struct DinPre {
  const float* qp[2];  // KC: q + b * ldq;            !KC: q + j
  const float* hp[2];  // KC: h + r * ldh;            !KC: h + j
  int off[2];          // KC: the unit's k offset inside a k-tile;  !KC: its row offset inside a k-tile
  int seg[2];          // !KC: the fixed segment
};
template <bool KC>
__device__ __forceinline__ void din_prepare(const DinGen& d, int mn0, int MN, int tid, DinPre& p) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, k;
    unit_pos<KC>(tid, i, row, k);
    p.off[i] = k;
    if (KC) {
      int r = mn0 + row;
      r = r < MN ? r : MN - 1;
      const int b = din_div_L(d, r);
      p.qp[i] = d.q + static_cast<int64_t>(b) * d.ldq;
      p.hp[i] = d.h + static_cast<int64_t>(r) * d.ldh;
      p.seg[i] = 0;
    } else {
      int c = mn0 + row;
      c = c < MN - 4 ? c : MN - 4;
      const int seg = static_cast<int>((static_cast<uint64_t>(static_cast<uint32_t>(c)) * d.inv_E) >> 32);
      p.seg[i] = seg;
      p.qp[i] = d.q + (c - seg * d.E);
      p.hp[i] = d.h + (c - seg * d.E);
    }
  }
}
// loads of a thread's 2 units of the GENERATED A tile (clamped like fetch_tile; the values are combined at staging)
template <bool KC>
__device__ __forceinline__ void fetch_din(const DinGen& d, const DinPre& p, int k0, int kend, int K, f32x4v (&rh)[2],
                                          f32x4v (&rq)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (KC) {
      int c = k0 + p.off[i];
      c = c < K - 4 ? c : K - 4;
      const int seg = static_cast<int>((static_cast<uint64_t>(static_cast<uint32_t>(c)) * d.inv_E) >> 32);
      const int j = c - seg * d.E;
      rq[i] = *reinterpret_cast<const f32x4v*>(p.qp[i] + j);
      rh[i] = *reinterpret_cast<const f32x4v*>(p.hp[i] + j);
    } else {
      int r = k0 + p.off[i];
      r = r < kend ? r : kend - 1;
      const int b = din_div_L(d, r);
      rq[i] = *reinterpret_cast<const f32x4v*>(p.qp[i] + static_cast<int64_t>(b) * d.ldq);
      rh[i] = *reinterpret_cast<const f32x4v*>(p.hp[i] + static_cast<int64_t>(r) * d.ldh);
    }
  }
}
template <bool KC>
__device__ __forceinline__ void combine_din(const DinGen& d, const DinPre& p, int k0, int K, f32x4v (&rh)[2],
                                            const f32x4v (&rq)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int seg = p.seg[i];
    if (KC) {
      int c = k0 + p.off[i];
      c = c < K - 4 ? c : K - 4;
      seg = static_cast<int>((static_cast<uint64_t>(static_cast<uint32_t>(c)) * d.inv_E) >> 32);
    }
    rh[i] = din_combine(seg, rq[i], rh[i]);
  }
}
// B rows of the input-gradient contraction in the permuted order: tile column c' of column tile tx is concat column
// (c' / 16) * E + tx * 16 + c' % 16 (all four segments of 16 embedding positions in one 64-column tile)
__device__ __forceinline__ void fetch_tile_din_perm(const float* __restrict__ P, int ld, int tx, int E, int k0, int kend, int K,
                                                    int tid, f32x4v (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, k;
    unit_pos<true>(tid, i, row, k);
    const int n = (row >> 4) * E + tx * 16 + (row & 15);
    k += k0;
    const int kpad = (K + 3) & ~3;
    k = k < kpad - 4 ? k : kpad - 4;
    r[i] = *reinterpret_cast<const f32x4v*>(P + static_cast<int64_t>(n) * ld + k);
  }
}

// The A operand as the PRODUCING layer's pre-normalisation values z with that layer's BatchNorm + activation applied while
// staging (er_gemm_f32_bn_a; reference layers/dnn.py:57-79: dense -> batch_normalization -> relu, then the next dense): a
// tall layer's activations y = act(((z - mean) * invstd) * gamma + beta) are then never read back from HBM by a launch of
// their own - the column tile tx == 0 of every row tile stores them (the backward reads y) while it stages them.  The
// arithmetic is bn_act_one's (er_dense.hip), operation by operation: the same bits as bn_finalize_apply_kernel writes.
struct BnA {
  const float* mean;    // [K] of the producing layer (er_bn_finalize_from_stats)
  const float* invstd;  // [K]
  const float* gamma;   // [K] or nullptr (ones)
  const float* beta;    // [K] or nullptr (zeros)
  int act;
  float* y;             // [M][ldy] or nullptr
  int ldy;
};
constexpr int kBnAMaxK = 256;  // coefficient table in LDS: [k][mean, invstd, gamma, beta]
__device__ __forceinline__ void bna_load_coef(const BnA& b, int K, float* __restrict__ coef) {
  for (int k = threadIdx.x; k < ((K + 3) & ~3); k += kBlock) {
    const bool in = k < K;
    f32x4v c = {in ? b.mean[k] : 0.f, in ? b.invstd[k] : 0.f, (in && b.gamma) ? b.gamma[k] : 1.f, (in && b.beta) ? b.beta[k] : 0.f};
    *reinterpret_cast<f32x4v*>(coef + 4 * k) = c;
  }
}
// a thread's 2 units (4 consecutive k of one row each, the same k offset) of the fetched z tile -> y; stored when `store`
__device__ __forceinline__ void bna_apply(const BnA& b, const float* __restrict__ coef, int tid, int m0, int M, int k0, int K,
                                          bool store, f32x4v (&r)[2]) {
  int kk = k0 + (tid & 7) * 4;
  const int kpad = (K + 3) & ~3;
  kk = kk < kpad - 4 ? kk : kpad - 4;  // (what fetch_tile read)
  f32x4v c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = *reinterpret_cast<const f32x4v*>(coef + 4 * (kk + j));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f32x4v v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = (r[i][j] - c[j][0]) * c[j][1];
      t = t * c[j][2] + c[j][3];
      if (b.act == ER_ACT_RELU) t = t > 0.f ? t : 0.f;
      v[j] = t;
    }
    r[i] = v;
    const int row = m0 + ((tid + i * kBlock) >> 3);
    if (store && row < M && k0 + (tid & 7) * 4 < K)
      *reinterpret_cast<f32x4v*>(b.y + static_cast<int64_t>(row) * b.ldy + kk) = v;
  }
}

// bx: index of the workgroup among the problem's (8-rounded) tiles, bz: its k-split.  BN_EPI: with the BnBwdEpi
// epilogue - the y / z values of the lane's 16 output positions are requested BEFORE the k loop so that their
// latency hides behind it (32 more VGPRs: a separate instantiation).
// XEPI: ER_EPI_CROSS_FWD / ER_EPI_CROSS_BWD with the record *xe (er_gemm_f32_cross: the DCN-v2 cross layer's elementwise
// part inside its contraction, reference layers/keras/interaction.py:276-286) - like BN_EPI, what the epilogue reads at the
// lane's 16 output positions is requested before the k loop.
// DIN: 1 = the A operand is DinGen's generated block (forward NN, weight gradient TN); 2 = the input-gradient contraction
// (NT) with permuted B rows and the epilogue that reduces dcat to dh / dq partials (nothing is stored to C); 3 = the A
// operand is a BatchNorm'd layer's z, normalised while staging (BnA *ba, its coefficient table `coef` in LDS; NN, K % 4 == 0).
template <bool A_KC, bool B_KC, bool BN_EPI = false, int XEPI = 0, int DIN = 0>
__device__ __forceinline__ void gemm_f32_block(const GemmArgs& g, int bx, int bz, float* __restrict__ lds,
                                               bool plain_tiles = false, const er_gemm_epilogue* xe = nullptr,
                                               const DinGen* dg = nullptr, const BnA* ba = nullptr,
                                               const float* __restrict__ coef = nullptr) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tx, ty;
  tile_coords(bx, static_cast<int>(ceil_div(g.N, BN)), static_cast<int>(ceil_div(g.M, BM)), tx, ty, plain_tiles);
  const int m0 = ty * BM, n0 = tx * BN;
  const int kbeg = bz * g.k_per_split;
  int kend = kbeg + g.k_per_split;
  if (kend > g.K) kend = g.K;
  const int T = (kend - kbeg + BK32 - 1) / BK32;
  float py[16], pz[16];
  if (BN_EPI && g.bn.partial != nullptr) {  // (a grouped launch may mix problems with and without the epilogue)
    int c = n0 + wn * 32 + (lane & 31) - g.bn.col0;  // (source column, clamped: a lane outside the block loads what it ignores)
    c = c < 0 ? 0 : (c < g.bn.n_src ? c : g.bn.n_src - 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      row = row < g.M ? row : g.M - 1;
      const int64_t i = static_cast<int64_t>(row) * g.bn.ld + c;
      py[r] = g.bn.y[i];
      pz[r] = g.bn.z[i];
    }
  }
  // cross epilogues: x0 / x_l (forward), dout / du_in and the lower layer's x0 / u / x_l / dx0 (backward) of the 16 positions
  float qa[XEPI ? 16 : 1], qb[XEPI ? 16 : 1], qc[XEPI == ER_EPI_CROSS_BWD ? 16 : 1], qd[XEPI == ER_EPI_CROSS_BWD ? 16 : 1],
      qe[XEPI == ER_EPI_CROSS_BWD ? 16 : 1], qf[XEPI == ER_EPI_CROSS_BWD ? 16 : 1];
  if (XEPI && XEPI != kEpiFrozenBn && XEPI != kEpiFrozenDz) {
    int c = n0 + wn * 32 + (lane & 31);
    c = c < g.N ? c : g.N - 1;
    const bool with_diag = xe->diag != 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      row = row < g.M ? row : g.M - 1;
      const int64_t rr = row;
      if (XEPI == ER_EPI_CROSS_FWD) {
        qa[r] = xe->x0[rr * xe->ld_x0 + c];
        qb[r] = xe->xl[rr * xe->ld_xl + c];
      } else {
        qa[r] = xe->dout[rr * xe->ld_dout + c];
        qb[r] = with_diag ? xe->du_in[rr * xe->ld_du_in + c] : 0.f;
        if (xe->prev_u) {
          qc[r] = xe->x0[rr * xe->ld_x0 + c];
          qd[r] = xe->prev_u[rr * xe->ld_prev_u + c];
          qe[r] = with_diag ? xe->xl[rr * xe->ld_xl + c] : 0.f;
          qf[r] = xe->accumulate_dx0 ? xe->dx0[rr * xe->ld_dx0 + c] : 0.f;
        }
      }
    }
  }
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

  if (DIN || (a_vec && b_vec)) {  // (the DIN variants are launched on aligned operands only: host check)
    f32x4v ra0[2], rb0[2], ra1[2], rb1[2];
    f32x4v rq0[2], rq1[2];  // (DIN == 1: the q pieces of the generated A units; ra holds the h pieces until staging)
    DinPre dpre;
    if (DIN == 1) din_prepare<A_KC>(*dg, m0, g.M, tid, dpre);
    // k-tile indices past the end are clamped to the last tile: the loop body has the same loads every
    // iteration (the compiler can then wait for exactly the older register set), the duplicate tile is never used
    auto fetch = [&](f32x4v (&ra)[2], f32x4v (&rb)[2], f32x4v (&rq)[2], int t) {
      const int k0 = kbeg + (t < T ? t : T - 1) * BK32;
      if (DIN == 1) fetch_din<A_KC>(*dg, dpre, k0, kend, g.K, ra, rq);
      else fetch_tile<A_KC>(g.A, g.lda, m0, g.M, k0, kend, g.K, tid, ra);
      if (DIN == 2) fetch_tile_din_perm(g.B, g.ldb, tx, dg->E, k0, kend, g.K, tid, rb);
      else if (DIN == 3 && !b_vec) fetch_tile_generic<B_KC>(g.B, g.ldb, n0, g.N, k0, kend, tid, rb);  // (e.g. a [K, 1] weight)
      else fetch_tile<B_KC>(g.B, g.ldb, n0, g.N, k0, kend, g.K, tid, rb);
    };
    auto stage = [&](int buf, f32x4v (&ra)[2], const f32x4v (&rb)[2], const f32x4v (&rq)[2], int t) {
      const int k0 = kbeg + t * BK32;  // unclamped: a tile past the end is masked to zero
      const bool interior = rows_full && (k0 + BK32 <= kend);
      if (DIN == 1) combine_din<A_KC>(*dg, dpre, kbeg + (t < T ? t : T - 1) * BK32, g.K, ra, rq);
      if (DIN == 3) bna_apply(*ba, coef, tid, m0, g.M, kbeg + (t < T ? t : T - 1) * BK32, g.K, tx == 0 && t < T && ba->y != nullptr, ra);
      stage_tile<A_KC>(lds + buf * 2 * kOpTile, tid, ra, interior, m0, g.M, k0, kend);
      stage_tile<B_KC>(lds + buf * 2 * kOpTile + kOpTile, tid, rb, interior, n0, g.N, k0, kend);
    };
    // One step = one k-tile: read its fragments from LDS stage `buf`, issue the global loads of k-tile t + 2 into
    // the register set that was written to LDS at the end of the previous step, contract, and - after three
    // quarters of the MFMAs - write the OTHER register set (k-tile t + 1, loaded during the previous step) to the
    // other LDS stage.  A global load has almost two k-tiles of matrix work to arrive.  Two steps per loop
    // iteration so that each register set keeps its registers; an odd k-tile count is rounded up with an
    // all-zero tile (masked in stage_tile).
    // (FOUR register sets - the loads of k-tiles t + 1 .. t + 4 in flight, every load of a K <= 128 contraction issued
    // before its first MFMA - were built and measured in round 5: bit-identical, 96 - 135 VGPRs instead of 64 - 100, and
    // SLOWER on every config - DeepFM 0.3287 -> 0.3343 ms, DCN-v2 0.785 -> 0.802, DIN 1.833 -> 1.892, MMoE 2.094 -> 2.173:
    // global latency is not what the k loop waits for; profiles/r05_s15_gemm_prefetch_depth_ab_lines.txt)
    // (the PANEL form of short contractions - K <= 256: B's whole panel staged once, A's fragments straight from memory to
    // registers, one barrier, every load in flight before the first MFMA - was built and measured in round 6: bit-identical,
    // 130 - 260 VGPRs and 37 - 74 KB of LDS, and SLOWER: DeepFM 0.2984 -> 0.3195 ms (forward 69.5 -> 77.9 us, BatchNorm-backward
    // dgrads 40.8 -> 50.8), DCN-v2 0.709 -> 0.717, DIN 1.719 -> 1.726, MMoE level; profiles/r06_s23_panel_gemm_ab_lines.txt)
    // (EIGHT-wave workgroups whose second four waves contract the second half of the k-tiles of the same tile, the halves
    // added through LDS - two waves per SIMD for launches of at most one workgroup per compute unit - were built and
    // measured in round 6: 624 -> 256 at B = 4096 20.4 -> 19.8 us, but 256 -> 128 10.8 -> 11.1, 128 -> 64 7.9 -> 8.5 and
    // the input-gradient launches 7.7 -> 9.6 / 8.4 -> 9.6: DeepFM 0.3015 -> 0.3072 ms.  These launches are ~5 us of
    // prologue + epilogue + drain and 0.73 us per k-tile; halving the k-tiles per wave buys less than the second barrier
    // population and the LDS add cost.  profiles/r06_s13_gemm_ksplit_pair_rejected_*.txt)
    // Fragments are read a quarter of the k-tile at a time, right before their four MFMAs, and the compiler places the
    // instructions (no scheduling fences): against "every fragment first, fences around the MFMA groups" the bare core
    // (tools/micro/gemm_core.hip, variants 9 -> 1) gains 10 % on 8192 x 1152 x 256, 7 % on 8192 x 256 x 1152.
    auto step = [&](int buf, f32x4v (&fa_)[2], f32x4v (&fb_)[2], f32x4v (&fq_)[2], f32x4v (&sa)[2], f32x4v (&sb)[2],
                    f32x4v (&sq)[2], int t) {
      const float* base = lds + buf * 2 * kOpTile;
      fetch(fa_, fb_, fq_, t + 2);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const f32x4v a = *reinterpret_cast<const f32x4v*>(base + fa + 4 * q);
        const f32x4v b = *reinterpret_cast<const f32x4v*>(base + fb + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
      }
      stage(buf ^ 1, sa, sb, sq, t + 1);
      {
        const f32x4v a = *reinterpret_cast<const f32x4v*>(base + fa + 12);
        const f32x4v b = *reinterpret_cast<const f32x4v*>(base + fb + 12);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
      }
      __syncthreads();
    };
    fetch(ra0, rb0, rq0, 0);
    fetch(ra1, rb1, rq1, 1);
    stage(0, ra0, rb0, rq0, 0);
    __syncthreads();
    for (int t = 0; t < T; t += 2) {
      step(0, ra0, rb0, rq0, ra1, rb1, rq1, t);
      step(1, ra1, rb1, rq1, ra0, rb0, rq0, t + 1);
    }
  } else {
    // unaligned operands (e.g. lda = 81): masked scalar loads, one k-tile at a time
    f32x4v ra[2], rb[2];
    for (int t = 0; t < T; ++t) {
      const int k0 = kbeg + t * BK32;
      fetch_tile_generic<A_KC>(g.A, g.lda, m0, g.M, k0, kend, tid, ra);
      fetch_tile_generic<B_KC>(g.B, g.ldb, n0, g.N, k0, kend, tid, rb);
      stage_tile<A_KC>(lds, tid, ra, true, m0, g.M, k0, kend);
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
                   g.col_stats + static_cast<int64_t>(ty) * g.N * 3);
  if (BN_EPI && g.bn.partial != nullptr) {
    tile_bn_bwd_partial(acc, g.bn, py, pz, m0 + wm * 32, g.M, col < g.N ? col - g.bn.col0 : -1, g.bn.n_src, wm, wn, lane, lds, ty);
    if (XEPI == kEpiFrozenDz && g.bn.dz_out) {  // (uniform per problem; host: col0 == 0, n_src == N, one k-split, no accumulate)
      if (col >= g.N) return;
      const float ga = g.bn.gamma ? g.bn.gamma[col] : 1.f;
      const float is = g.bn.invstd[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (row < g.M) {
          float gg = acc[r] + bv;
          if (g.bn.act == ER_ACT_RELU && !(py[r] > 0.f)) gg = 0.f;
          g.C[static_cast<int64_t>(row) * g.ldc + col] = ga * is * gg;  // (bn_bwd_finalize_apply_body's frozen form)
        }
      }
      return;
    }
  }
  if (DIN == 2) {
    // dcat tile (64 rows x [seg0 | seg1 | seg2 | seg3] of 16 embedding positions) -> LDS -> dh and the dq partials
    const DinGen& d = *dg;
    constexpr int kD = 65;
    float* D = lds;                 // [64][65]
    float* Es = lds + 64 * kD;      // [64][16]
    __syncthreads();                // LDS operand tiles are dead
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      D[rl * kD + wn * 32 + (lane & 31)] = acc[r];
    }
    __syncthreads();
    const int j = tid & 15, rg = tid >> 4;
    const int J = tx * 16 + j;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = rg * 4 + i;
      const int row = m0 + rl;
      float e = 0.f;
      if (row < g.M) {
        const int b = din_div_L(d, row);
        const float qv = d.q[static_cast<int64_t>(b) * d.ldq + J];
        const float hv = d.h[static_cast<int64_t>(row) * d.ldh + J];
        const float d0 = D[rl * kD + j], d1 = D[rl * kD + 16 + j], d2 = D[rl * kD + 32 + j], d3 = D[rl * kD + 48 + j];
        const float gh = (d1 - d2) + qv * d3;
        float* p = d.dh + static_cast<int64_t>(row) * d.lddh + J;
        *p = d.accumulate_dh ? *p + gh : gh;
        e = (d0 + d2) + hv * d3;
      }
      Es[rl * 16 + j] = e;
    }
    __syncthreads();
    const int b0 = din_div_L(d, m0);
    for (int u = tid; u < d.slots * 16; u += kBlock) {
      const int slot = u >> 4, jj = u & 15;
      const int b = b0 + slot;
      int lo = b * d.L, hi = lo + d.L;
      lo = lo > m0 ? lo : m0;
      hi = hi < m0 + BM ? hi : m0 + BM;
      hi = hi < g.M ? hi : g.M;
      float sum = 0.f;
      for (int row = lo; row < hi; ++row) sum = sum + Es[(row - m0) * 16 + jj];
      d.dq_partial[(static_cast<int64_t>(ty) * d.slots + slot) * d.E + tx * 16 + jj] = sum;
    }
    return;
  }
  if (XEPI == ER_EPI_CROSS_FWD) {
    // x_{l+1} = x0 * (acc + b + diag * x_l) + x_l, in cross_v2_fwd_kernel's order; u keeps acc
    if (col >= g.N) return;
    const bool with_diag = xe->diag != 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (row < g.M) {
        const int64_t rr = row;
        if (xe->u) xe->u[rr * xe->ld_u + col] = acc[r];
        float t = acc[r] + bv;
        if (with_diag) t = t + xe->diag * qb[r];
        g.C[rr * g.ldc + col] = qa[r] * t + qb[r];
      }
    }
    return;
  }
  if (XEPI == ER_EPI_CROSS_BWD) {
    // v = acc + dout + diag * du_in = the whole gradient of x_{l-1}; C (+)= v; the lower cross layer's elementwise backward
    const bool with_diag = xe->diag != 0.f;
    const bool prev = xe->prev_u != nullptr;
    const float pb = (prev && xe->prev_bias && col < g.N) ? xe->prev_bias[col] : 0.f;
    float cs = 0.f;
    if (col < g.N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (row < g.M) {
          const int64_t rr = row;
          float v = acc[r] + bv + qa[r];
          if (with_diag) v = v + xe->diag * qb[r];
          float* p = g.C + rr * g.ldc + col;
          *p = g.accumulate ? *p + v : v;
          if (prev) {
            const float du = v * qc[r];
            float t = qd[r] + pb;
            if (with_diag) t = t + xe->diag * qe[r];
            const float a = v * t;
            xe->dx0[rr * xe->ld_dx0 + col] = xe->accumulate_dx0 ? qf[r] + a : a;
            if (xe->du_out) xe->du_out[rr * xe->ld_du_out + col] = du;
            if (xe->du_out_bf16) xe->du_out_bf16[rr * xe->ld_du_out_bf16 + col] = static_cast<uint16_t>(f32_to_bf16_rne(du));
            cs = cs + du;
          }
        }
      }
    }
    if (prev && xe->partial) {  // (uniform) per-tile column sums of du_out: lane pair, then the two waves sharing the columns
      const float o = __shfl_xor(cs, 32, 64);
      const float a = (khalf ? o : cs) + (khalf ? cs : o);
      __syncthreads();  // LDS operand tiles are dead
      float* slot = lds + (wn * 32 + (lane & 31));
      if (wm == 1 && khalf == 0) slot[0] = a;
      __syncthreads();
      if (wm == 0 && khalf == 0 && col < g.N) xe->partial[static_cast<int64_t>(ty) * g.N + col] = a + slot[0];
    }
    return;
  }
  if (col >= g.N) return;
  if (XEPI == kEpiFrozenBn && g.fz_y != nullptr) {  // (a grouped launch may mix problems with and without it; host: splits == 1)
    const float fb = g.bn.zbias ? g.bn.zbias[col] : 0.f;
    const float mu = g.bn.mean[col];
    const float is = 1.f / sqrtf(g.bn.invstd[col] + g.fz_eps);
    const float ga = g.bn.z ? g.bn.z[col] : 1.f, be = g.bn.y ? g.bn.y[col] : 0.f;
    if (ty == 0 && wm == 0 && khalf == 0) {
      g.fz_save[col] = mu;
      g.fz_save[g.N + col] = is;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (row < g.M) {
        const float zv = acc[r] + bv;
        g.C[static_cast<int64_t>(row) * g.ldc + col] = zv;
        float v = ((zv + fb) - mu) * is;
        v = v * ga + be;
        if (g.bn.act == ER_ACT_RELU) v = v > 0.f ? v : 0.f;
        g.fz_y[static_cast<int64_t>(row) * g.ldc + col] = v;
      }
    }
    return;
  }
  float* Cz = g.C + (g.splits > 1 ? static_cast<int64_t>(bz) * g.M * g.N : 0);
  const int ldc = g.splits > 1 ? g.N : g.ldc;
  if (g.accumulate && g.splits == 1) {
    // C +=: the lane's 16 old values are requested together from clamped addresses, then added and stored (a load inside
    // the per-row branch is waited for there, and the store before it too: 16 read-modify-write round trips in a row)
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      row = row < g.M ? row : g.M - 1;
      old[r] = Cz[static_cast<int64_t>(row) * ldc + col];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (row < g.M) Cz[static_cast<int64_t>(row) * ldc + col] = old[r] + (acc[r] + bv);
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    if (row < g.M) Cz[static_cast<int64_t>(row) * ldc + col] = acc[r] + bv;
  }
}

// Grouped launch: up to kMaxGroup independent problems of one layout in ONE grid (the weight gradients of all
// layers of a step: each is a small M x N with K = batch, far too few tiles to fill 256 CUs on its own).
//
// Two regions.  XCD region, workgroups [0, 8 * xstart[n]): workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md: observed
// dispatch order, used for speed only - any placement gives the same result) and every XCD has its own L2.  The tiles
// of ONE k-split all read the same rows of both operands, so splits are dealt to the XCDs whole; inside its share
// j = b / 8 of the region, XCD x runs problem p's workgroups [xstart[p], xstart[p + 1]):
//   xper[p] == 1 (splits >= 8): the first xsplits[p] = 8 * (splits / 8) splits - XCD x takes splits x, x + 8, ..., all
//     their tiles, split-major;
//   xper[p] == 2 | 4 (splits == 4 | 2): split x / xper on xper neighbouring XCDs, tile slot * xper + x % xper (a slot past
//     the last tile idles: fewer than 8 workgroups per problem, not the surplus that aliased with the CU round robin).
// With 8 (4) splits per problem - DeepFM's weight gradients at B = 4096, stand-alone (in the step's tail) - an XCD reads
// one eighth (quarter) of the batch rows of every operand and nothing else: counter traffic 121 MB -> 47.5 MB per launch
// = 1.0x operands + workspace (profiles/r05_s10_pmc_wgrad_by_xcd.txt).  The launch is NOT faster for it (39 us either
// way: it never waited for HBM), and batch-long contractions (DIN's B x L = 204,800 rows, ~100 splits of 1 - 4 tiles)
// got SLOWER (190 -> 245 us: every L2 holding a copy of the hot rows is 8x the L2 bandwidth of one), so the host places
// by XCD only the problems whose splits are sized by the workgroup target, not by the contraction's length (plan_grouped).
// Legacy region, after it: the remaining splits (xsplits[p] .. splits - 1) of problem p at [start[p], start[p + 1]):
// tile slot = local % tiles (XCD-aware tile order inside the problem, tile_coords), k-split = local / tiles.
constexpr int kMaxGroup = 16;
struct GroupedArgs {
  int n;
  int start[kMaxGroup + 1];
  int tiles[kMaxGroup];
  int xstart[kMaxGroup + 1];
  int xsplits[kMaxGroup];
  int xper[kMaxGroup];
  GemmArgs p[kMaxGroup];
};
struct GroupedCoords {
  int p, tile, split;  // split < 0: an idle workgroup
  bool plain;          // tile is the tile itself (XCD region), not a slot of tile_coords' order
};
__host__ __device__ inline GroupedCoords grouped_coords(const int* __restrict__ start, const int* __restrict__ tiles,
                                                        const int* __restrict__ xstart, const int* __restrict__ xsplits,
                                                        const int* __restrict__ xper, int n, int b) {
  const int r1 = 8 * xstart[n];
  if (b < r1) {
    const int x = b & 7, j = b >> 3;
    int p = 0;
    while (p + 1 < n && j >= xstart[p + 1]) ++p;
    const int l = j - xstart[p];
    const int xp = xper[p];
    if (xp == 1) return GroupedCoords{p, l % tiles[p], x + 8 * (l / tiles[p]), true};
    const int tile = l * xp + x % xp;
    return GroupedCoords{p, tile, tile < tiles[p] ? x / xp : -1, true};
  }
  b -= r1;
  int p = 0;
  while (p + 1 < n && b >= start[p + 1]) ++p;
  const int local = b - start[p];
  return GroupedCoords{p, local % tiles[p], xsplits[p] + local / tiles[p], false};
}
__device__ __forceinline__ GroupedCoords grouped_coords(const GroupedArgs& ga, int b) {
  return grouped_coords(ga.start, ga.tiles, ga.xstart, ga.xsplits, ga.xper, ga.n, b);
}

// C[i, j] (+)= bias[j] + sum_s ws[s, i, j]   (split order fixed: deterministic).  VEC = 4: float4 per lane, the
// `splits` loads of a lane are independent and unrolled by 16 (by 4, the 256 splits of DIN's first-layer weight gradient -
// 16 workgroups' worth of output - were 64 dependent trips per lane: 19.8 us for 16.7 MB).
template <int VEC>
__device__ __forceinline__ void splitk_reduce_elems(const float* __restrict__ ws, int64_t mn, int N, int splits,
                                                    const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                    int accumulate, int64_t i) {
  if (i >= mn) return;
  float s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
#pragma unroll 16
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
__device__ __forceinline__ void splitk_reduce_grouped_block(const GroupedReduceArgs& ra, int b) {
  int p = 0;
  while (p + 1 < ra.n && b >= ra.start[p + 1]) ++p;
  const ReduceItem& r = ra.r[p];
  const int64_t lane = static_cast<int64_t>(b - ra.start[p]) * kBlock + threadIdx.x;
  if (VEC == 4 && r.vec) splitk_reduce_elems<4>(r.ws, r.mn, r.N, r.splits, r.bias, r.C, r.ldc, r.accumulate, lane * 4);
  else splitk_reduce_elems<1>(r.ws, r.mn, r.N, r.splits, r.bias, r.C, r.ldc, r.accumulate, lane);
}

// what er_gemm_grouped_f32 launches for one group of problems (er_gemm.hip); er_emb_bwd_fused_wgrad (er_embedding.hip)
// launches the same records next to the embedding row update
struct GroupedPlan {
  GroupedArgs ga;
  GroupedReduceArgs ra;
  bool any_bn;
  bool any_fz;  // a problem with the frozen-BatchNorm forward epilogue (GemmArgs.fz_y)
  bool any_dz;  // a problem whose BatchNorm-backward epilogue also stores dz (BnBwdEpi.dz_out)
};
// the two regions of a grouped grid from the problems' tile and k-split counts (GroupedArgs); by_xcd[p] 0: problem p in
// the legacy region only
inline void grouped_layout(const int* tiles, const int* splits, int n, const int* by_xcd, int* start, int* xstart, int* xsplits,
                           int* xper) {
  start[0] = 0;
  xstart[0] = 0;
  for (int p = 0; p < n; ++p) {
    int xs = 0, xp = 0, per_xcd = 0;
    if (by_xcd && by_xcd[p]) {
      if (splits[p] >= 8) {
        xs = (splits[p] / 8) * 8; xp = 1; per_xcd = tiles[p] * (xs / 8);
      } else if (splits[p] == 4 || splits[p] == 2) {
        xs = splits[p]; xp = 8 / splits[p]; per_xcd = (tiles[p] + xp - 1) / xp;
      }
    }
    xsplits[p] = xs;
    xper[p] = xp;
    xstart[p + 1] = xstart[p] + per_xcd;
    start[p + 1] = start[p] + tiles[p] * (splits[p] - xs);
  }
}
inline int grouped_grid(const int* start, const int* xstart, int n) { return 8 * xstart[n] + start[n]; }
inline int grouped_grid(const GroupedArgs& ga) { return grouped_grid(ga.start, ga.xstart, ga.n); }
// target_override > 0: the number of workgroups the k-splits aim at instead of the launch's own default (512)
int plan_grouped(int layout, const er_gemm_problem* pr, int n, bool bf16, GroupedPlan* plan, int64_t target_override = 0);
int launch_grouped_reduce(const GroupedReduceArgs& ra, hipStream_t s);

}  // namespace er
