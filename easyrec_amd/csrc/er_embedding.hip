// K2/K3/K4: embedding lookup+combine forward, de-duplicated backward, row-wise optimizers.
//
// Replaces (reference, TF-op chains issued from Python):
//   forward : feature_column_v2.py:3434-3462 (safe_embedding_lookup_sparse), compat/feature_column/
//             feature_column.py:384-414 (reshape+concat), layers/input_layer.py:369-375 (L2 on outputs)
//   backward: IndexedSlices gradient of the same chain + optimizer sparse apply
//             (tf.train.AdamOptimizer._apply_sparse, compat/adam_s.py:185-213).
//
// MI355X design notes
//   * all lookups of a model run in ONE launch: a block handles 256/G output rows of one lookup,
//     G = lanes per row (dim/4 rounded up to a power of two, 128-bit loads: a D=16 row is one 64 B
//     segment read by 4 lanes, 16 rows per wave instruction).  Outputs are written straight into
//     the concatenated [B, sum(dim)] buffer - no concat pass.
//   * backward is sort-based and deterministic: 32-bit keys (row inside the table group), stable
//     radix sort (rocPRIM), then a two-level in-order segmented reduction; the head lane-group of
//     every run applies the optimizer, so each touched row's var/m/v is read and written once.
//   * TF's dense-decay Adam additionally streams every untouched row once per step
//     (adam_decay_sweep_kernel): pure HBM streaming, float4, grid-stride, bitmap-skipped.
//   * everything here is HBM/latency-bound integer+fp32 work; no MFMA on purpose.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "er_common.h"
#include "er_gemm_core.h"
#include "er_dense_tail.h"
#include "er_decay.h"
#include "er_grad_finish.h"

namespace er {

constexpr uint32_t kInvalidKey = 0xFFFFFFFFu;

__host__ __device__ __forceinline__ void lane_geom(int dim, int& V, int& G) {
  V = (dim % 4 == 0) ? 4 : 1;
  const int n = dim / V;
  G = 1;
  while (G < n) G <<= 1;
}

__device__ __forceinline__ int find_lookup(const int32_t* __restrict__ blk_start, int n, int bid) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blk_start[mid] <= bid) lo = mid; else hi = mid;
  }
  return lo;
}

struct Acc4 {
  float x, y, z, w;
};

// ------------------------------------------------------------------------------------------------
// backward: entries, sort, reduce, apply
// ------------------------------------------------------------------------------------------------
// One thread per output row of one lookup: emits (key, grad pointer, scale) for every id of the row.
// Row sharding of embedding-parallel training (reference compat/feature_column/feature_column.py:296,317,
// 461-463): owner = id mod world, local row = id div world.  The routed key of an entry is
// owner * shard_stride + local_base[lookup] + id div world, so that one ascending sort groups the
// entries by owner and, inside an owner, by the row of the owner's shard storage.  world == 1 gives
// key = key_base + id (local_base == key_base): the single-GPU layout.
struct Route {
  int32_t world;
  int64_t shard_stride;
  const int64_t* local_base;  // [n_lookups] (device) or nullptr when world == 1
};

__device__ __forceinline__ uint32_t routed_key(const er_lookup_desc& d, const Route& rt, int l, int64_t id) {
  if (rt.local_base == nullptr) return static_cast<uint32_t>(d.key_base + id);
  const int64_t owner = id % rt.world;
  return static_cast<uint32_t>(owner * rt.shard_stride + rt.local_base[l] + id / rt.world);
}

__device__ __forceinline__ void build_body(int bid, const er_lookup_desc* __restrict__ descs,
                                           const int32_t* __restrict__ blk_start, const int64_t* __restrict__ ent_base,
                                           int n_lookups, const Route& rt, int64_t n_active, uint32_t* __restrict__ keys,
                                           uint32_t* __restrict__ vals, const float** __restrict__ ent_gptr,
                                           float* __restrict__ ent_scale) {
  const int l = find_lookup(blk_start, n_lookups, bid);
  const er_lookup_desc d = descs[l];
  const int r = (bid - blk_start[l]) * kBlock + static_cast<int>(threadIdx.x);
  if (r >= d.n_rows || r >= n_active) return;
  int64_t kb, ke;
  if (d.offsets) {
    kb = d.offsets[r];
    ke = d.offsets[r + 1];
  } else {
    kb = r;
    ke = r + 1;
  }
  const bool prune_nonpos = (d.weights != nullptr) && (d.combiner != ER_COMBINER_SUM);
  float den = 1.f;
  if (d.combiner != ER_COMBINER_SUM) {
    float wsum = 0.f, w2sum = 0.f;
    for (int64_t k = kb; k < ke; ++k) {
      const int64_t id = d.ids[k];
      if (id < 0 || id >= d.rows) continue;
      const float w = d.weights ? d.weights[k] : 1.f;
      if (prune_nonpos && !(w > 0.f)) continue;
      wsum = wsum + w;
      w2sum = w2sum + w * w;
    }
    den = (d.combiner == ER_COMBINER_MEAN) ? wsum : sqrtf(w2sum);
    if (wsum == 0.f) den = 1.f;
  }
  const float* gp = d.out + static_cast<int64_t>(r) * d.out_stride + d.out_col;
  const int64_t base = ent_base[l];
  for (int64_t k = kb; k < ke; ++k) {
    const int64_t j = base + k;
    const int64_t id = d.ids[k];
    const float w = d.weights ? d.weights[k] : 1.f;
    bool ok = !(id < 0 || id >= d.rows);
    if (prune_nonpos && !(w > 0.f)) ok = false;
    keys[j] = ok ? routed_key(d, rt, l, id) : kInvalidKey;
    vals[j] = static_cast<uint32_t>(j);
    ent_gptr[j] = gp;
    ent_scale[j] = w / den;
  }
}

__global__ void __launch_bounds__(kBlock)
emb_bwd_build_kernel(const er_lookup_desc* __restrict__ descs, const int32_t* __restrict__ blk_start,
                     const int64_t* __restrict__ ent_base, int n_lookups, Route rt, int64_t n_active,
                     uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                     const float** __restrict__ ent_gptr, float* __restrict__ ent_scale) {
  build_body(blockIdx.x, descs, blk_start, ent_base, n_lookups, rt, n_active, keys, vals, ent_gptr, ent_scale);
}

struct BuildArgs {
  const er_lookup_desc* descs;
  const int32_t* blk_start;
  const int64_t* ent_base;
  int n_lookups;
  Route rt;
  int64_t n_active;
  uint32_t* keys;
  uint32_t* vals;
  const float** ent_gptr;
  float* ent_scale;
};
struct BuildMulti {
  int n;
  int start[4 + 1];
  BuildArgs a[4];
};

__global__ void __launch_bounds__(kBlock)
emb_bwd_build_multi_kernel(BuildMulti ma) {
  int i = 0;
  while (i + 1 < ma.n && static_cast<int>(blockIdx.x) >= ma.start[i + 1]) ++i;
  const BuildArgs& a = ma.a[i];
  build_body(blockIdx.x - ma.start[i], a.descs, a.blk_start, a.ent_base, a.n_lookups, a.rt, a.n_active, a.keys, a.vals,
             a.ent_gptr, a.ent_scale);
}

// Marks the rows an upcoming step will touch (same validity rule as emb_bwd_build_kernel), so that the
// dense-decay sweep over the UNTOUCHED rows can run concurrently with forward/backward on another
// stream: untouched rows are neither read by this step's lookup nor written by its row updates.
__global__ void __launch_bounds__(kBlock)
emb_mark_kernel(const er_lookup_desc* __restrict__ descs, const int32_t* __restrict__ blk_start, int n_lookups,
                int64_t n_active, uint32_t* __restrict__ bitmap) {
  const int l = find_lookup(blk_start, n_lookups, blockIdx.x);
  const er_lookup_desc d = descs[l];
  const int r = (blockIdx.x - blk_start[l]) * kBlock + static_cast<int>(threadIdx.x);
  if (r >= d.n_rows || r >= n_active) return;
  int64_t kb, ke;
  if (d.offsets) {
    kb = d.offsets[r];
    ke = d.offsets[r + 1];
  } else {
    kb = r;
    ke = r + 1;
  }
  const bool prune_nonpos = (d.weights != nullptr) && (d.combiner != ER_COMBINER_SUM);
  for (int64_t k = kb; k < ke; ++k) {
    const int64_t id = d.ids[k];
    if (id < 0 || id >= d.rows) continue;
    if (prune_nonpos && !(d.weights[k] > 0.f)) continue;
    const uint32_t key = static_cast<uint32_t>(d.key_base + id);
    atomicOr(&bitmap[key >> 5], 1u << (key & 31));
  }
}

template <int V>
struct Vec;
template <>
struct Vec<4> {
  float4 v;
  __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void loadu(const float* p) { v = make_float4(p[0], p[1], p[2], p[3]); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
  __device__ __forceinline__ void add_scaled(const Vec& o, float s) {
    v.x = v.x + o.v.x * s; v.y = v.y + o.v.y * s; v.z = v.z + o.v.z * s; v.w = v.w + o.v.w * s;
  }
  __device__ __forceinline__ void add(const Vec& o) {
    v.x = v.x + o.v.x; v.y = v.y + o.v.y; v.z = v.z + o.v.z; v.w = v.w + o.v.w;
  }
};
template <>
struct Vec<1> {
  float v;
  __device__ __forceinline__ void zero() { v = 0.f; }
  __device__ __forceinline__ void load(const float* p) { v = p[0]; }
  __device__ __forceinline__ void loadu(const float* p) { v = p[0]; }
  __device__ __forceinline__ void store(float* p) const { p[0] = v; }
  __device__ __forceinline__ void add_scaled(const Vec& o, float s) { v = v + o.v * s; }
  __device__ __forceinline__ void add(const Vec& o) { v = v + o.v; }
};

template <int V>
__device__ __forceinline__ void gather_grad(Vec<V>& acc, const float* gp, int c, float scale) {
  Vec<V> g;
  if (V == 4 && (reinterpret_cast<uintptr_t>(gp + c) & 15) == 0) g.load(gp + c); else g.loadu(gp + c);
  acc.add_scaled(g, scale);
}

struct RowUpdate {
  float* var;
  float* m;
  float* v;
  uint32_t* bitmap;
  // lazy dense decay (ER_OPT_ADAM without the sweep): last step whose update each row has received, and the
  // device step counter (== index of the current step + 1 once er_hyper_select has run)
  int32_t* last_step;
  const int64_t* step_counter;
  // Row pitch (er_emb_group_set_row_pitch).  ld: floats between consecutive rows of var / m / v - dim for three plain
  // [rows, dim] arrays; larger when a row's var | m | v (| last_step) lie side by side in ONE record, so that a
  // touched row is one contiguous HBM access instead of three or four scattered ones.  ls_ld: int32 words between
  // consecutive rows' last_step (1: an array of its own).
  int64_t ld;
  int64_t ls_ld;
  __device__ __forceinline__ int64_t off(int64_t key, int c) const { return key * ld + c; }
  __device__ __forceinline__ int32_t& ls(int64_t key) const { return last_step[key * ls_ld]; }
};

__device__ __forceinline__ float adam_elem(float& m, float& v, float var, float g, const er_opt_hyper& h) {
  // tf.train.AdamOptimizer._apply_sparse_shared / compat/adam_s.py:193-213 (same row arithmetic):
  //   m_t = m*beta1 + g*(1-beta1); v_t = v*beta2 + (g*g)*(1-beta2); var -= lr_t*m_t/(sqrt(v_t)+eps)
  const float mt = m * h.beta1 + g * h.one_minus_beta1;
  const float vt = v * h.beta2 + (g * g) * h.one_minus_beta2;
  m = mt;
  v = vt;
  return var - (h.lr_t * mt) / (sqrtf(vt) + h.eps);
}

template <int V>
__device__ __forceinline__ void ld_vec(float (&r)[V], const float* p) {
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  } else {
    r[0] = p[0];
  }
}
template <int V>
__device__ __forceinline__ void st_vec(float* p, const float (&r)[V]) {
  if constexpr (V == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
    p[0] = r[0];
  }
}

template <int V>
__device__ __forceinline__ void update_row(const RowUpdate& t, int opt_kind, const er_opt_hyper& h, int64_t off,
                                           const float* g) {
  float var[V], m[V], v[V];
  ld_vec<V>(var, t.var + off);
  if (opt_kind == ER_OPT_ADAM || opt_kind == ER_OPT_LAZY_ADAM) {
    ld_vec<V>(m, t.m + off);
    ld_vec<V>(v, t.v + off);
#pragma unroll
    for (int i = 0; i < V; ++i) var[i] = adam_elem(m[i], v[i], var[i], g[i], h);
    st_vec<V>(t.m + off, m);
    st_vec<V>(t.v + off, v);
  } else if (opt_kind == ER_OPT_ADAGRAD) {
    ld_vec<V>(v, t.v + off);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      v[i] = v[i] + g[i] * g[i];
      var[i] = var[i] - (g[i] * h.lr) / sqrtf(v[i]);
    }
    st_vec<V>(t.v + off, v);
  } else {  // SGD
#pragma unroll
    for (int i = 0; i < V; ++i) var[i] = var[i] - h.lr * g[i];
  }
  st_vec<V>(t.var + off, var);
}

// ------------------------------------------------------------------------------------------------
// TF-exact Adam WITHOUT the dense sweep ("lazy dense decay").  tf.train.AdamOptimizer._apply_sparse decays m, v
// and moves var of EVERY row at EVERY step; a row that no lookup reads between two of its touches cannot
// influence anything in between, so its decay-only steps can be replayed the next time it is touched - the same
// fp32 operations in the same order (m*=b1; v*=b2; var -= lr_t(s)*m/(sqrt(v)+eps) for each missed step s, with
// lr_t(s) read from the per-step history written by er_hyper_select), i.e. bit-identical to the sweep.  After
// about 900 idle steps m has settled on a denormal fixed point of fl(m*b1) (|m| <= 5.6e-45) whose update var
// absorbs: var and m no longer change, and the remaining decay of v is applied in closed form v *= b2^k (the
// only deviation from step-by-step rounding: <= 1e-6 relative on v, on rows idle that long).
// The rows of a step are caught up BEFORE the step's lookup reads them; er_emb_flush_decay brings every row
// current (checkpoint / state_dict / evaluation).  HBM traffic per step drops from 24 B x every table element
// to the touched rows.
// ------------------------------------------------------------------------------------------------
constexpr int32_t kClosedFormMin = 2048;
constexpr int kLrStage = 512;  // the last kLrStage entries of the lr_t history are staged in LDS by replay_block

// The scalars of the replay, pinned in SGPRs (readfirstlane: the compiler otherwise re-loads the fields of the
// er_opt_hyper record from memory inside the per-step loop, a dependent scalar load per iteration).
struct DecayConsts {
  float b1, b2, eps;
  bool can_absorb;  // b1 < sqrt(b2): the update of a decay-only row shrinks from step to step
};
__device__ __forceinline__ float pin_scalar(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
__device__ __forceinline__ DecayConsts pin_decay_consts(const er_opt_hyper& h) {
  DecayConsts k;
  k.b1 = pin_scalar(h.beta1);
  k.b2 = pin_scalar(h.beta2);
  k.eps = pin_scalar(h.eps);
  k.can_absorb = k.b1 < 0.999f * sqrtf(k.b2);
  return k;
}

// lr_t(s): the newest kLrStage steps from LDS (staged once per workgroup), anything older from HBM.  The replay's
// per-step loop is a chain of dependent operations; a global load per iteration (per-lane address: the lanes of a
// wave replay different steps) made it latency-bound.
struct LrHist {
  const float* __restrict__ global;
  const float* staged;  // [kLrStage] = global[base .. base + kLrStage), or nullptr
  int32_t base;
  __device__ __forceinline__ float at(int32_t s) const {
    return (staged != nullptr && s >= base) ? staged[s - base] : global[s];
  }
};

// What a replay needs besides the lr_t history.  A == nullptr: the exact step-by-step replay (lr_max: prefix maxima of
// the history, enables its absorbed regime); A != nullptr: the closed form of er_decay.h.
struct DecayAux {
  const float* lr_max = nullptr;
  const float* A = nullptr;  // [K][kDecayLd] for the s_end of THIS launch
  const float* C = nullptr;  // [capacity + 1][kDecayLd]
  double ln_b1 = 0.0, ln_b2 = 0.0;
  int K = 0;
};

// steps s_begin .. s_end-1 were decay-only for this row (s_begin < s_end): closed form, er_decay.h
template <int V>
__device__ __forceinline__ void replay_closed(float* var, float* m, float* v, const DecayAux& x, int32_t s_begin,
                                              int32_t s_end, float eps) {
  const int32_t k = s_end - s_begin;
  const float* __restrict__ T = k <= x.K ? x.A + static_cast<int64_t>(k - 1) * kDecayLd
                                         : x.C + static_cast<int64_t>(s_begin) * kDecayLd;  // (t0 + 1 = s_begin)
  const float4 ta = *reinterpret_cast<const float4*>(T);
  const float2 tb = *reinterpret_cast<const float2*>(T + 4);
  const float p1 = static_cast<float>(exp(x.ln_b1 * static_cast<double>(k)));
  const float p2 = static_cast<float>(exp(x.ln_b2 * static_cast<double>(k)));
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const float a = sqrtf(v[i]);
    const float d = a + eps;
    const float z = a / d;
    float poly = tb.y;
    poly = poly * z + tb.x;
    poly = poly * z + ta.w;
    poly = poly * z + ta.z;
    poly = poly * z + ta.y;
    poly = poly * z + ta.x;
    var[i] = var[i] - (m[i] * poly) / d;
    m[i] = m[i] * p1;
    v[i] = v[i] * p2;
  }
}

// k more decay-only steps of a row whose m no longer moves and whose update var absorbs: v *= b2, k times.
template <int V>
__device__ __forceinline__ void decay_v_only(float* v, int32_t k, float b2) {
  if (k > kClosedFormMin) {
    const double f = pow(static_cast<double>(b2), static_cast<double>(k));
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = static_cast<float>(static_cast<double>(v[i]) * f);
    return;
  }
  for (int32_t j = 0; j < k; ++j) {
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = v[i] * b2;
  }
}

template <int V>
__device__ __forceinline__ void replay_decay(float* var, float* m, float* v, const LrHist& lr, int32_t s_begin,
                                             int32_t s_end, const DecayConsts& k, float lr_cap2,
                                             bool* cheap_from_here = nullptr) {
  // steps s_begin .. s_end - 1 were decay-only for this row.  cheap_from_here (optional): did the row leave the full
  // regime by the end of this call (replay_block sorts its tasks by it).  lr_cap2 = 2 * max lr_t over the history so
  // far (0: no absorbed regime).  Three regimes, the first two bit-identical to the step-by-step sweep:
  //  (1) full step: m *= b1; v *= b2; var -= lr_t(s) * m / (sqrt(v) + eps)      (~45 instructions/element: IEEE sqrt, division)
  //  (2) ABSORBED: the update has fallen below a quarter ulp of var and can only shrink from here, so var no longer
  //      changes and only the two decays remain (2 multiplications/element/step).  Why it can only shrink: in exact
  //      arithmetic q(s) = lr_t(s) |m_s| / (sqrt(v_s) + eps) obeys q(s+1) / q(s) <= (lr_t(s+1) / lr_t(s)) * b1 / sqrt(b2)
  //      (+ rounding of 2^-23 per step); with L = max lr_t over the whole history so far (written next to the history
  //      by er_hyper_select) every later update is <= (L / lr_t(s)) * q(s) as long as b1 < sqrt(b2).  The test is on
  //      the COMPUTED update with a 2x margin and without a division:
  //      |upd| * 2 L < 2^-26 |var| * lr_t(s)  =>  every later |upd| < 2^-26 |var| <= ulp(var) / 4, which also covers
  //      var at a power of two (half-sized ulp below).  Typically reached ~130 steps after a row's last touch.
  //  (3) settled: fp32 m never reaches 0 under m *= b1 - below 5 denormal units fl(m * 0.9) == m - so ~900 steps
  //      after the touch m is constant and only v *= b2 is left (found by regime 2: m stops changing): replayed step
  //      by step (1 multiplication/element/step, still bit-identical) as long as at most kClosedFormMin steps are
  //      pending - always, when the rolling flush (er_emb_flush_window) bounds the idle time; a longer backlog (a
  //      flush after thousands of steps without the rolling flush) takes the closed form v *= b2^k, the one
  //      documented deviation (<= 1e-6 relative on v).
  //  A row whose var is 0 (or so small that no update is ever absorbed) stays in regime 1: correct, merely slower.
  const bool can_absorb = k.can_absorb && lr_cap2 > 0.f;
  int32_t s = s_begin;
  bool absorbed = false;
  for (; s < s_end && !absorbed; ++s) {
    const float lr_t = lr.at(s);
    const float thr = 1.4901161193847656e-08f * lr_t;  // 2^-26 lr_t  (lr_t > 0: the schedules have a positive floor)
    bool all_small = can_absorb;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float mt = m[i] * k.b1;
      const float vt = v[i] * k.b2;
      const float upd = (lr_t * mt) / (sqrtf(vt) + k.eps);
      m[i] = mt;
      v[i] = vt;
      var[i] = var[i] - upd;
      all_small = all_small && (fabsf(upd) * lr_cap2 < fabsf(var[i]) * thr);
    }
    absorbed = all_small;
  }
  if (cheap_from_here) *cheap_from_here = absorbed;
  // (2) var is fixed from here on (every lane element absorbed; lanes of a row decide independently - each owns its
  // elements).  m and v keep their step-by-step rounding.
  for (; s < s_end; ++s) {
    bool m_fixed = true;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float mt = m[i] * k.b1;
      m_fixed = m_fixed && (mt == m[i]);
      m[i] = mt;
      v[i] = v[i] * k.b2;
    }
    if (m_fixed && s + 1 < s_end) {  // m sits on its fixed point: only v is left (regime 3)
      decay_v_only<V>(v, s_end - s - 1, k.b2);
      return;
    }
  }
}

// 2 * (largest lr_t of steps 0 .. s_end-1), or 0 when the running maxima are not kept (no absorbed regime)
__device__ __forceinline__ float lr_cap_twice(const float* __restrict__ lr_max, int32_t s_end) {
  return (lr_max != nullptr && s_end > 0) ? 2.0f * lr_max[s_end - 1] : 0.f;
}

// Block-cooperative replay.  A lane's task = the V elements it owns of one row with pending decay steps
// [s_begin, s_end) (s_end uniform over the workgroup).  The cost of a task is dominated by its steps in the FULL regime
// (sqrt + IEEE division per element and step) - ~130 of them after a touch, none for a row that has long been idle -
// and rows of both kinds sit side by side in a table: lanes replaying in place would idle while one lane of their wave
// grinds through full steps (measured: two thirds of the lane-cycles).  So: every task runs ONE step in place, which
// also classifies it (still full / cheap from here); unfinished tasks are compacted into two LDS queues - registers
// and all - and the workgroup's lanes take them densely, the full ones first.  Results do not depend on the order
// the queues fill in.  All threads of the workgroup must call this (it synchronises); smem: kReplaySmemWords words.
constexpr int kReplayEntry = 3 * 4 + 3;                        // var, m, v (V <= 4) + element offset (2) + next step
constexpr int kReplaySmemWords = 2 * kBlock * kReplayEntry + 4 + kLrStage;

template <int V>
__device__ __forceinline__ void replay_block(bool has, int64_t off, int32_t s_begin, int32_t s_end, const RowUpdate& tab,
                                             const float* __restrict__ lr_hist, const DecayAux& aux,
                                             const er_opt_hyper* __restrict__ hyper, uint32_t* __restrict__ smem) {
  if (aux.A != nullptr) {  // closed form: a fixed cost per element whatever the backlog - nothing to balance
    __syncthreads();       // (every lane has read last_step: the lanes of a row may sit in two wavefronts)
    if (has && s_begin < s_end) {
      float var[V], m[V], v[V];
      ld_vec<V>(m, tab.m + off);
      ld_vec<V>(v, tab.v + off);
      bool live = false;
#pragma unroll
      for (int j = 0; j < V; ++j) live = live || (m[j] != 0.f) || (v[j] != 0.f);
      if (live) {
        ld_vec<V>(var, tab.var + off);
        replay_closed<V>(var, m, v, aux, s_begin, s_end, pin_scalar(hyper->eps));
        st_vec<V>(tab.var + off, var);
        st_vec<V>(tab.m + off, m);
        st_vec<V>(tab.v + off, v);
      }
    }
    return;
  }
  int* cnt = reinterpret_cast<int*>(smem);  // [2]
  uint32_t* queue = smem + 4;               // [2][kBlock][kReplayEntry]
  float* lr_staged = reinterpret_cast<float*>(smem + 4 + 2 * kBlock * kReplayEntry);  // [kLrStage]
  if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
  const int32_t stage_base = s_end - kLrStage;
  for (int i = threadIdx.x; i < kLrStage; i += kBlock) {
    const int32_t si = stage_base + i;
    lr_staged[i] = si >= 0 ? lr_hist[si] : 0.f;
  }
  __syncthreads();
  const LrHist lr{lr_hist, lr_staged, stage_base};
  const DecayConsts k = pin_decay_consts(*hyper);
  const float lr_cap2 = lr_cap_twice(aux.lr_max, s_end);
  float var[V], m[V], v[V];
  bool task = false;
  int32_t s = s_begin;
  if (has && s_begin < s_end) {
    ld_vec<V>(m, tab.m + off);
    ld_vec<V>(v, tab.v + off);
#pragma unroll
    for (int j = 0; j < V; ++j) task = task || (m[j] != 0.f) || (v[j] != 0.f);  // never touched: a fixed point of the decay
  }
  if (task) {
    ld_vec<V>(var, tab.var + off);
    bool cheap = false;
    replay_decay<V>(var, m, v, lr, s, s + 1, k, lr_cap2, &cheap);
    ++s;
    if (s >= s_end) {
      st_vec<V>(tab.var + off, var);
      st_vec<V>(tab.m + off, m);
      st_vec<V>(tab.v + off, v);
    } else {
      const int q = cheap ? 1 : 0;
      uint32_t* e = queue + (static_cast<size_t>(q) * kBlock + atomicAdd(&cnt[q], 1)) * kReplayEntry;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        e[j] = __builtin_bit_cast(uint32_t, var[j]);
        e[4 + j] = __builtin_bit_cast(uint32_t, m[j]);
        e[8 + j] = __builtin_bit_cast(uint32_t, v[j]);
      }
      e[12] = static_cast<uint32_t>(static_cast<uint64_t>(off) & 0xFFFFFFFFu);
      e[13] = static_cast<uint32_t>(static_cast<uint64_t>(off) >> 32);
      e[14] = static_cast<uint32_t>(s);
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int q = 0; q < 2; ++q) {
    const int n = cnt[q];
    for (int t = threadIdx.x; t < n; t += kBlock) {
      const uint32_t* e = queue + (static_cast<size_t>(q) * kBlock + t) * kReplayEntry;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        var[j] = __builtin_bit_cast(float, e[j]);
        m[j] = __builtin_bit_cast(float, e[4 + j]);
        v[j] = __builtin_bit_cast(float, e[8 + j]);
      }
      const int64_t o = static_cast<int64_t>(static_cast<uint64_t>(e[12]) | (static_cast<uint64_t>(e[13]) << 32));
      replay_decay<V>(var, m, v, lr, static_cast<int32_t>(e[14]), s_end, k, lr_cap2);
      st_vec<V>(tab.var + o, var);
      st_vec<V>(tab.m + o, m);
      st_vec<V>(tab.v + o, v);
    }
  }
}

// one lane group per unique row touched by the coming step (keys from er_emb_route); brings the row to
// "after step t-1" where t = *step_counter - 1 is the step being executed
template <int V>
__device__ __forceinline__ void catch_up_body(int bid, const uint32_t* __restrict__ ukeys,
                                              const int32_t* __restrict__ n_unique, int64_t capacity, const RowUpdate& tab,
                                              const float* __restrict__ lr_hist, const er_opt_hyper* __restrict__ hyper,
                                              int dim, int G, const DecayAux& aux, uint32_t* __restrict__ smem) {
  const int64_t i = (static_cast<int64_t>(bid) * kBlock + threadIdx.x) / G;
  const int c = (static_cast<int>(threadIdx.x) % G) * V;
  const int32_t t = static_cast<int32_t>(*tab.step_counter - 1);
  bool has = i < capacity && i < *n_unique && c < dim;
  int64_t off = 0;
  int32_t s_begin = t;
  if (has) {
    const uint32_t key = ukeys[i];
    s_begin = tab.ls(key) + 1;  // (touched at step t-1, or never updated yet and current: nothing pending)
    off = tab.off(key, c);
  }
  replay_block<V>(has, off, s_begin, t, tab, lr_hist, aux, hyper, smem);  // (synchronises first: every lane has read last_step)
  // The row is current up to step t-1 now.  This step's row update will set last_step = t; until then the record must
  // already say t-1, because the rolling flush of the step may run CONCURRENTLY (er_emb_flush_window with lag 1 on a
  // second stream) and has to find nothing pending on the rows the step touches.
  if (has && c == 0 && s_begin < t) tab.ls(ukeys[i]) = t - 1;
}

template <int V>
__global__ void __launch_bounds__(kBlock)
emb_catch_up_kernel(const uint32_t* __restrict__ ukeys, const int32_t* __restrict__ n_unique, int64_t capacity,
                    RowUpdate tab, const float* __restrict__ lr_hist, const er_opt_hyper* __restrict__ hyper, int dim,
                    int G, DecayAux aux) {
  __shared__ uint32_t smem[kReplaySmemWords];
  catch_up_body<V>(blockIdx.x, ukeys, n_unique, capacity, tab, lr_hist, hyper, dim, G, aux, smem);
}

// Horizontal fusion: the same per-group work of up to kMaxMulti table groups in ONE grid - workgroups
// [start[i], start[i + 1]) run group i's body (a wave-uniform dispatch on its lane width V).  The groups' kernels are
// independent chains of dependent random accesses that leave most CUs idle: side by side they overlap, and the step
// has one launch instead of one per group.
constexpr int kMaxMulti = 4;

struct CatchUpArgs {
  const uint32_t* ukeys;
  const int32_t* n_unique;
  int64_t capacity;
  RowUpdate tab;
  const float* lr_hist;
  DecayAux aux;
  int dim, G, V;
};
struct CatchUpMulti {
  int n;
  int start[kMaxMulti + 1];
  const er_opt_hyper* hyper;
  CatchUpArgs a[kMaxMulti];
};

__global__ void __launch_bounds__(kBlock)
emb_catch_up_multi_kernel(CatchUpMulti ma) {
  int i = 0;
  while (i + 1 < ma.n && static_cast<int>(blockIdx.x) >= ma.start[i + 1]) ++i;
  __shared__ uint32_t smem[kReplaySmemWords];
  const CatchUpArgs& a = ma.a[i];
  const int bid = blockIdx.x - ma.start[i];
  if (a.V == 4) catch_up_body<4>(bid, a.ukeys, a.n_unique, a.capacity, a.tab, a.lr_hist, ma.hyper, a.dim, a.G, a.aux, smem);
  else catch_up_body<1>(bid, a.ukeys, a.n_unique, a.capacity, a.tab, a.lr_hist, ma.hyper, a.dim, a.G, a.aux, smem);
}

// The same catch-up in closed form (er_decay.h): no replay queues, no LDS - a lane group loads its row's pending
// interval, evaluates the closed form and stores; the lanes of a row sit in one wavefront (G <= 64, a power of two), so
// all of them have read last_step before lane 0 stores it.
template <int V>
__device__ __forceinline__ void catch_up_closed_body(int bid, const CatchUpArgs& a, const er_opt_hyper* __restrict__ hyper) {
  const int64_t i = (static_cast<int64_t>(bid) * kBlock + threadIdx.x) / a.G;
  const int c = (static_cast<int>(threadIdx.x) % a.G) * V;
  if (i >= a.capacity || i >= *a.n_unique || c >= a.dim) return;
  const int32_t t = static_cast<int32_t>(*a.tab.step_counter - 1);
  const uint32_t key = a.ukeys[i];
  const int32_t s_begin = a.tab.ls(key) + 1;
  if (s_begin >= t) return;
  const int64_t off = a.tab.off(key, c);
  float var[V], m[V], v[V];
  ld_vec<V>(m, a.tab.m + off);
  ld_vec<V>(v, a.tab.v + off);
  bool live = false;
#pragma unroll
  for (int j = 0; j < V; ++j) live = live || (m[j] != 0.f) || (v[j] != 0.f);
  if (live) {
    ld_vec<V>(var, a.tab.var + off);
    replay_closed<V>(var, m, v, a.aux, s_begin, t, pin_scalar(hyper->eps));
    st_vec<V>(a.tab.var + off, var);
    st_vec<V>(a.tab.m + off, m);
    st_vec<V>(a.tab.v + off, v);
  }
  if (c == 0) a.tab.ls(key) = t - 1;
}

__global__ void __launch_bounds__(kBlock)
emb_catch_up_closed_kernel(CatchUpMulti ma) {
  int i = 0;
  while (i + 1 < ma.n && static_cast<int>(blockIdx.x) >= ma.start[i + 1]) ++i;
  const CatchUpArgs& a = ma.a[i];
  const int bid = blockIdx.x - ma.start[i];
  if (a.V == 4) catch_up_closed_body<4>(bid, a, ma.hyper);
  else catch_up_closed_body<1>(bid, a, ma.hyper);
}

__global__ void decay_lag_sync_kernel(int64_t* __restrict__ lag, const int64_t* __restrict__ counter) {
  if (threadIdx.x == 0 && blockIdx.x == 0) lag[0] = *counter;
}

// A[k-1][:] = T(s_end-1-k, k), k = 1..K, s_end = *counter - lag: one wavefront per k (er_decay.h)
__global__ void __launch_bounds__(kBlock)
decay_tables_kernel(DecayTabDev t, const float* __restrict__ hist, const int64_t* __restrict__ counter, int lag) {
  const int k = static_cast<int>((blockIdx.x * kBlock + threadIdx.x) >> 6) + 1;
  if (k > t.K) return;
  const int64_t s_end = *counter - lag;
  const float mine = decay_sum_wave(t, hist, s_end - 1 - k, k);
  const int lane = threadIdx.x & 63;
  float* A = t.A + static_cast<int64_t>(lag) * kDecayKMax * kDecayLd;  // (one table per lag: see decay_aux_for)
  if (lane < kDecayLd) A[static_cast<int64_t>(k - 1) * kDecayLd + lane] = mine;  // (lanes >= kDecayN hold 0)
}

// every row: replay the pending decay steps up to and including step (*step_counter - 1); last_step = that
template <int V>
__global__ void __launch_bounds__(kBlock)
emb_flush_decay_kernel(RowUpdate tab, int64_t total_rows, const float* __restrict__ lr_hist,
                       const er_opt_hyper* __restrict__ hyper, int dim, int G, DecayAux aux) {
  const int64_t row = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / G;
  const int sub = static_cast<int>(threadIdx.x) % G;
  const int c = sub * V;
  if (row >= total_rows || c >= dim) return;
  const int32_t done = static_cast<int32_t>(*tab.step_counter);  // steps 0 .. done-1 have been executed
  const int32_t last = tab.ls(row);
  if (last + 1 < done) {
    const int64_t off = tab.off(row, c);
    float var[V], m[V], v[V];
    ld_vec<V>(var, tab.var + off);
    ld_vec<V>(m, tab.m + off);
    ld_vec<V>(v, tab.v + off);
    bool live = false;
#pragma unroll
    for (int j = 0; j < V; ++j) live = live || (m[j] != 0.f) || (v[j] != 0.f);
    if (live) {
      if (aux.A != nullptr) replay_closed<V>(var, m, v, aux, last + 1, done, pin_scalar(hyper->eps));
      else replay_decay<V>(var, m, v, LrHist{lr_hist, nullptr, 0}, last + 1, done, pin_decay_consts(*hyper), lr_cap_twice(aux.lr_max, done));
      st_vec<V>(tab.var + off, var);
      st_vec<V>(tab.m + off, m);
      st_vec<V>(tab.v + off, v);
    }
  }
}

// ROLLING flush (er_emb_flush_window): step t brings the rows of window (t mod n_windows) - rows
// [w * chunk, (w + 1) * chunk) - current.  Every row is visited once per n_windows steps, so no row is ever more
// than n_windows steps behind: the catch-up of a touched row replays at most that many steps (instead of the
// thousands a cold row of a 1 M-row table accumulates in a real epoch), at a cost of one pass over 1 / n_windows of
// the tables per step (10.6 GB / 256 = 41 MB for DeepFM-Criteo).  Same arithmetic as the full flush -> still
// bit-identical to the every-row sweep.  The G lanes of a row sit in one wavefront: all of them have read last_step
// (one instruction) before lane 0 stores it.
template <int V>
__device__ __forceinline__ void flush_window_body(int bid, const RowUpdate& tab, int64_t total_rows, int64_t chunk,
                                                  int n_windows, int lag, const float* __restrict__ lr_hist,
                                                  const DecayAux& aux,
                                                  const er_opt_hyper* __restrict__ hyper, int dim, int G,
                                                  uint32_t* __restrict__ smem) {
  // lag 0: called after the step's row updates, rows brought to the step just executed.  lag 1: called DURING step t
  // (after its catch-up), rows brought to step t-1 - the state the catch-up leaves the touched rows in, so the launch
  // has no work on them and can overlap the step's lookup / dense part / row update on another stream.
  const int32_t done = static_cast<int32_t>(*tab.step_counter) - lag;
  const int64_t w = static_cast<int64_t>(done) % n_windows;
  const int64_t row = w * chunk + (static_cast<int64_t>(bid) * kBlock + threadIdx.x) / G;
  const int sub = static_cast<int>(threadIdx.x) % G;
  const int c = sub * V;
  const int64_t end = (w + 1) * chunk < total_rows ? (w + 1) * chunk : total_rows;
  const bool in_window = row < end;
  const bool has = in_window && c < dim;
  const int32_t s_begin = in_window ? tab.ls(row) + 1 : done;
  replay_block<V>(has, tab.off(row, c), s_begin, done, tab, lr_hist, aux, hyper, smem);  // (synchronises first)
  if (in_window && sub == 0 && s_begin < done) tab.ls(row) = done - 1;
}

struct FlushWindowArgs {
  RowUpdate tab;
  int64_t total_rows, chunk;
  const float* lr_hist;
  DecayAux aux;
  int dim, G, V;
};
struct FlushWindowMulti {
  int n, n_windows, lag;
  int start[4 + 1];
  const er_opt_hyper* hyper;
  FlushWindowArgs a[4];
};

// The grid may be smaller than the number of tiles (er_emb_flush_window's max_blocks): each workgroup then walks the
// tiles b, b + gridDim.x, ...  That is the form that runs NEXT TO the step on a second stream: the replay is VALU-bound
// (IEEE sqrt + division per element and pending step), the dense part it overlaps is latency-bound, and a full grid
// (tens of thousands of workgroups, 30 KB of LDS each) would fill every CU and make the step's own kernels queue
// behind it (measured: GEMMs 12 -> 31 us, no net gain); a couple of resident workgroups per CU leave the rest of the
// CU's wave slots and LDS to the step.
__global__ void __launch_bounds__(kBlock)
emb_flush_window_kernel(FlushWindowMulti ma) {
  __shared__ uint32_t smem[kReplaySmemWords];
  const int total = ma.start[ma.n];
  for (int b = blockIdx.x; b < total; b += gridDim.x) {
    int i = 0;
    while (i + 1 < ma.n && b >= ma.start[i + 1]) ++i;
    const FlushWindowArgs& a = ma.a[i];
    const int bid = b - ma.start[i];
    if (a.V == 4) flush_window_body<4>(bid, a.tab, a.total_rows, a.chunk, ma.n_windows, ma.lag, a.lr_hist, a.aux, ma.hyper, a.dim, a.G, smem);
    else flush_window_body<1>(bid, a.tab, a.total_rows, a.chunk, ma.n_windows, ma.lag, a.lr_hist, a.aux, ma.hyper, a.dim, a.G, smem);
    __syncthreads();  // (the next tile resets the queues' counters)
  }
}

// second pass of the flush (all lanes of a row must have read last_step before it changes)
__global__ void __launch_bounds__(kBlock)
emb_flush_mark_kernel(int32_t* __restrict__ last_step, int64_t ls_ld, int64_t total_rows,
                      const int64_t* __restrict__ step_counter) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const int32_t v = static_cast<int32_t>(*step_counter) - 1;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total_rows; i += stride) last_step[i * ls_ld] = v;
}

// ------------------------------------------------------------------------------------------------
// forward
//
// LAZY (er_emb_fwd_lazy, the fused single-GPU step): a lookup whose table group decays lazily (TF-exact Adam without the
// sweep, closed-form replay) brings every row it reads current IN REGISTERS - the row's pending decay-only steps are
// evaluated from its record (var, m, v, last_step: one contiguous access under er_emb_group_set_row_pitch) and only the
// caught-up var enters the sum; nothing is written back.  The row update of the same step (finish_run) repeats the same
// evaluation on the same bits before it applies the gradient, so the state a step leaves is exactly that of a
// catch-up launch followed by lookup and update - without the launch and without writing and re-reading var, m, v.
// ------------------------------------------------------------------------------------------------
struct FwdLazy {
  int n;                       // table groups with lazy decay (0: plain lookups only)
  const int8_t* lookup_group;  // [n_lookups] index into tab / aux, or -1 (device memory)
  const er_opt_hyper* hyper;
  // the lag-1 replay table was built by the step prologue (er_decay_tables_set_prologue_build): the lookup launch checks
  // that it was built for THIS step and leaves the counter for the next prologue (DecayTabDev.lag); else nullptr
  int64_t* lag;
  const int64_t* counter;
  RowUpdate tab[kMaxMulti];
  DecayAux aux[kMaxMulti];
};

template <int V>
__device__ __forceinline__ void lazy_row(float (&var)[V], const RowUpdate& tab, const DecayAux& aux, int64_t key, int c,
                                         int32_t t, float eps) {
  const int64_t off = tab.off(key, c);
  ld_vec<V>(var, tab.var + off);
  const int32_t s_begin = tab.ls(key) + 1;
  if (s_begin >= t) return;
  float m[V], v[V];
  ld_vec<V>(m, tab.m + off);
  ld_vec<V>(v, tab.v + off);
  bool live = false;
#pragma unroll
  for (int j = 0; j < V; ++j) live = live || (m[j] != 0.f) || (v[j] != 0.f);
  if (live) replay_closed<V>(var, m, v, aux, s_begin, t, eps);
}

// one block of kBlock lanes of the lookup launch: block `vb` of the plan, lane `vt` of it (a workgroup of the merged
// sort + lookup launch holds several such blocks); returns the lane's share of the sum of squares of what it wrote
template <bool LAZY>
__device__ __forceinline__ float fwd_rows(int vb, int vt, const er_lookup_desc* __restrict__ descs,
                                          const int32_t* __restrict__ blk_start, int n_lookups, const FwdLazy* lz) {
  const int l = find_lookup(blk_start, n_lookups, vb);
  const er_lookup_desc d = descs[l];
  int V, G;
  lane_geom(d.dim, V, G);
  const int rows_per_block = kBlock / G;
  const int r = (vb - blk_start[l]) * rows_per_block + vt / G;
  const int c = (vt % G) * V;
  if (LAZY && lz->lag != nullptr && vb == 0 && vt == 0) {
    const int64_t cnt = *lz->counter;
    if (lz->lag[1] != cnt - 1) lz->lag[2] = 1;  // the prologue built the table for another step: the step is void
    lz->lag[0] = cnt;
  }
  // (uniform over the workgroup: the group's records are read with scalar loads from the kernel arguments)
  int gi = -1;
  if (LAZY) gi = __builtin_amdgcn_readfirstlane(static_cast<int>(lz->lookup_group[l]));
  int32_t t_now = 0;
  float eps = 0.f;
  if (LAZY && gi >= 0) {
    t_now = static_cast<int32_t>(*lz->tab[gi].step_counter - 1);
    eps = pin_scalar(lz->hyper->eps);
  }
  float ss = 0.f;
  if (r < d.n_rows && c < d.dim) {
    int64_t kb, ke;
    if (d.offsets) {
      kb = d.offsets[r];
      ke = d.offsets[r + 1];
    } else {
      kb = r;
      ke = r + 1;
    }
    const bool prune_nonpos = (d.weights != nullptr) && (d.combiner != ER_COMBINER_SUM);
    Acc4 a{0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f, w2sum = 0.f;
    for (int64_t k = kb; k < ke; ++k) {
      const int64_t id = d.ids[k];
      if (id < 0 || id >= d.rows) continue;
      float w = 1.f;
      if (d.weights) {
        w = d.weights[k];
        if (prune_nonpos && !(w > 0.f)) continue;
      }
      Acc4 e{0.f, 0.f, 0.f, 0.f};
      if (LAZY && gi >= 0) {
        if (V == 4) {
          float q[4];
          lazy_row<4>(q, lz->tab[gi], lz->aux[gi], d.key_base + id, c, t_now, eps);
          e = Acc4{q[0], q[1], q[2], q[3]};
        } else {
          float q[1];
          lazy_row<1>(q, lz->tab[gi], lz->aux[gi], d.key_base + id, c, t_now, eps);
          e.x = q[0];
        }
      } else {
        const float* row = d.table + id * (d.table_ld ? d.table_ld : d.dim) + c;
        if (V == 4) {
          const float4 q = *reinterpret_cast<const float4*>(row);
          e = Acc4{q.x, q.y, q.z, q.w};
        } else {
          e.x = row[0];
        }
      }
      if (d.weights) {
        a.x = a.x + e.x * w; a.y = a.y + e.y * w; a.z = a.z + e.z * w; a.w = a.w + e.w * w;
        wsum = wsum + w;
        w2sum = w2sum + w * w;
      } else {
        a.x = a.x + e.x; a.y = a.y + e.y; a.z = a.z + e.z; a.w = a.w + e.w;
        wsum = wsum + 1.f;
        w2sum = w2sum + 1.f;
      }
    }
    if (d.combiner != ER_COMBINER_SUM && wsum != 0.f) {
      const float den = (d.combiner == ER_COMBINER_MEAN) ? wsum : sqrtf(w2sum);
      a.x = a.x / den; a.y = a.y / den; a.z = a.z / den; a.w = a.w / den;
    }
    float* o = d.out + static_cast<int64_t>(r) * d.out_stride + d.out_col + c;
    if (V == 4) {
      if (((d.out_stride | d.out_col) & 3) == 0 && (reinterpret_cast<uintptr_t>(d.out) & 15) == 0) {
        *reinterpret_cast<float4*>(o) = make_float4(a.x, a.y, a.z, a.w);
      } else {
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
      }
      ss = (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    } else {
      o[0] = a.x;
      ss = a.x * a.x;
    }
  }
  return ss;
}

__global__ void __launch_bounds__(kBlock)
emb_fwd_kernel(const er_lookup_desc* __restrict__ descs, const int32_t* __restrict__ blk_start, int n_lookups,
               float* __restrict__ sumsq_partials) {
  __shared__ float red[4];
  const float ss = fwd_rows<false>(blockIdx.x, threadIdx.x, descs, blk_start, n_lookups, nullptr);
  if (sumsq_partials) {
    const float tot = block_sum_256(ss, red);
    if (threadIdx.x == 0) sumsq_partials[blockIdx.x] = tot;
  }
}

__global__ void __launch_bounds__(kBlock)
emb_fwd_lazy_kernel(const er_lookup_desc* __restrict__ descs, const int32_t* __restrict__ blk_start, int n_lookups,
                    float* __restrict__ sumsq_partials, FwdLazy lz) {
  __shared__ float red[4];
  const float ss = fwd_rows<true>(blockIdx.x, threadIdx.x, descs, blk_start, n_lookups, &lz);
  if (sumsq_partials) {
    const float tot = block_sum_256(ss, red);
    if (threadIdx.x == 0) sumsq_partials[blockIdx.x] = tot;
  }
}

// ------------------------------------------------------------------------------------------------
// De-duplicated gradient = segmented sum over the SORTED entries, then the row-wise optimizer.
//
// Tile kernel: a workgroup owns T = 4 * (256 / G) consecutive sorted entries (G lanes x 16 B per row).
//   1. gather the entries' scaled upstream-gradient rows into LDS (random 64 B reads, 4 in flight/lane);
//   2. in-LDS segmented inclusive scan over the tile (Hillis-Steele with key equality as the segment
//      flag: log2(T) steps, fixed combination order -> deterministic, hot keys cost log T not run length);
//   3. the last entry of every run that lies completely inside the tile applies the optimizer to its
//      row (each touched row's var/m/v is read and written once) or, in reduce mode, emits the sum;
//      runs that cross a tile boundary leave their partial in tile_first / tile_last.
// Fix kernel: one lane group per tile whose LAST run starts in it and crosses into the next tile:
//   sum = tile_last[s] + tile_last[whole tiles] + tile_first[end tile], then the same apply / emit.
// ------------------------------------------------------------------------------------------------
struct ReduceOut {
  int mode;                      // 0: apply optimizer, 1: emit (out_grads[u], u = index of the run),
                                 // 2: emit into a dense row-indexed buffer (out_grads[key * ld + 0 .. dim), 1 at + dim)
  const uint32_t* flags;         // mode 1: head flags / exclusive scan of them over sorted positions
  const uint32_t* head_index;
  uint32_t* out_keys;
  float* out_grads;
  int ld;                        // mode 2: floats per row of out_grads (>= dim + 1)
};

// TF-exact Adam on a row whose pending decay-only steps have NOT been replayed by a catch-up launch (the fused step with
// er_emb_fwd_lazy): the row's record is read once, caught up in registers (the evaluation the lookup made on the same
// bits), stepped and written once.  The G lanes of the row sit in one wavefront: all read last_step before lane 0 of the
// row stores it (finish_run).
template <int V>
__device__ __forceinline__ void update_row_lazy(const RowUpdate& t, const DecayAux& aux, const er_opt_hyper& h, int64_t key,
                                                int c, const float* g) {
  const int64_t off = t.off(key, c);
  float var[V], m[V], v[V];
  ld_vec<V>(var, t.var + off);
  ld_vec<V>(m, t.m + off);
  ld_vec<V>(v, t.v + off);
  const int32_t now = static_cast<int32_t>(*t.step_counter - 1);
  const int32_t s_begin = t.ls(key) + 1;
  if (s_begin < now) {
    bool live = false;
#pragma unroll
    for (int j = 0; j < V; ++j) live = live || (m[j] != 0.f) || (v[j] != 0.f);
    if (live) replay_closed<V>(var, m, v, aux, s_begin, now, pin_scalar(h.eps));
  }
#pragma unroll
  for (int i = 0; i < V; ++i) var[i] = adam_elem(m[i], v[i], var[i], g[i], h);
  st_vec<V>(t.m + off, m);
  st_vec<V>(t.v + off, v);
  st_vec<V>(t.var + off, var);
}

// aux.A != nullptr: the rows of this step have not been caught up by a launch of their own (update_row_lazy).  (By
// reference to the kernel-argument record, like tab: a POINTER to a by-value argument's field makes the compiler copy the
// whole argument struct to scratch - 1 KB per lane, +10 us on the fix launch.)
template <int V>
__device__ __forceinline__ void finish_run(const RowUpdate& tab, int opt_kind, const er_opt_hyper* hyper,
                                           const ReduceOut& ro, uint32_t key, int64_t p, int sub, int c, int dim,
                                           const float* gsum, const DecayAux& aux = DecayAux{}) {
  float g[V];
#pragma unroll
  for (int i = 0; i < V; ++i) g[i] = gsum[i];
  if (ro.mode == 0) {
    const er_opt_hyper h = *hyper;
#pragma unroll
    for (int i = 0; i < V; ++i) g[i] = g[i] * h.grad_scale;
    if (h.clip_scale != 0.f) {  // clip_by_global_norm (er_clip_scale); 0 = no clipping
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] = g[i] * h.clip_scale;
    }
    if (aux.A != nullptr && opt_kind == ER_OPT_ADAM && tab.last_step != nullptr) update_row_lazy<V>(tab, aux, h, key, c, g);
    else update_row<V>(tab, opt_kind, h, tab.off(key, c), g);
    if (opt_kind == ER_OPT_ADAM && sub == 0) {
      if (tab.last_step) tab.ls(key) = static_cast<int32_t>(*tab.step_counter - 1);
      else atomicOr(&tab.bitmap[key >> 5], 1u << (key & 31));
    }
  } else if (ro.mode == 2) {
    float* d = ro.out_grads + static_cast<int64_t>(key) * ro.ld;
#pragma unroll
    for (int i = 0; i < V; ++i) d[c + i] = g[i];
    if (sub == 0) d[dim] = 1.f;  // "this row has a gradient" (summed over ranks by the all-reduce that follows)
  } else {
    const uint32_t u = ro.head_index[p] + ro.flags[p] - 1u;
    if (sub == 0 && ro.out_keys) ro.out_keys[u] = key;
#pragma unroll
    for (int i = 0; i < V; ++i) ro.out_grads[static_cast<int64_t>(u) * (ro.ld ? ro.ld : dim) + c + i] = g[i];
  }
}

// entries per lane of a tile: 4 for 16-byte lanes (dim 16: 64 entries per pass -> tiles of 256), 2 for scalar lanes
// (dim 1: 256 entries per pass -> tiles of 512).  The kernels are latency-bound (dependent random reads), so a group of
// 100 k entries should spread over hundreds of workgroups - but in the fused own launch the dim-1 tiles of DeepFM are the
// critical path (per-workgroup stamps: profiles/r05_own_launch_workgroup_stamps.txt) and 1,664 workgroups need two
// rounds on the chip: 512-entry dim-1 tiles (two gathers in flight per lane) take the own launch from 35.7 to ~32.5 us
// (1024-entry tiles: 33.4; 8 instead of 16 column-reduction parts per one-row table: no further gain, and a longer serial
// sum per part; 5 waves per SIMD: none; profiles/r05_own_launch_knobs_ab_lines.txt)
#ifndef ER_TILE_PASSES_V1
#define ER_TILE_PASSES_V1 2
#endif
__host__ __device__ constexpr int tile_passes(int V) { return V == 4 ? 4 : ER_TILE_PASSES_V1; }

template <int V>
__device__ __forceinline__ void tile_body(int bid, float* __restrict__ smem, const uint32_t* __restrict__ skeys,
                                          const uint32_t* __restrict__ svals, const float* const* __restrict__ ent_gptr,
                                          const float* __restrict__ ent_scale, int64_t n, int dim, int G,
                                          const RowUpdate& tab, int opt_kind, const er_opt_hyper* __restrict__ hyper,
                                          const ReduceOut& ro, float* __restrict__ tile_first,
                                          float* __restrict__ tile_last) {
  const int epp = kBlock / G;            // entries per pass
  constexpr int kTilePasses = tile_passes(V);
  const int T = kTilePasses * epp;       // entries per tile
  float* vals = smem;                    // [T][dim]
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem + static_cast<size_t>(T) * dim);  // [T + 2]: prev, tile, next
  const int tid = threadIdx.x;
  const int sub = tid % G;
  const int c = sub * V;
  const int64_t t0 = static_cast<int64_t>(bid) * T;
  for (int i = tid; i < T + 2; i += kBlock) {
    const int64_t p = t0 - 1 + i;
    keys[i] = (p >= 0 && p < n) ? skeys[p] : kInvalidKey;
  }
  __syncthreads();
  const bool col_ok = c < dim;
  // 1. gather
#pragma unroll
  for (int ps = 0; ps < kTilePasses; ++ps) {
    const int e = ps * epp + tid / G;
    const int64_t p = t0 + e;
    Vec<V> acc;
    acc.zero();
    if (col_ok && p < n && keys[e + 1] != kInvalidKey) {
      const uint32_t j = svals[p];
      gather_grad<V>(acc, ent_gptr[j], c, ent_scale[j]);
    }
    if (col_ok) acc.store(vals + static_cast<size_t>(e) * dim + c);
  }
  __syncthreads();
  // 2. segmented inclusive scan (sorted keys: equal key at distance `off` => same run)
  for (int off = 1; off < T; off <<= 1) {
    Vec<V> add[kTilePasses];
    int any = 0;
#pragma unroll
    for (int ps = 0; ps < kTilePasses; ++ps) {
      const int e = ps * epp + tid / G;
      add[ps].zero();
      if (col_ok && e >= off && keys[e + 1] != kInvalidKey && keys[e + 1 - off] == keys[e + 1]) {
        add[ps].load(vals + static_cast<size_t>(e - off) * dim + c);
        any = 1;
      }
    }
    // sorted keys: no equal pair at distance `off` means no run of the tile is longer than `off` - the scan is done
    // (the barrier doubles as the one that separates the reads above from the writes below)
    if (!__syncthreads_or(any)) break;
#pragma unroll
    for (int ps = 0; ps < kTilePasses; ++ps) {
      const int e = ps * epp + tid / G;
      if (col_ok && e >= off && keys[e + 1] != kInvalidKey && keys[e + 1 - off] == keys[e + 1]) {
        Vec<V> cur;
        cur.load(vals + static_cast<size_t>(e) * dim + c);
        cur.add(add[ps]);
        cur.store(vals + static_cast<size_t>(e) * dim + c);
      }
    }
    __syncthreads();
  }
  // 3. run ends
#pragma unroll
  for (int ps = 0; ps < kTilePasses; ++ps) {
    const int e = ps * epp + tid / G;
    const int64_t p = t0 + e;
    if (!col_ok || p >= n) continue;
    const uint32_t key = keys[e + 1];
    if (key == kInvalidKey) continue;
    if (keys[e + 2] == key && e != T - 1) continue;           // not the last entry of its run in this tile
    const bool from_prev = keys[0] == key;                    // sorted: then the run covers the tile up to e
    const bool to_next = (e == T - 1) && keys[T + 1] == key;  // keys[T + 1] = first key of the next tile
    const float* gs = vals + static_cast<size_t>(e) * dim + c;
    if (!from_prev && !to_next) {
      finish_run<V>(tab, opt_kind, hyper, ro, key, p, sub, c, dim, gs);
    } else {
      Vec<V> r;
      r.load(gs);
      if (from_prev) r.store(tile_first + static_cast<size_t>(bid) * dim + c);
      if (to_next) r.store(tile_last + static_cast<size_t>(bid) * dim + c);
    }
  }
}

template <int V>
__global__ void __launch_bounds__(kBlock)
emb_bwd_tile_kernel(const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ svals,
                    const float* const* __restrict__ ent_gptr, const float* __restrict__ ent_scale, int64_t n,
                    int dim, int G, RowUpdate tab, int opt_kind, const er_opt_hyper* __restrict__ hyper,
                    ReduceOut ro, float* __restrict__ tile_first, float* __restrict__ tile_last) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  tile_body<V>(blockIdx.x, smem, skeys, svals, ent_gptr, ent_scale, n, dim, G, tab, opt_kind, hyper, ro, tile_first,
               tile_last);
}

template <int V>
__device__ __forceinline__ void fix_body(int bid, const uint32_t* __restrict__ skeys, int64_t n, int dim, int G, int T,
                                         int n_tiles, const RowUpdate& tab, int opt_kind,
                                         const er_opt_hyper* __restrict__ hyper, const ReduceOut& ro,
                                         const float* __restrict__ tile_first, const float* __restrict__ tile_last,
                                         const DecayAux& aux = DecayAux{}) {
  const int64_t s = (static_cast<int64_t>(bid) * kBlock + threadIdx.x) / G;  // start tile candidate
  const int sub = static_cast<int>(threadIdx.x) % G;
  const int c = sub * V;
  if (s >= n_tiles || c >= dim) return;
  const int64_t next0 = (s + 1) * T;
  if (next0 >= n) return;                       // last tile: nothing to cross into
  const uint32_t key = skeys[next0 - 1];
  if (key == kInvalidKey || skeys[next0] != key) return;  // the tile's last run ends with the tile
  if (s > 0 && skeys[s * T - 1] == key) return;            // the run came from an earlier tile: not its owner
  Vec<V> acc;
  acc.load(tile_last + s * dim + c);
  for (int64_t m = s + 1; m < n_tiles; ++m) {
    const int64_t mnext = (m + 1) * T;
    Vec<V> part;
    if (mnext < n && skeys[mnext - 1] == key && skeys[mnext] == key) {  // whole tile m, and it goes on
      part.load(tile_last + m * dim + c);
      acc.add(part);
    } else {                                                            // the run ends inside tile m
      part.load(tile_first + m * dim + c);
      acc.add(part);
      break;
    }
  }
  float g[V];
  if constexpr (V == 4) { g[0] = acc.v.x; g[1] = acc.v.y; g[2] = acc.v.z; g[3] = acc.v.w; } else { g[0] = acc.v; }
  finish_run<V>(tab, opt_kind, hyper, ro, key, next0 - 1, sub, c, dim, g, aux);
}

template <int V>
__global__ void __launch_bounds__(kBlock)
emb_bwd_fix_kernel(const uint32_t* __restrict__ skeys, int64_t n, int dim, int G, int T, int n_tiles,
                   RowUpdate tab, int opt_kind, const er_opt_hyper* __restrict__ hyper, ReduceOut ro,
                   const float* __restrict__ tile_first, const float* __restrict__ tile_last) {
  fix_body<V>(blockIdx.x, skeys, n, dim, G, T, n_tiles, tab, opt_kind, hyper, ro, tile_first, tile_last);
}

// ------------------------------------------------------------------------------------------------
// The fused single-GPU backward (er_emb_bwd_fused): the tile kernel with er_group_grad_finish folded in, + the fix kernel.
//   * the finished output gradient is evaluated WHILE gathering (grad_finish_value: base + deferred terms + lambda * out) -
//     the [B, sum(dim)] buffer is never rewritten and re-read;
//   * runs that cross tile boundaries are combined by the fix kernel, as in the three-launch path (second launch);
//   * lookups into ONE-ROW tables (the RawFeature projections: B entries, one key) do not go through the sort at all
//     (emb_front_sort_kernel skips them): each is a weighted column sum of the finished gradient, reduced by kProjParts
//     workgroups whose LAST arriver (one agent-scope ticket) combines the partials in a fixed order and applies the
//     optimizer to the row.
// Entry j of a group -> (lookup l, output row r) by a binary search over the lookups' entry offsets (dense-mode lookups:
// entry i of a lookup is its row i); the per-entry pointer / scale arrays of the three-launch path are not read.
// ------------------------------------------------------------------------------------------------
#ifndef ER_PROJ_PARTS
#define ER_PROJ_PARTS 16
#endif
constexpr int kProjParts = ER_PROJ_PARTS;
constexpr int kOwnMaxLookups = 128;
constexpr int kOwnMaxGG = 8;

struct OwnLookup;
struct OwnArgs {
  const uint32_t* skeys;
  const uint32_t* svals;
  int64_t n;
  const er_lookup_desc* descs;   // this group's lookups (out = the feature group's gradient buffer)
  const int64_t* ent_base;
  const OwnLookup* lookups;      // [n_lookups] gather records (device memory; built by er_emb_bwd_fused)
  int cap_shift;                 // >= 0: every lookup holds 2^cap_shift entries (lookup of entry j = j >> cap_shift)
  int n_lookups;
  int dim, G, V, n_tiles;
  RowUpdate tab;
  DecayAux aux;                  // A != nullptr: this step's rows were not caught up by a launch (er_emb_fwd_lazy) - the
                                 // closed-form replay's tables (lag 1) for the row update to do it in registers
  float* tile_first;             // partial sums of the runs that cross tile boundaries (emb_bwd_fix_multi_kernel)
  float* tile_last;
  int n_proj;
  const int32_t* proj_lookup;    // [n_proj] lookups into one-row tables
  float* proj_partial;           // [n_proj][kProjParts][dim + 1] (last: number of valid entries)
  uint32_t* proj_ticket;         // [n_proj] monotonic arrival counters
};
struct OwnMulti {
  int n;
  int start[kMaxMulti + 1];       // tile workgroups
  int proj_start[kMaxMulti + 1];  // projection workgroups (after all tiles)
  int opt_kind;
  const er_opt_hyper* hyper;
  unsigned long long* dbg;        // probe hook (er_debug_stamps): 16 wall-clock stamps per workgroup, or nullptr
  int n_gg;
  er_grad_group gg[kOwnMaxGG];    // (their term pointers change every step: by value; a workgroup copies them to LDS once -
  OwnArgs a[kMaxMulti];           //  the gather indexes them per lane, which on a kernel argument would be a vector load
};                                //  from the kernarg segment per access)

struct OwnLookup {  // what the gather needs of a lookup: built on the host (er_emb_bwd_fused), copied to LDS per workgroup
  const float* weights;
  int32_t base;      // first entry
  int32_t out_col;
  int32_t combiner;
  int8_t gg;         // which finish descriptor
  uint8_t tmask;     // which of its terms cover the lookup's column block (a term covers a block entirely or not at all)
  int16_t pad_;
};

// scale of a dense-mode lookup's single id (emb_bwd_build_kernel: w / den; den = sum w | sqrt(sum w^2) for mean | sqrtn)
__device__ __forceinline__ float own_scale(const OwnLookup& L, int r) {
  const float w = L.weights ? L.weights[r] : 1.f;
  float den = 1.f;
  if (L.combiner != ER_COMBINER_SUM) {
    den = (L.combiner == ER_COMBINER_MEAN) ? w : sqrtf(w * w);
    if (w == 0.f) den = 1.f;
  }
  return w / den;
}

__device__ __forceinline__ int own_find(const OwnLookup* __restrict__ L, int n, int j, int cap_shift) {
  if (cap_shift >= 0) return j >> cap_shift;
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (L[mid].base <= j) lo = mid; else hi = mid;
  }
  return lo;
}

template <int V>
__device__ __forceinline__ Vec<V> own_ld(const float* p) {
  Vec<V> v;
  if (V == 4 && (reinterpret_cast<uintptr_t>(p) & 15) != 0) v.loadu(p); else v.load(p);
  return v;
}

// V columns [col, col + V) of row b of the finished output gradient (grad_finish_value's arithmetic, component by
// component in the same order): base + the terms of tmask + lambda * out; cb = the first column's offset inside the
// lookup's block (= the FM term's d: the host checked that the block is one field of the term)
template <int V>
__device__ __forceinline__ Vec<V> own_finish(const er_grad_group& g, uint32_t tmask, int64_t b, int col, int cb) {
  Vec<V> v;
  if (g.has_base) v = own_ld<V>(g.dout + b * g.ld + col); else v.zero();
  Vec<V> o;
  o.zero();
  const bool need_out = g.lambda != 0.f || (tmask & 0xF0u) != 0;  // (high nibble: an FM term is among them)
  if (need_out) o = own_ld<V>(g.out + b * g.ld + col);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (!((tmask >> t) & 1u)) continue;
    const er_grad_term& q = g.terms[t];
    if (q.kind == ER_GRAD_TERM_ROWSUM) {
      const float sgl = q.g[b * q.g_ld];
      if constexpr (V == 4) { v.v.x = v.v.x + sgl; v.v.y = v.v.y + sgl; v.v.z = v.v.z + sgl; v.v.w = v.v.w + sgl; } else { v.v = v.v + sgl; }
    } else {
      const Vec<V> gf = own_ld<V>(q.g + b * q.g_ld + cb), sv = own_ld<V>(q.saved + b * q.dim + cb);
      if constexpr (V == 4) {
        v.v.x = v.v.x + gf.v.x * (sv.v.x - o.v.x); v.v.y = v.v.y + gf.v.y * (sv.v.y - o.v.y);
        v.v.z = v.v.z + gf.v.z * (sv.v.z - o.v.z); v.v.w = v.v.w + gf.v.w * (sv.v.w - o.v.w);
      } else {
        v.v = v.v + gf.v * (sv.v - o.v);
      }
    }
  }
  if (g.lambda != 0.f) {
    if constexpr (V == 4) {
      v.v.x = v.v.x + g.lambda * o.v.x; v.v.y = v.v.y + g.lambda * o.v.y;
      v.v.z = v.v.z + g.lambda * o.v.z; v.v.w = v.v.w + g.lambda * o.v.w;
    } else {
      v.v = v.v + g.lambda * o.v;
    }
  }
  return v;
}

// gather + segmented scan of the T entries at [q, q + T) into vals (LDS); keys[0] = key before, keys[1..T] = the chunk,
// keys[T + 1] = key after
template <int V>
__device__ __forceinline__ void own_chunk(int64_t q, const er_grad_group* __restrict__ ggs, const OwnArgs& a,
                                          const OwnLookup* __restrict__ L, float* __restrict__ vals,
                                          uint32_t* __restrict__ keys, int T, unsigned long long* dbg) {
  constexpr int kTilePasses = tile_passes(V);
  const int G = a.G, dim = a.dim;
  const int epp = kBlock / G;
  const int tid = threadIdx.x;
  const int c = (tid % G) * V;
  const bool col_ok = c < dim;
  for (int i = tid; i < T + 2; i += kBlock) {
    const int64_t p = q - 1 + i;
    keys[i] = (p >= 0 && p < a.n) ? a.skeys[p] : kInvalidKey;
  }
  __syncthreads();
  if (dbg && tid == 0) dbg[8] = wall_clock64();
#pragma unroll
  for (int ps = 0; ps < kTilePasses; ++ps) {
    const int e = ps * epp + tid / G;
    const int64_t p = q + e;
    Vec<V> acc;
    acc.zero();
    if (col_ok && p < a.n && keys[e + 1] != kInvalidKey) {
      const int j = static_cast<int>(a.svals[p]);
      const OwnLookup lk = L[own_find(L, a.n_lookups, j, a.cap_shift)];
      const int r = j - lk.base;
      const float sc = own_scale(lk, r);
      const Vec<V> g = own_finish<V>(ggs[lk.gg], lk.tmask, r, lk.out_col + c, c);
      acc.add_scaled(g, sc);
    }
    if (col_ok) acc.store(vals + static_cast<size_t>(e) * dim + c);
  }
  if (dbg && tid == 0) dbg[9] = wall_clock64();
  __syncthreads();
  if (dbg && tid == 0) dbg[10] = wall_clock64();
  for (int off = 1; off < T; off <<= 1) {
    Vec<V> add[kTilePasses];
    int any = 0;
#pragma unroll
    for (int ps = 0; ps < kTilePasses; ++ps) {
      const int e = ps * epp + tid / G;
      add[ps].zero();
      if (col_ok && e >= off && keys[e + 1] != kInvalidKey && keys[e + 1 - off] == keys[e + 1]) {
        add[ps].load(vals + static_cast<size_t>(e - off) * dim + c);
        any = 1;
      }
    }
    if (!__syncthreads_or(any)) break;
#pragma unroll
    for (int ps = 0; ps < kTilePasses; ++ps) {
      const int e = ps * epp + tid / G;
      if (col_ok && e >= off && keys[e + 1] != kInvalidKey && keys[e + 1 - off] == keys[e + 1]) {
        Vec<V> cur;
        cur.load(vals + static_cast<size_t>(e) * dim + c);
        cur.add(add[ps]);
        cur.store(vals + static_cast<size_t>(e) * dim + c);
      }
    }
    __syncthreads();
  }
}

template <int V>
__device__ __forceinline__ void own_tile_body(int bid, const OwnMulti& ma, const OwnArgs& a, float* __restrict__ smem) {
  constexpr int kTilePasses = tile_passes(V);
  const int G = a.G, dim = a.dim;
  const int epp = kBlock / G;
  const int T = kTilePasses * epp;
  float* vals = smem;                                                                         // [T][dim]
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem + static_cast<size_t>(T) * dim);          // [T + 2]
  OwnLookup* L = reinterpret_cast<OwnLookup*>(keys + T + 2 + ((T + 2) & 1));                   // [n_lookups] (8-byte aligned)
  er_grad_group* ggs = reinterpret_cast<er_grad_group*>(L + a.n_lookups);                      // [n_gg]
  const int tid = threadIdx.x;
  const int sub = tid % G;
  const int c = sub * V;
  const bool col_ok = c < dim;
  const int64_t t0 = static_cast<int64_t>(bid) * T;
  unsigned long long* dbg = ma.dbg ? ma.dbg + static_cast<int64_t>(blockIdx.x) * 16 : nullptr;
  if (dbg && tid == 0) dbg[0] = wall_clock64();
  // the lookups' gather records (device memory, built by the host) and the finish descriptors (kernel arguments: copied by
  // ONE wavefront, struct by struct with uniform indices - scalar loads; a per-lane index into a by-value argument would be
  // a vector load from the kernarg segment, which is host memory: measured 140 us per launch)
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.lookups);
    uint32_t* dst = reinterpret_cast<uint32_t*>(L);
    for (int i = tid; i < a.n_lookups * static_cast<int>(sizeof(OwnLookup) / 4); i += kBlock) dst[i] = src[i];
  }
  if (tid < kWave)
    for (int k = 0; k < ma.n_gg; ++k) ggs[k] = ma.gg[k];
  const ReduceOut ro{0, nullptr, nullptr, nullptr, nullptr, 0};
  if (dbg && tid == 0) dbg[2] = wall_clock64();
  own_chunk<V>(t0, ggs, a, L, vals, keys, T, dbg);  // (its first barrier also orders the LDS set-up above before the gather)
  if (dbg && tid == 0) dbg[3] = wall_clock64();
  // run ends, as the three-launch path's tile kernel: a run inside the tile is finished here; a run that crosses a tile
  // boundary leaves its partial in tile_first / tile_last for emb_bwd_fix_multi_kernel.  (A first version had the workgroup
  // that holds a run's FIRST entry follow the run through the tiles behind it - no partials, no second launch - and lost:
  // a tile's gather + scan is 6 us (dim 16) to 17 us (dim 1) of dependent round trips, and a Zipf-hot key's run spans four
  // of them in series on one workgroup while the launch waits: 97 us against 27 + 12.)
#pragma unroll
  for (int ps = 0; ps < kTilePasses; ++ps) {
    const int e = ps * epp + tid / G;
    const int64_t p = t0 + e;
    if (!col_ok || p >= a.n) continue;
    const uint32_t key = keys[e + 1];
    if (key == kInvalidKey) continue;
    if (keys[e + 2] == key && e != T - 1) continue;           // not the last entry of its run in this tile
    const bool from_prev = keys[0] == key;                    // sorted: then the run covers the tile up to e
    const bool to_next = (e == T - 1) && keys[T + 1] == key;  // keys[T + 1] = first key of the next tile
    const float* gs = vals + static_cast<size_t>(e) * dim + c;
    if (!from_prev && !to_next) {
      finish_run<V>(a.tab, ma.opt_kind, ma.hyper, ro, key, p, sub, c, dim, gs, a.aux);
    } else {
      Vec<V> r;
      r.load(gs);
      if (from_prev) r.store(a.tile_first + static_cast<size_t>(bid) * dim + c);
      if (to_next) r.store(a.tile_last + static_cast<size_t>(bid) * dim + c);
    }
  }
  if (dbg && tid == 0) dbg[1] = wall_clock64();
}

// one-row tables: workgroup (projection pj, part k) sums its rows' scaled finished gradients; the last of the kProjParts
// to arrive combines the partials (fixed order) and updates the row
template <int V>
__device__ __forceinline__ void own_proj_body(int local, const OwnMulti& ma, const OwnArgs& a, float* __restrict__ smem) {
  const int pj = local / kProjParts, part = local % kProjParts;
  const int G = a.G, dim = a.dim;
  const int tid = threadIdx.x;
  const int sub = tid % G, rl = tid / G, rpp = kBlock / G;
  const int c = sub * V;
  const bool col_ok = c < dim;
  unsigned long long* dbg = ma.dbg ? ma.dbg + static_cast<int64_t>(blockIdx.x) * 16 : nullptr;
  if (dbg && tid == 0) dbg[0] = wall_clock64();
  const int l = a.proj_lookup[pj];
  const er_lookup_desc d = a.descs[l];
  const OwnLookup lk = a.lookups[l];
  // (uniform index made visibly uniform: the argument array is then read with scalar loads)
  const er_grad_group& gg = ma.gg[__builtin_amdgcn_readfirstlane(static_cast<int>(lk.gg))];
  const int rows_per_part = static_cast<int>(ceil_div(d.n_rows, kProjParts));
  const int r0 = part * rows_per_part;
  const int r1 = r0 + rows_per_part < d.n_rows ? r0 + rows_per_part : d.n_rows;
  Vec<V> acc;
  acc.zero();
  float cnt = 0.f;
  for (int r = r0 + rl; r < r1; r += rpp) {
    const int64_t id = d.ids[r];
    bool ok = id == 0;  // (rows == 1: the only valid id)
    if (d.weights != nullptr && d.combiner != ER_COMBINER_SUM && !(d.weights[r] > 0.f)) ok = false;
    if (!ok || !col_ok) continue;
    const float sc = own_scale(lk, r);
    const Vec<V> g = own_finish<V>(gg, lk.tmask, r, lk.out_col + c, c);
    acc.add_scaled(g, sc);
    cnt = cnt + 1.f;
  }
  // combine the row lanes of every column in a fixed order
  float* s_acc = smem;                 // [rpp][dim]
  float* s_cnt = smem + rpp * dim;     // [rpp]
  if (col_ok) acc.store(s_acc + static_cast<size_t>(rl) * dim + c);
  if (sub == 0) s_cnt[rl] = cnt;
  __syncthreads();
  // the partials cross workgroups INSIDE the launch.  A release fence here would write back the XCD L2's dirty lines - and
  // the tile workgroups next door are dirtying megabytes of table rows.  Instead the partials are stored and loaded with
  // system-scope atomics (write-through / cache-bypassing: MI355X_MICROARCH.md "valid forms"), the stores drained
  // (s_waitcnt) before the arrival ticket, itself a relaxed agent-scope atomic.
  float* mine = a.proj_partial + (static_cast<int64_t>(pj) * kProjParts + part) * (dim + 1);
  if (tid < dim) {
    float t = 0.f;
    for (int q = 0; q < rpp; ++q) t = t + s_acc[static_cast<size_t>(q) * dim + tid];
    __hip_atomic_store(mine + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (tid == kBlock - 1) {
    float t = 0.f;
    for (int q = 0; q < rpp; ++q) t = t + s_cnt[q];
    __hip_atomic_store(mine + dim, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached memory
  __syncthreads();                                     // ... and every wave's
  __shared__ int s_last;
  if (tid == 0) {
    const uint32_t t = __hip_atomic_fetch_add(a.proj_ticket + pj, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t % kProjParts) == kProjParts - 1 ? 1 : 0;
  }
  __syncthreads();
  if (dbg && tid == 0) dbg[1] = wall_clock64();
  if (!s_last) return;
  // every partial requested at once (one round trip), then combined from LDS in a fixed order
  const float* all = a.proj_partial + static_cast<int64_t>(pj) * kProjParts * (dim + 1);
  float* s_all = smem;  // [kProjParts][dim + 1] (the row-lane sums above are consumed)
  for (int i = tid; i < kProjParts * (dim + 1); i += kBlock)
    s_all[i] = __hip_atomic_load(all + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (tid >= G || !col_ok) return;
  float g[V];
#pragma unroll
  for (int j = 0; j < V; ++j) g[j] = 0.f;
  float n_valid = 0.f;
  for (int k = 0; k < kProjParts; ++k) {
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] = g[j] + s_all[k * (dim + 1) + c + j];
    n_valid = n_valid + s_all[k * (dim + 1) + dim];
  }
  if (n_valid == 0.f) return;  // no id of the batch read the row: TensorFlow's IndexedSlices has no entry for it
  const ReduceOut ro{0, nullptr, nullptr, nullptr, nullptr, 0};
  finish_run<V>(a.tab, ma.opt_kind, ma.hyper, ro, static_cast<uint32_t>(d.key_base), 0, sub, c, dim, g, a.aux);
  if (dbg && tid == 0) dbg[1] = wall_clock64();
}

#ifndef ER_OWN_WAVES
#define ER_OWN_WAVES 4
#endif
__device__ __forceinline__ void own_block(int bid, const OwnMulti& ma, float* smem) {
  if (bid < ma.start[ma.n]) {
    int i = 0;
    while (i + 1 < ma.n && bid >= ma.start[i + 1]) ++i;
    const OwnArgs& a = ma.a[i];
    if (a.V == 4) own_tile_body<4>(bid - ma.start[i], ma, a, smem);
    else own_tile_body<1>(bid - ma.start[i], ma, a, smem);
    return;
  }
  const int pb = bid - ma.start[ma.n];
  int i = 0;
  while (i + 1 < ma.n && pb >= ma.proj_start[i + 1]) ++i;
  const OwnArgs& a = ma.a[i];
  if (a.V == 4) own_proj_body<4>(pb - ma.proj_start[i], ma, a, smem);
  else own_proj_body<1>(pb - ma.proj_start[i], ma, a, smem);
}

__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ER_OWN_WAVES)))
emb_bwd_own_kernel(OwnMulti ma) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  own_block(blockIdx.x, ma, smem);
}

// The step's TAIL in one grid (er_emb_bwd_fused_wgrad): the weight gradients dW_l = x_l^T . dz_l of all dense layers (the
// grouped TN launch of er_gemm_grouped_f32: workgroups [0, n_gemm), first in the grid so that its XCD region keeps
// block % 8 = XCD) NEXT TO the embedding backward (gradient finish + tile reduce + row update: the workgroups behind
// them).  The two are independent - both need only the last input-gradient GEMM - and bound by different things: the
// contraction by the matrix cores and LDS of the ~2 workgroups per CU it launches, the row update by the latency of its
// random row records at 4 workgroups per CU.  Back to back they took 39 + 32 us of a 367 us DeepFM step; the round-4
// attempt to overlap them as two graph branches lost to launch-slot contention - one launch has no
// second launch to contend with.  Same bodies, same arithmetic: bit-identical to the two launches.
// n_own: the row update's workgroups; a workgroup behind them (er_emb_bwd_fused_tail with a loss-tail job) runs the step's
// scalar loss tail (er_dense_tail.h: the 1024 lanes of er_loss_tail on 256 threads, same sums, same order).
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ER_OWN_WAVES)))
emb_bwd_own_wgrad_kernel(OwnMulti ma, GroupedArgs ga, int n_gemm, int n_own, LossTailArgs lt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // >= the GEMM's two operand stages
  const int bid = blockIdx.x;
  if (bid < n_gemm) {
    const GroupedCoords c = grouped_coords(ga, bid);
    if (c.split < 0) return;
    gemm_f32_block<false, false>(ga.p[c.p], c.tile, c.split, smem, c.plain);
    return;
  }
  if (bid - n_gemm < n_own) own_block(bid - n_gemm, ma, smem);
  else loss_tail_body<kBlock>(lt, smem);
}

// The catch-up of a step's rows from the per-lookup lists of distinct keys the fused front's sort leaves (ukeys_seg:
// lookup l's keys at [ent_base[l], ent_base[l] + seg_count[l])): catch_up_closed_body's arithmetic, one lane group per
// potential key (those past their lookup's count leave at once), no cross-lookup scan and no route launch.  The last
// workgroup of a group takes its one-row tables, which the fused front keeps out of the sort.
struct CatchHeadsArgs {
  const uint32_t* ukeys_seg;
  const uint32_t* seg_count;
  const int64_t* ent_base;
  int n_lookups;
  int64_t n;
  RowUpdate tab;
  DecayAux aux;
  int dim, G, V;
  const uint32_t* extra_keys;  // one-row tables' keys (or nullptr)
  int n_extra;
  int cap_shift;               // >= 0: every lookup holds 2^cap_shift entries
};
struct CatchHeadsMulti {
  int n;
  int start[kMaxMulti + 1];
  const er_opt_hyper* hyper;
  CatchHeadsArgs a[kMaxMulti];
};

template <int V>
__device__ __forceinline__ void catch_up_row(const CatchHeadsArgs& a, uint32_t key, int c, int32_t t, float eps) {
  const int32_t s_begin = a.tab.ls(key) + 1;
  if (s_begin >= t) return;
  const int64_t off = a.tab.off(key, c);
  float var[V], m[V], v[V];
  ld_vec<V>(m, a.tab.m + off);
  ld_vec<V>(v, a.tab.v + off);
  bool live = false;
#pragma unroll
  for (int j = 0; j < V; ++j) live = live || (m[j] != 0.f) || (v[j] != 0.f);
  if (live) {
    ld_vec<V>(var, a.tab.var + off);
    replay_closed<V>(var, m, v, a.aux, s_begin, t, eps);
    st_vec<V>(a.tab.var + off, var);
    st_vec<V>(a.tab.m + off, m);
    st_vec<V>(a.tab.v + off, v);
  }
  if (c == 0) a.tab.ls(key) = t - 1;
}

template <int V>
__device__ __forceinline__ void catch_heads_body(int bid, const CatchHeadsArgs& a, const er_opt_hyper* __restrict__ hyper) {
  __shared__ int s_base[kOwnMaxLookups + 1], s_cnt[kOwnMaxLookups + 1];
  const int tid = threadIdx.x;
  const int ppb = kBlock / a.G;  // potential keys per workgroup
  const int c = (tid % a.G) * V;
  const int64_t n_blocks = ceil_div(a.n, ppb);
  if (bid >= n_blocks) {  // the group's one-row tables
    const int h = tid / a.G;
    if (h < a.n_extra && c < a.dim)
      catch_up_row<V>(a, a.extra_keys[h], c, static_cast<int32_t>(*a.tab.step_counter - 1), pin_scalar(hyper->eps));
    return;
  }
  // (the key is requested before anything else: the lookup search and the count check need no memory round trip after it)
  const int64_t p = static_cast<int64_t>(bid) * ppb + tid / a.G;
  const uint32_t key = p < a.n ? a.ukeys_seg[p] : kInvalidKey;
  if (a.cap_shift >= 0) {  // (uniform) equal power-of-two capacities: no table, no barrier
    if (p >= a.n || c >= a.dim) return;
    const int l = static_cast<int>(p >> a.cap_shift);
    if (p - (static_cast<int64_t>(l) << a.cap_shift) >= static_cast<int64_t>(a.seg_count[l])) return;
  } else {
    for (int i = tid; i <= a.n_lookups; i += kBlock) {
      s_base[i] = static_cast<int>(a.ent_base[i]);
      s_cnt[i] = i < a.n_lookups ? static_cast<int>(a.seg_count[i]) : 0;
    }
    __syncthreads();
    if (p >= a.n || c >= a.dim) return;
    int lo = 0, hi = a.n_lookups;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_base[mid] <= p) lo = mid; else hi = mid;
    }
    if (p - s_base[lo] >= s_cnt[lo]) return;
  }
  catch_up_row<V>(a, key, c, static_cast<int32_t>(*a.tab.step_counter - 1), pin_scalar(hyper->eps));
}

__global__ void __launch_bounds__(kBlock)
emb_catch_up_heads_kernel(CatchHeadsMulti ma) {
  int i = 0;
  while (i + 1 < ma.n && static_cast<int>(blockIdx.x) >= ma.start[i + 1]) ++i;
  const CatchHeadsArgs& a = ma.a[i];
  const int bid = blockIdx.x - ma.start[i];
  if (a.V == 4) catch_heads_body<4>(bid, a, ma.hyper);
  else catch_heads_body<1>(bid, a, ma.hyper);
}

// Row-wise optimizer on ready-made row sums (er_emb_apply_unique): one lane group per de-duplicated key.
template <int V>
__global__ void __launch_bounds__(kBlock)
emb_apply_unique_kernel(const uint32_t* __restrict__ ukeys, const float* __restrict__ ugrads, int ld,
                        const int32_t* __restrict__ n_unique, int64_t capacity, RowUpdate tab, int opt_kind,
                        const er_opt_hyper* __restrict__ hyper, int dim, int G) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / G;
  const int sub = static_cast<int>(threadIdx.x) % G;
  const int c = sub * V;
  if (i >= capacity || i >= *n_unique || c >= dim) return;
  const uint32_t key = ukeys[i];
  if (key == kInvalidKey) return;
  const float* gp = ugrads + i * ld + c;
  float g[V];
#pragma unroll
  for (int j = 0; j < V; ++j) g[j] = gp[j];
  const ReduceOut ro{0, nullptr, nullptr, nullptr, nullptr, 0};
  finish_run<V>(tab, opt_kind, hyper, ro, key, 0, sub, c, dim, g);
}

// the tile / fix kernels of several table groups in one grid each (see emb_catch_up_multi_kernel)
struct RunArgs {
  const uint32_t* skeys;
  const uint32_t* svals;
  const float* const* ent_gptr;
  const float* ent_scale;
  int64_t n;
  int dim, G, V, T, n_tiles;
  RowUpdate tab;
  ReduceOut ro;
  float* tile_first;
  float* tile_last;
  DecayAux aux;         // (fix launch of the fused step: see OwnArgs)
};
struct RunMulti {
  int n;
  int start[kMaxMulti + 1];
  int opt_kind;
  const er_opt_hyper* hyper;
  RunArgs a[kMaxMulti];
};

__global__ void __launch_bounds__(kBlock)
emb_bwd_tile_multi_kernel(RunMulti ma) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int i = 0;
  while (i + 1 < ma.n && static_cast<int>(blockIdx.x) >= ma.start[i + 1]) ++i;
  const RunArgs& a = ma.a[i];
  const int bid = blockIdx.x - ma.start[i];
  if (a.V == 4)
    tile_body<4>(bid, smem, a.skeys, a.svals, a.ent_gptr, a.ent_scale, a.n, a.dim, a.G, a.tab, ma.opt_kind, ma.hyper, a.ro,
                 a.tile_first, a.tile_last);
  else
    tile_body<1>(bid, smem, a.skeys, a.svals, a.ent_gptr, a.ent_scale, a.n, a.dim, a.G, a.tab, ma.opt_kind, ma.hyper, a.ro,
                 a.tile_first, a.tile_last);
}

__device__ __forceinline__ void tile_multi_block(int b, const RunMulti& ma, float* smem) {
  int i = 0;
  while (i + 1 < ma.n && b >= ma.start[i + 1]) ++i;
  const RunArgs& a = ma.a[i];
  const int bid = b - ma.start[i];
  if (a.V == 4)
    tile_body<4>(bid, smem, a.skeys, a.svals, a.ent_gptr, a.ent_scale, a.n, a.dim, a.G, a.tab, ma.opt_kind, ma.hyper, a.ro,
                 a.tile_first, a.tile_last);
  else
    tile_body<1>(bid, smem, a.skeys, a.svals, a.ent_gptr, a.ent_scale, a.n, a.dim, a.G, a.tab, ma.opt_kind, ma.hyper, a.ro,
                 a.tile_first, a.tile_last);
}

// The embedding-parallel requester's tail of the compute segment in ONE grid (er_emb_reduce_local_tail): the step's weight
// gradients (grouped TN contraction, workgroups [0, n_gemm)), the local gradient reductions of every table group - the
// replicated groups' dense row sums and the sharded groups' per-(owner, id) sums, emb_bwd_tile_multi_kernel's bodies - behind
// them, and the scalar loss tail as one more workgroup: the arrangement of the single-GPU step's fused tail
// (emb_bwd_own_wgrad_kernel) for the step that the 1/2/4/8-GPU metric runs.  Same bodies, same bits as the launches apart.
__global__ void __launch_bounds__(kBlock)
emb_reduce_local_wgrad_kernel(RunMulti ma, GroupedArgs ga, int n_gemm, int n_tile, LossTailArgs lt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  if (bid < n_gemm) {
    const GroupedCoords c = grouped_coords(ga, bid);
    if (c.split < 0) return;
    gemm_f32_block<false, false>(ga.p[c.p], c.tile, c.split, smem, c.plain);
    return;
  }
  if (bid - n_gemm < n_tile) tile_multi_block(bid - n_gemm, ma, smem);
  else loss_tail_body<kBlock>(lt, smem);
}

__device__ __forceinline__ void fix_block(int b, const RunMulti& ma) {
  int i = 0;
  while (i + 1 < ma.n && b >= ma.start[i + 1]) ++i;
  const RunArgs& a = ma.a[i];
  const int bid = b - ma.start[i];
  if (a.V == 4)
    fix_body<4>(bid, a.skeys, a.n, a.dim, a.G, a.T, a.n_tiles, a.tab, ma.opt_kind, ma.hyper, a.ro, a.tile_first,
                a.tile_last, a.aux);
  else
    fix_body<1>(bid, a.skeys, a.n, a.dim, a.G, a.T, a.n_tiles, a.tab, ma.opt_kind, ma.hyper, a.ro, a.tile_first,
                a.tile_last, a.aux);
}

__global__ void __launch_bounds__(kBlock)
emb_bwd_fix_multi_kernel(RunMulti ma) {
  fix_block(blockIdx.x, ma);
}

// ... and the tail's second launch: the runs that cross tile boundaries (workgroups [0, n_fix)) next to the split-K
// reduce of the weight gradients
__global__ void __launch_bounds__(kBlock)
emb_bwd_fix_reduce_kernel(RunMulti ma, GroupedReduceArgs ra, int n_fix) {
  const int b = blockIdx.x;
  if (b < n_fix) fix_block(b, ma);
  else splitk_reduce_grouped_block<4>(ra, b - n_fix);
}

// ... or next to the dense optimizer, which finishes the k-split weight gradients while it reads them (dense_opt_block)
__global__ void __launch_bounds__(kBlock)
emb_bwd_fix_opt_kernel(RunMulti ma, GroupedReduceArgs ra, DenseOptArgs da, int n_fix) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  if (b < n_fix) fix_block(b, ma);
  else dense_opt_block(da, ra, b - n_fix, red);
}

// Segmented sort: when every lookup of a group owns its own table (disjoint, increasing key ranges - the normal
// case: one embedding column per table) and has at most kSegSortMax entries, the global sort of the group's
// entries is the concatenation of the per-lookup sorts.  One workgroup sorts one lookup's entries (bitonic network
// over 64-bit (key << 32 | entry index) composites: ties broken by the entry index, i.e. exactly the result of
// the stable radix sort), replacing the 8-kernel global rocPRIM radix sort of the group by ONE launch.
//
// Each thread keeps E consecutive composites in registers.  A compare-exchange stage of stride j runs
//   j <  E        inside the thread,
//   j <  64 E     between lanes of one wavefront (__shfl_xor, no barrier),
//   j >= 64 E     through LDS, log2(E) stages per round trip: a thread gathers the E elements whose indices
//                 differ in the log2(E) stride bits of those stages and finishes them in registers.
// For P = 4096 that is 6 LDS round trips (12 barriers) instead of 78 barrier-separated passes.
constexpr int kSegSortMax = 8192;
constexpr int kSegSortMin = 256;

template <typename C>
__device__ __forceinline__ void seg_cmpx(C& a, C& b, bool up) {
  const bool sw = (a > b) == up;
  const C lo = sw ? b : a, hi = sw ? a : b;
  a = lo;
  b = hi;
}

// The NARROW composites (rel << log2 P | position) by a stable LSD radix sort on the `rel` bits alone - the positions start
// in ascending order, so a stable sort by rel IS the composite order, and the padding (all ones) sorts last.  5-bit digits:
// a 1 M-row table takes 4 passes, a one-row table (every entry `missing`) one - where the bitonic network below always runs
// its 78 compare-exchange stages (P = 4096), ~1300 VALU instructions per thread on ONE compute unit per lookup: the sort
// workgroups were VALU-issue-bound, 20 us, and the whole front launch waited for them (SQ_WAIT_INST_ANY / SQ_INSTS_VALU of
// emb_front_fwd_kernel, profiles/r06_s6_sq_counters_by_kernel.json).
// Per pass every wave ranks its own 64 E keys (striped: round e holds keys e * 64 + lane of the wave's chunk, so the key
// order is (wave, round, lane)): the lanes holding the same digit are found with 5 ballots, rank = the wave's count of that
// digit in earlier rounds (an LDS counter only the group's first lane advances) + the same-digit lanes below; one exclusive
// scan over the [digit][wave] counters places every (digit, wave) group; the keys are scattered into the other LDS buffer.
// Composites are distinct, so the result is the bitonic network's, bit for bit.
constexpr int kRadixBits = 5;
template <int E>
__device__ __forceinline__ void seg_radix_sort_narrow(uint32_t (&x)[E], uint32_t* buf /* [2 P] */, int P, int logP, uint32_t rel_max) {
  constexpr int NB = 1 << kRadixBits;
  constexpr int kMaxWaves = kSegSortMax / 8 / 64;
  __shared__ uint32_t cnt[NB * kMaxWaves];  // [digit][wave]
  __shared__ uint32_t wtot[kMaxWaves];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nW = static_cast<int>(blockDim.x >> 6);
  const int n_cnt = NB * nW;  // <= blockDim.x / 2
  uint32_t* A = buf;
  uint32_t* B = buf + P;
#pragma unroll
  for (int e = 0; e < E; ++e) A[t * E + e] = x[e];
  __syncthreads();
  const int key_bits = 32 - __clz(static_cast<int>(rel_max | 1u));
  const int chunk = wave * 64 * E;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint32_t key[E];
#pragma unroll
  for (int e = 0; e < E; ++e) key[e] = A[chunk + e * 64 + lane];
  for (int shift = logP; shift < logP + key_bits; shift += kRadixBits) {
    if (t < n_cnt) cnt[t] = 0u;
    __syncthreads();
    uint32_t rank[E], slot[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint32_t d = (key[e] >> shift) & static_cast<uint32_t>(NB - 1);
      unsigned long long m = ~0ull;
#pragma unroll
      for (int b = 0; b < kRadixBits; ++b) {
        const bool bit = ((d >> b) & 1u) != 0u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
      }
      const uint32_t lower = static_cast<uint32_t>(__popcll(m & lt_mask));
      const int sl = static_cast<int>(d) * nW + wave;
      const uint32_t prior = cnt[sl];
      rank[e] = prior + lower;
      slot[e] = static_cast<uint32_t>(sl);
      if (lower == 0u) cnt[sl] = prior + static_cast<uint32_t>(__popcll(m));
    }
    __syncthreads();
    // exclusive scan of the n_cnt counters in index order (digit-major): wave-inclusive scans, then the wave totals
    uint32_t v = 0u, inc = 0u;
    if (t < n_cnt) v = cnt[t];
    inc = v;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const uint32_t o = __shfl_up(inc, dd, 64);
      if (lane >= dd) inc += o;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t before = 0u;
    for (int w = 0; w < wave; ++w) before += wtot[w];
    if (t < n_cnt) cnt[t] = before + inc - v;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) B[cnt[slot[e]] + rank[e]] = key[e];
    __syncthreads();
    uint32_t* T = A; A = B; B = T;
#pragma unroll
    for (int e = 0; e < E; ++e) key[e] = A[chunk + e * 64 + lane];
  }
#pragma unroll
  for (int e = 0; e < E; ++e) x[e] = A[t * E + e];
}

// HEADS: also emit, per sorted position, the run-head flag and the number of heads before it INSIDE the lookup, and
// the lookup's number of distinct valid keys (emb_route_seg_kernel adds the lookups before it): the head-flag /
// exclusive-scan / count launches of the generic path disappear.
// NARROW: 32-bit composites (row inside the lookup's table << log2 P | position inside the lookup; a missing id
// sorts as row == rows) when (rows + 2) * P <= 2^32 for every lookup of the group - e.g. 1 M-row tables at P = 4096:
// half the shuffles, LDS traffic and registers of the 64-bit (group key << 32 | group entry) form.
// Routed keys (embedding-parallel requester: key = owner * stride + local_base[lookup] + id / W, rt.local_base != 0):
// inside one lookup that order is (owner, local row), which is all the all-to-all needs; NARROW then packs
// owner * shard_rows + local row, and with HEADS the kernel also counts the distinct keys per (lookup, owner)
// (seg_count[lookup * 64 + owner]) for emb_route_seg_routed_kernel.
// INLINE (the fused front of a single-GPU step, er_emb_front): the entries are built HERE from the lookups' ids (dense-mode
// lookups only: entry i of lookup lk is its output row i) instead of being read from keys_in - emb_bwd_build_kernel's rule
// for a valid id - and entries into ONE-ROW tables (the RawFeature projections: every valid id is row 0) sort as missing
// when skip_one_row is set: the fused backward reduces those tables by columns (emb_bwd_own_kernel), not through the sort.
template <int E, bool HEADS, bool NARROW, bool INLINE>
__device__ __forceinline__ void seg_sort_body(int lk, const uint32_t* __restrict__ keys_in, const int64_t* __restrict__ ent_base,
                        const er_lookup_desc* __restrict__ descs, int P, Route rt, uint32_t* __restrict__ keys_out,
                        uint32_t* __restrict__ vals_out, uint32_t* __restrict__ flags_out,
                        uint32_t* __restrict__ hidx_out, uint32_t* __restrict__ seg_count, int skip_one_row,
                        unsigned long long* sk_raw, uint32_t* __restrict__ ukeys_seg = nullptr,
                        int64_t n_limit = INT64_MAX) {  // (entries at positions >= n_limit do not exist: chunks of an owner group)
  typedef typename std::conditional<NARROW, uint32_t, unsigned long long>::type C;
  C* sk = reinterpret_cast<C*>(sk_raw);
  constexpr int L = E == 8 ? 3 : (E == 4 ? 2 : 1);
  static_assert(E == 2 || E == 4 || E == 8, "E");
  const int t = threadIdx.x;
  const int i0 = t * E;
  const int64_t base = ent_base[lk];
  const int64_t end = ent_base[lk + 1] < n_limit ? ent_base[lk + 1] : n_limit;
  const int cnt = end > base ? static_cast<int>(end - base) : 0;
  const int logP = __ffs(P) - 1;
  const bool routed = rt.local_base != nullptr;
  const uint32_t stride = static_cast<uint32_t>(rt.shard_stride);
  // NARROW: `rel` in [0, rows) enumerates the lookup's keys in key order; rel == rows is a missing id
  uint32_t key_base = 0u, rows = 0u, srows = 1u;
  if (NARROW) {
    const er_lookup_desc& d = descs[lk];
    if (routed) {
      srows = static_cast<uint32_t>((d.rows + rt.world - 1) / rt.world);  // rows of one rank's shard of this table
      rows = srows * static_cast<uint32_t>(rt.world);
      key_base = static_cast<uint32_t>(rt.local_base[lk]);
    } else {
      rows = static_cast<uint32_t>(d.rows);
      key_base = static_cast<uint32_t>(d.key_base);
    }
  }
  auto key_of = [&](C c) -> uint32_t {
    if (NARROW) {
      const uint32_t rel = static_cast<uint32_t>(c >> logP);
      if (rel >= rows) return kInvalidKey;  // (padding decodes as invalid too)
      if (!routed) return rel + key_base;
      const uint32_t owner = rel / srows;
      return owner * stride + key_base + (rel - owner * srows);
    }
    return static_cast<uint32_t>(static_cast<unsigned long long>(c) >> 32);
  };
  C x[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = i0 + e;
    if (i >= cnt) {
      x[e] = static_cast<C>(~0ull);
    } else if (INLINE) {
      const er_lookup_desc& d = descs[lk];
      const int64_t id = d.ids[i];
      bool ok = !(id < 0 || id >= d.rows);
      if (d.weights != nullptr && d.combiner != ER_COMBINER_SUM && !(d.weights[i] > 0.f)) ok = false;
      if (skip_one_row && d.rows == 1) ok = false;
      if (NARROW) {
        const uint32_t rel = ok ? static_cast<uint32_t>(id) : static_cast<uint32_t>(d.rows);
        x[e] = static_cast<C>((rel << logP) | static_cast<uint32_t>(i));
      } else {
        const uint32_t key = ok ? static_cast<uint32_t>(d.key_base + id) : kInvalidKey;
        x[e] = static_cast<C>((static_cast<unsigned long long>(key) << 32) | static_cast<unsigned>(base + i));
      }
    } else if (NARROW) {
      const uint32_t key = keys_in[base + i];
      uint32_t rel = rows;
      if (key != kInvalidKey) {
        if (routed) {
          const uint32_t owner = key / stride;
          rel = owner * srows + (key - owner * stride - key_base);
        } else {
          rel = key - key_base;
        }
      }
      x[e] = static_cast<C>((rel << logP) | static_cast<uint32_t>(i));
    } else {
      x[e] = static_cast<C>((static_cast<unsigned long long>(keys_in[base + i]) << 32) | static_cast<unsigned>(base + i));
    }
  }
  if constexpr (NARROW) {
    seg_radix_sort_narrow<E>(x, reinterpret_cast<uint32_t*>(sk_raw), P, logP, rows);
  } else
  for (int k = 2; k <= P; k <<= 1) {
    int j = k >> 1;
    bool in_lds = false;
    while (j >= 64 * E) {
      if (!in_lds) {
#pragma unroll
        for (int e = 0; e < E; ++e) sk[i0 + e] = x[e];
        __syncthreads();
      }
      const int b = __ffs(j) - 1, lb = b - L + 1;
      const int bs = ((t >> lb) << (b + 1)) | (t & ((1 << lb) - 1));
      const bool up = (bs & k) == 0;
      C y[E];
#pragma unroll
      for (int c = 0; c < E; ++c) y[c] = sk[bs + (c << lb)];
#pragma unroll
      for (int jj = E / 2; jj > 0; jj >>= 1)
#pragma unroll
        for (int c = 0; c < E; ++c)
          if ((c & jj) == 0) seg_cmpx(y[c], y[c | jj], up);
#pragma unroll
      for (int c = 0; c < E; ++c) sk[bs + (c << lb)] = y[c];
      __syncthreads();
      in_lds = true;
      j >>= L;
    }
    if (in_lds) {
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = sk[i0 + e];
    }
    if (j >= E) {
      const bool up = (i0 & k) == 0;  // k >= 2 j >= 2 E: bit k lies in the thread part of the index
      for (; j >= E; j >>= 1) {
        const int lj = j / E;
        const bool keep_min = ((t & lj) == 0) == up;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const C o = __shfl_xor(x[e], lj);
          const C mn = x[e] < o ? x[e] : o, mx = x[e] < o ? o : x[e];
          x[e] = keep_min ? mn : mx;
        }
      }
    }
#pragma unroll
    for (int jj = E / 2; jj > 0; jj >>= 1) {
      if (jj <= j) {
#pragma unroll
        for (int e = 0; e < E; ++e)
          if ((e & jj) == 0) seg_cmpx(x[e], x[e | jj], ((i0 + e) & k) == 0);
      }
    }
  }
  // every sorted composite's key (and, routed, its owner) once: the routed form costs two integer divisions per call, and
  // the stores, the head flags, the distinct-key list and the per-owner counts below each asked for it again
  uint32_t kk[E], ow[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    kk[e] = key_of(x[e]);
    ow[e] = routed ? kk[e] / stride : 0u;
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = i0 + e;
    if (i < cnt) {
      keys_out[base + i] = kk[e];
      vals_out[base + i] = NARROW ? static_cast<uint32_t>(base) + (static_cast<uint32_t>(x[e]) & static_cast<uint32_t>(P - 1))
                                  : static_cast<uint32_t>(static_cast<unsigned long long>(x[e]) & 0xFFFFFFFFu);
    }
  }
  if (HEADS) {
    __shared__ uint32_t wave_sum[kSegSortMax / 8 / 64];
    uint32_t* k32 = reinterpret_cast<uint32_t*>(sk_raw);  // [P] sorted keys, to look one position back
    __syncthreads();                                      // every thread has taken its composites out of sk
#pragma unroll
    for (int e = 0; e < E; ++e) k32[i0 + e] = kk[e];
    __syncthreads();
    uint32_t prev = i0 > 0 ? k32[i0 - 1] : kInvalidKey;
    uint32_t f[E], c = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint32_t key = kk[e];
      f[e] = (key != kInvalidKey && (i0 + e == 0 || prev != key)) ? 1u : 0u;
      c += f[e];
      prev = key;
    }
    // exclusive scan of c over the workgroup: inclusive scan inside the wave, then the wave totals
    const int lane = t & 63, wave = t >> 6, n_waves = blockDim.x >> 6;
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < n_waves; ++w) {
      const uint32_t ws = wave_sum[w];
      if (w < wave) before += ws;
      total += ws;
    }
    uint32_t run = before + inc - c;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = i0 + e;
      if (i < cnt) {
        flags_out[base + i] = f[e];
        hidx_out[base + i] = run;
        // (the fused front: the lookup's distinct keys, compact inside its own segment - er_emb_front's catch-up reads
        // [base, base + seg_count[lookup]) of it, no cross-lookup scan and no launch for one)
        if (ukeys_seg != nullptr && f[e]) ukeys_seg[base + run] = kk[e];
      }
      run += f[e];
    }
    if (!routed) {
      if (t == 0) seg_count[lk] = total;
    } else {
      __shared__ uint32_t owner_cnt[64];
      if (t < 64) owner_cnt[t] = 0u;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (f[e]) atomicAdd(&owner_cnt[ow[e]], 1u);  // integer: exact, order-independent
      __syncthreads();
      if (t < 64) seg_count[lk * 64 + t] = owner_cnt[t];
    }
  }
}

template <int E, bool HEADS, bool NARROW>
__global__ void __launch_bounds__(kSegSortMax / 8)
emb_segment_sort_kernel(const uint32_t* __restrict__ keys_in, const int64_t* __restrict__ ent_base,
                        const er_lookup_desc* __restrict__ descs, int P, Route rt, uint32_t* __restrict__ keys_out,
                        uint32_t* __restrict__ vals_out, uint32_t* __restrict__ flags_out,
                        uint32_t* __restrict__ hidx_out, uint32_t* __restrict__ seg_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk_raw[];  // [P] composites
  seg_sort_body<E, HEADS, NARROW, false>(blockIdx.x, keys_in, ent_base, descs, P, rt, keys_out, vals_out, flags_out, hidx_out,
                                         seg_count, 0, sk_raw);
}

// The front of a single-GPU training step in ONE launch (er_emb_front): workgroup lk < n_lookups builds, sorts and
// head-flags lookup lk's entries (seg_sort_body, INLINE); the workgroups behind them compute the closed-form replay's
// per-launch table A for lag 1 (decay_tables_kernel's job: the step counter is stable here - the prologue that increments
// it is an earlier launch - and the catch-up that reads A is a later one).
template <int E, bool NARROW>
__global__ void __launch_bounds__(kSegSortMax / 8)
emb_front_sort_kernel(const int64_t* __restrict__ ent_base, const er_lookup_desc* __restrict__ descs, int n_lookups, int P,
                      uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t* __restrict__ flags_out,
                      uint32_t* __restrict__ hidx_out, uint32_t* __restrict__ seg_count, int skip_one_row, DecayTabDev tabs,
                      const float* __restrict__ hist, const int64_t* __restrict__ counter, uint32_t* __restrict__ ukeys_seg) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk_raw[];  // [P] composites
  if (static_cast<int>(blockIdx.x) >= n_lookups) {
    const int k = (static_cast<int>(blockIdx.x) - n_lookups) * static_cast<int>(blockDim.x >> 6) + static_cast<int>(threadIdx.x >> 6) + 1;
    if (k > tabs.K) return;
    const int64_t s_end = *counter - 1;  // lag 1: the rows are brought to the step before this one
    const float mine = decay_sum_wave(tabs, hist, s_end - 1 - k, k);
    const int lane = threadIdx.x & 63;
    float* A = tabs.A + static_cast<int64_t>(kDecayKMax) * kDecayLd;  // (lag 1's table: decay_aux_for)
    if (lane < kDecayLd) A[static_cast<int64_t>(k - 1) * kDecayLd + lane] = mine;
    return;
  }
  seg_sort_body<E, true, NARROW, true>(blockIdx.x, nullptr, ent_base, descs, P, Route{1, 0, nullptr}, keys_out, vals_out,
                                       flags_out, hidx_out, seg_count, skip_one_row, sk_raw, ukeys_seg);
}

// The sort of a single-GPU step and its LOOKUP in one launch (er_emb_front_fwd).  The lookup does not need the sort - it
// reads the ids - and the sort keeps only n_lookups workgroups (one CU each) busy for ~20 us: the lookup's blocks, kBlock
// lanes each, fill the rest of the chip meanwhile - blockDim / kBlock of them per workgroup behind the sort's.  Possible
// once nothing in the launch produces what another part of it consumes: the lag-1 replay table the lazy lookup reads is
// built one launch earlier, by the step prologue (er_decay_tables_set_prologue_build).
template <int E, bool NARROW>
__global__ void __launch_bounds__(kSegSortMax / 8)
emb_front_fwd_kernel(const int64_t* __restrict__ ent_base, const er_lookup_desc* __restrict__ descs, int n_lookups, int P,
                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t* __restrict__ flags_out,
                     uint32_t* __restrict__ hidx_out, uint32_t* __restrict__ seg_count, int skip_one_row,
                     uint32_t* __restrict__ ukeys_seg, const er_lookup_desc* __restrict__ fwd_descs,
                     const int32_t* __restrict__ fwd_blk_start, int fwd_lookups, int fwd_blocks,
                     float* __restrict__ sumsq_partials, FwdLazy lz) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk_raw[];  // [P] composites (sort workgroups)
  if (static_cast<int>(blockIdx.x) < n_lookups) {
    seg_sort_body<E, true, NARROW, true>(blockIdx.x, nullptr, ent_base, descs, P, Route{1, 0, nullptr}, keys_out, vals_out,
                                         flags_out, hidx_out, seg_count, skip_one_row, sk_raw, ukeys_seg);
    return;
  }
  __shared__ float red[kSegSortMax / 8 / 64];
  const int per_wg = static_cast<int>(blockDim.x) / kBlock;
  const int vb = (static_cast<int>(blockIdx.x) - n_lookups) * per_wg + static_cast<int>(threadIdx.x) / kBlock;
  const int vt = static_cast<int>(threadIdx.x) % kBlock;
  float ss = 0.f;
  if (vb < fwd_blocks) ss = fwd_rows<true>(vb, vt, fwd_descs, fwd_blk_start, fwd_lookups, &lz);
  if (sumsq_partials) {  // block_sum_256 per kBlock lanes: wave sums, then (w0 + w1) + (w2 + w3)
    ss = wave_sum(ss);
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wid] = ss;
    __syncthreads();
    if (vt == 0 && vb < fwd_blocks) {
      const int w0 = (static_cast<int>(threadIdx.x) / kBlock) * 4;
      sumsq_partials[vb] = (red[w0] + red[w0 + 1]) + (red[w0 + 2] + red[w0 + 3]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The device-wide sort of a group whose lookups share a table or hold more than kSegSortMax entries (DIN's [B, L]
// sequence lookups, MMoE's tag lists, dlrm_on_criteo_parquet_ep_v2-style shared tables): hand-written, no library.
//   1. emb_chunk_sort_kernel: the entries in chunks of P <= kSegSortMax, one workgroup each, by the bitonic network of the
//      per-lookup sort (64-bit composites key << 32 | entry: the order of a stable sort by key);
//   2. emb_merge_pass_kernel, log2(chunks) times: adjacent sorted runs merged pairwise, one workgroup per kMergeTile
//      outputs - two merge-path searches (where the tile's diagonals cut the two runs), the tile's inputs staged in LDS,
//      then every lane finds its own kMergeVt outputs by a merge-path search in LDS and merges them serially;
//   3. emb_head_scan_kernel / emb_head_offsets_kernel: run-head flags and their exclusive scan (the index of every
//      entry's run among the distinct keys) in two levels - per tile, then one workgroup over the tile totals.
// (rounds 1-4: rocPRIM's radix_sort_pairs + exclusive_scan, 18 launches and ~100 us per DIN / MMoE step.)
// ------------------------------------------------------------------------------------------------
template <int E>
__global__ void __launch_bounds__(kSegSortMax / 8)
emb_chunk_sort_kernel(const uint32_t* __restrict__ keys_in, const int64_t* __restrict__ chunk_base, int P, int64_t n,
                      uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk_raw[];
  seg_sort_body<E, false, false, false>(blockIdx.x, keys_in, chunk_base, nullptr, P, Route{1, 0, nullptr}, keys_out, vals_out,
                                        nullptr, nullptr, nullptr, 0, sk_raw, nullptr, n);
}

// entries per chunk of the device-wide sort: 4 composites per thread (the 8-per-thread network that 8192-entry chunks
// need ran 54 us for DIN's 25 chunks against ~20 us for 50 chunks of 4096 + one more 6 us merge pass)
constexpr int kChunkSortMax = 4096;
constexpr int kMergeVt = 8;                      // outputs per lane
constexpr int kMergeTile = kBlock * kMergeVt;    // outputs per workgroup

__device__ __forceinline__ unsigned long long merge_comp(const uint32_t* __restrict__ k, const uint32_t* __restrict__ v, int64_t i) {
  return (static_cast<unsigned long long>(k[i]) << 32) | v[i];
}

// the number of elements the first `d` outputs of merge(A, B) take from A (composites are distinct: no ties)
template <typename GetA, typename GetB>
__device__ __forceinline__ int merge_path(int d, int na, int nb, GetA a_at, GetB b_at) {
  int lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a_at(mid) < b_at(d - 1 - mid)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(kBlock)
emb_merge_pass_kernel(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin, int64_t n, int64_t run,
                      uint32_t* __restrict__ kout, uint32_t* __restrict__ vout) {
  __shared__ unsigned long long sm[kMergeTile];
  __shared__ int s_lo[2], s_hi[2];
  const int tid = threadIdx.x;
  const int64_t o0 = static_cast<int64_t>(blockIdx.x) * kMergeTile;
  const int64_t o1 = o0 + kMergeTile < n ? o0 + kMergeTile : n;
  const int64_t pair0 = (o0 / (2 * run)) * (2 * run);           // start of this tile's pair of runs
  const int64_t a_beg = pair0;
  const int64_t a_end = a_beg + run < n ? a_beg + run : n;
  const int64_t b_end = a_end + run < n ? a_end + run : n;
  const int na = static_cast<int>(a_end - a_beg), nb = static_cast<int>(b_end - a_end);
  // Where the tile's two diagonals cut the runs: the merge-path search over global memory, 128 probes per step and
  // diagonal (one half of the workgroup each) - 3 steps for runs of 2^17 entries where a lane's binary search takes 17
  // dependent round trips (12 -> ~6 us per pass).  The predicate A[a] < B[d - 1 - a] is true below the cut, false from it on.
  {
    constexpr int K = kBlock / 2;
    const int half = tid / K, t = tid % K;
    const int d = static_cast<int>((half == 0 ? o0 : o1) - pair0);
    if (t == 0) {
      s_lo[half] = d > nb ? d - nb : 0;
      s_hi[half] = d < na ? d : na;
    }
    __syncthreads();
    for (;;) {
      const int lo = s_lo[half], hi = s_hi[half];
      const int r = hi - lo;
      if (!__syncthreads_or(r > 0)) break;  // (also: every lane has read lo / hi before anyone moves them)
      if (r > 0) {
        const int p = lo + static_cast<int>((static_cast<int64_t>(r) * (t + 1)) / (K + 1));  // in [lo, hi)
        const bool below = merge_comp(kin, vin, a_beg + p) < merge_comp(kin, vin, a_end + (d - 1 - p));
        if (below) atomicMax(&s_lo[half], p + 1); else atomicMin(&s_hi[half], p);
      }
      __syncthreads();
    }
  }
  const int a0 = s_lo[0], a1 = s_lo[1];
  const int b0 = static_cast<int>(o0 - pair0) - a0, b1 = static_cast<int>(o1 - pair0) - a1;
  const int ta = a1 - a0, tb = b1 - b0;  // the tile's inputs: ta + tb == o1 - o0
  const int total = ta + tb;
  {
    // the tile's kMergeVt inputs per lane requested TOGETHER (slot i of the LDS tile comes from A below ta, from B behind it;
    // a slot past the end reads the last input and is not stored): the two loops `for (i = tid; i < ta; i += kBlock)` this
    // replaces had data-dependent trip counts, were not unrolled, and waited for each iteration's two loads before the next -
    // up to nine dependent round trips per pass and workgroup, seven passes per DIN / MMoE step
    uint32_t kv[kMergeVt], vv[kMergeVt];
#pragma unroll
    for (int j = 0; j < kMergeVt; ++j) {
      int i = tid + j * kBlock;
      i = i < total ? i : total - 1;
      i = i < 0 ? 0 : i;
      const int64_t at = i < ta ? a_beg + a0 + i : a_end + b0 + (i - ta);
      kv[j] = kin[at];
      vv[j] = vin[at];
    }
#pragma unroll
    for (int j = 0; j < kMergeVt; ++j) {
      const int i = tid + j * kBlock;
      if (i < total) sm[i] = (static_cast<unsigned long long>(kv[j]) << 32) | vv[j];
    }
  }
  __syncthreads();
  const int d = tid * kMergeVt < total ? tid * kMergeVt : total;
  int ia = merge_path(d, ta, tb, [&](int i) { return sm[i]; }, [&](int i) { return sm[ta + i]; });
  int ib = d - ia;
#pragma unroll
  for (int j = 0; j < kMergeVt; ++j) {
    const int64_t o = o0 + d + j;
    if (d + j >= total) break;
    const bool take_a = ib >= tb || (ia < ta && sm[ia] < sm[ta + ib]);
    const unsigned long long c = take_a ? sm[ia] : sm[ta + ib];
    ia += take_a ? 1 : 0;
    ib += take_a ? 0 : 1;
    kout[o] = static_cast<uint32_t>(c >> 32);
    vout[o] = static_cast<uint32_t>(c & 0xFFFFFFFFu);
  }
}

// run-head flags of the sorted keys + their exclusive scan inside tiles of kMergeTile positions; tile_sum[t] = heads in tile t
__global__ void __launch_bounds__(kBlock)
emb_head_scan_kernel(const uint32_t* __restrict__ skeys, int64_t n, uint32_t* __restrict__ flags,
                     uint32_t* __restrict__ head_index, uint32_t* __restrict__ tile_sum) {
  __shared__ uint32_t wsum[kBlock / 64];
  const int tid = threadIdx.x;
  const int64_t p0 = static_cast<int64_t>(blockIdx.x) * kMergeTile + static_cast<int64_t>(tid) * kMergeVt;
  uint32_t f[kMergeVt], c = 0;
  uint32_t prev = (p0 > 0 && p0 <= n) ? skeys[p0 - 1] : kInvalidKey;
#pragma unroll
  for (int j = 0; j < kMergeVt; ++j) {
    const int64_t p = p0 + j;
    const uint32_t key = p < n ? skeys[p] : kInvalidKey;
    f[j] = (p < n && key != kInvalidKey && (p == 0 || prev != key)) ? 1u : 0u;
    c += f[j];
    prev = key;
  }
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t inc = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    const uint32_t ws = wsum[w];
    if (w < wave) before += ws;
    total += ws;
  }
  uint32_t run = before + inc - c;
#pragma unroll
  for (int j = 0; j < kMergeVt; ++j) {
    const int64_t p = p0 + j;
    if (p < n) {
      flags[p] = f[j];
      head_index[p] = run;
    }
    run += f[j];
  }
  if (tid == 0) tile_sum[blockIdx.x] = total;
}

// head_index += (heads of the tiles before); *n_unique = all heads.  Every workgroup scans the tile totals itself (a few
// hundred words from L2) instead of waiting for a scan launch: two launches in all.
__global__ void __launch_bounds__(kBlock)
emb_head_offsets_kernel(const uint32_t* __restrict__ tile_sum, int n_tiles, int64_t n, uint32_t* __restrict__ head_index,
                        int32_t* __restrict__ n_unique) {
  __shared__ uint32_t red[kBlock / 64];
  const int tid = threadIdx.x;
  const int t = blockIdx.x;
  uint32_t before = 0, all = 0;
  for (int i = tid; i < n_tiles; i += kBlock) {
    const uint32_t v = tile_sum[i];
    all += v;
    if (i < t) before += v;
  }
  // (two sums over the workgroup, integer: exact in any order)
  uint32_t x = before;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d, 64);
  if ((tid & 63) == 0) red[tid >> 6] = x;
  __syncthreads();
  uint32_t off = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) off += red[w];
  __syncthreads();
  if (t == 0 && n_unique != nullptr) {
    uint32_t y = all;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) y += __shfl_xor(y, d, 64);
    if ((tid & 63) == 0) red[tid >> 6] = y;
    __syncthreads();
    if (tid == 0) {
      uint32_t tot = 0;
      for (int w = 0; w < kBlock / 64; ++w) tot += red[w];
      *n_unique = static_cast<int32_t>(tot);
    }
  }
  if (off == 0) return;
  const int64_t p0 = static_cast<int64_t>(t) * kMergeTile;
  for (int i = tid; i < kMergeTile; i += kBlock) {
    const int64_t p = p0 + i;
    if (p < n) head_index[p] += off;
  }
}

// Routed counterpart of emb_route_seg_kernel.  The de-duplicated keys must come out grouped by owner (the
// all-to-all send order) and, inside an owner, by lookup then row - which is the ascending order of the routed
// key.  A run of lookup l owned by w gets the index
//   u = (keys of owners < w) + (keys of owner w in lookups < l) + (its rank among lookup l's keys of owner w),
// all from the 64-column count matrix the sort kernel left.  head_index[p] is stored as u + 1 - flag[p] so that the
// reduction kernels' `head_index + flag - 1` yields u at every entry of the run.
__global__ void __launch_bounds__(kBlock)
emb_route_seg_routed_kernel(const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ svals,
                            const uint32_t* __restrict__ flags, uint32_t* __restrict__ head_index,
                            const int64_t* __restrict__ ent_base, const uint32_t* __restrict__ seg_count, int n_lookups,
                            int world, uint32_t stride, uint32_t* __restrict__ unique_keys, int64_t* __restrict__ uidx,
                            int32_t* __restrict__ n_unique, int32_t* __restrict__ owner_counts, uint32_t peer_cap,
                            uint32_t peer_hdr, int32_t* __restrict__ overflow) {
  __shared__ uint32_t s_total[64], s_before[64], s_row[64], s_obase[64], s_lbase[64];
  const int l = blockIdx.y;
  const int tid = threadIdx.x;
  if (tid < 64) {
    uint32_t tot = 0, before = 0;
    if (tid < world)
      for (int j = 0; j < n_lookups; ++j) {
        const uint32_t c = seg_count[j * 64 + tid];
        if (j < l) before += c;
        tot += c;
      }
    s_total[tid] = tot;
    s_before[tid] = before;
    s_row[tid] = tid < world ? seg_count[l * 64 + tid] : 0u;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t a = 0, b = 0;
    bool over = false;
    for (int w = 0; w < world; ++w) {
      s_obase[w] = peer_cap ? static_cast<uint32_t>(w) * peer_cap : a;  // padded: owner w's keys start at w * peer_cap
      s_lbase[w] = b;
      a += s_total[w];
      b += s_row[w];
      over = over || (peer_cap && s_total[w] > peer_cap);
    }
    if (l == n_lookups - 1 && blockIdx.x == 0) {
      *n_unique = static_cast<int32_t>(a);
      if (over) *overflow = 1;
    }
  }
  __syncthreads();
  if (l == n_lookups - 1 && blockIdx.x == 0 && tid < world) {
    if (owner_counts) owner_counts[tid] = static_cast<int32_t>(s_total[tid]);
    // padded with a header: the count travels in front of the owner's keys (one all-to-all for both)
    if (peer_hdr) unique_keys[static_cast<uint32_t>(tid) * (peer_cap + 1u)] = s_total[tid];
  }
  const int64_t base = ent_base[l];
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + tid;
  if (i >= ent_base[l + 1] - base) return;
  const int64_t p = base + i;
  const uint32_t key = skeys[p];
  const uint32_t j = svals[p];
  if (key == kInvalidKey) {
    if (uidx) uidx[j] = -1;
    return;
  }
  const uint32_t f = flags[p];
  const uint32_t w = key / stride;
  const uint32_t run = head_index[p] + f - 1u;  // index of the run among the lookup's runs
  uint32_t r = s_before[w] + (run - s_lbase[w]);  // index of the key among owner w's keys
  const bool fits = !peer_cap || r < peer_cap;
  if (!fits) r = peer_cap - 1u;                    // (*overflow is set: the step's results are void, stay in bounds)
  const uint32_t u = s_obase[w] + r;
  head_index[p] = u + 1u - f;
  if (uidx) uidx[j] = fits ? static_cast<int64_t>(u) : -1;
  if (f && fits) unique_keys[peer_hdr ? w * (peer_cap + 1u) + 1u + r : u] = key;
}

// Second half of the fused route of the segmented path: add the distinct-key counts of the lookups before this one
// (head_index becomes the global exclusive scan), write the de-duplicated key list, the per-entry index into it and
// the total.  grid = (ceil(P / 256), lookups).
__global__ void __launch_bounds__(kBlock)
emb_route_seg_kernel(const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ svals,
                     const uint32_t* __restrict__ flags, uint32_t* __restrict__ head_index,
                     const int64_t* __restrict__ ent_base, const uint32_t* __restrict__ seg_count, int n_lookups,
                     uint32_t* __restrict__ unique_keys, int64_t* __restrict__ uidx, int32_t* __restrict__ n_unique) {
  __shared__ uint32_t red[kBlock / 64];
  const int l = blockIdx.y;
  uint32_t acc = 0;
  for (int j = threadIdx.x; j < l; j += kBlock) acc += seg_count[j];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  uint32_t prefix = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) prefix += red[w];
  if (l == n_lookups - 1 && blockIdx.x == 0 && threadIdx.x == 0) *n_unique = static_cast<int32_t>(prefix + seg_count[l]);
  const int64_t base = ent_base[l];
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= ent_base[l + 1] - base) return;
  const int64_t p = base + i;
  const uint32_t h = head_index[p] + prefix;
  head_index[p] = h;
  const uint32_t key = skeys[p];
  const uint32_t j = svals[p];
  if (key == kInvalidKey) {
    if (uidx) uidx[j] = -1;
    return;
  }
  const uint32_t f = flags[p];
  const uint32_t u = h + f - 1u;
  if (uidx) uidx[j] = static_cast<int64_t>(u);
  if (f) unique_keys[u] = key;
}

__global__ void __launch_bounds__(kBlock)
emb_head_flag_kernel(const uint32_t* __restrict__ skeys, int64_t n, uint32_t* __restrict__ flags) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= n) return;
  const uint32_t key = skeys[p];
  flags[p] = (key != kInvalidKey && (p == 0 || skeys[p - 1] != key)) ? 1u : 0u;
}

__global__ void emb_count_unique_kernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ head_index,
                                        int64_t n, int32_t* __restrict__ n_unique) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *n_unique = static_cast<int32_t>(head_index[n - 1] + flags[n - 1]);
}

// Embedding-parallel routing: per sorted position p the index of its run among the valid runs
// (= index into the de-duplicated key list) is head_index[p] + flags[p] - 1.  Writes the unique keys
// (ascending = grouped by owner) and, per ENTRY (source order), the index of its unique key, or -1.
__global__ void __launch_bounds__(kBlock)
emb_route_kernel(const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ svals,
                 const uint32_t* __restrict__ flags, const uint32_t* __restrict__ head_index, int64_t n,
                 uint32_t* __restrict__ unique_keys, int64_t* __restrict__ uidx) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= n) return;
  const uint32_t key = skeys[p];
  const uint32_t j = svals[p];
  if (key == kInvalidKey) {
    if (uidx) uidx[j] = -1;
    return;
  }
  const uint32_t u = head_index[p] + flags[p] - 1u;
  if (uidx) uidx[j] = static_cast<int64_t>(u);
  if (flags[p]) unique_keys[u] = key;
}

// Fixed-capacity layout after the device-wide sort (lookups that share a table, or > 8192 entries: the per-lookup
// sort does not apply): the ascending unique keys are already grouped by owner, so key number u of owner w moves to
// slot w * cap + (u - keys of the owners before w); head_index and the per-entry index follow.  compact_keys is the
// de-duplicated list emb_route_kernel wrote (a scratch buffer), counts[w] come from emb_owner_counts_kernel.
__global__ void __launch_bounds__(kBlock)
emb_route_pad_kernel(const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ svals,
                     const uint32_t* __restrict__ flags, uint32_t* __restrict__ head_index, int64_t n,
                     const int32_t* __restrict__ counts, int world, uint32_t stride, uint32_t cap, uint32_t hdr,
                     uint32_t* __restrict__ unique_keys, int64_t* __restrict__ uidx, int32_t* __restrict__ owner_counts,
                     int32_t* __restrict__ overflow) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p < world) {
    const int32_t c = counts[p];
    if (hdr) unique_keys[static_cast<uint32_t>(p) * (cap + 1u)] = static_cast<uint32_t>(c);
    if (owner_counts) owner_counts[p] = c;
    if (static_cast<uint32_t>(c) > cap) *overflow = 1;
  }
  if (p >= n) return;
  const uint32_t key = skeys[p];
  if (key == kInvalidKey) return;  // (emb_route_kernel left uidx = -1)
  const uint32_t w = key / stride;
  uint32_t obase = 0;
  for (uint32_t q = 0; q < w; ++q) obase += static_cast<uint32_t>(counts[q]);
  const uint32_t f = flags[p];
  uint32_t r = head_index[p] + f - 1u - obase;
  const bool fits = r < cap;
  if (!fits) r = cap - 1u;
  const uint32_t u = w * cap + r;
  head_index[p] = u + 1u - f;
  if (uidx) uidx[svals[p]] = fits ? static_cast<int64_t>(u) : -1;
  if (f && fits) unique_keys[w * (cap + hdr) + hdr + r] = key;
}

// counts[w] = number of unique keys owned by rank w (keys in [w*stride, (w+1)*stride)).
__global__ void emb_owner_counts_kernel(const uint32_t* __restrict__ unique_keys, const int32_t* __restrict__ n_unique,
                                        int world, int64_t stride, int32_t* __restrict__ counts) {
  const int w = threadIdx.x;
  if (w >= world) return;
  const int n = *n_unique;
  auto lower = [&](int64_t v) {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (static_cast<int64_t>(unique_keys[mid]) < v) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  counts[w] = lower((w + 1) * stride) - lower(w * stride);
}

// out[i, :] = table[keys[i] - key_sub, :]  (owner side of the lookup exchange)
template <int V>
__global__ void __launch_bounds__(kBlock)
gather_rows_kernel(const float* __restrict__ table, int64_t table_ld, const uint32_t* __restrict__ keys, int64_t n, int dim,
                   int G, int64_t key_sub, int64_t table_rows, float* __restrict__ out) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / G;
  const int c = (static_cast<int>(threadIdx.x) % G) * V;
  if (i >= n || c >= dim) return;
  const int64_t row = static_cast<int64_t>(keys[i]) - key_sub;
  Vec<V> e;
  if (row >= 0 && row < table_rows) e.load(table + row * table_ld + c); else e.zero();
  e.store(out + i * dim + c);
}

// ------------------------------------------------------------------------------------------------
// Owner side of the embedding-parallel exchange.  What a rank receives is `world` runs (one per requester), each
// ascending and free of duplicates (the requesters' er_emb_route wrote them), so the stable sort by key the
// reduction needs is a MERGE: an entry's sorted position is the number of received keys below its own in every
// run (a binary search per run, all L2-resident) plus the equal keys of earlier runs.  One launch instead of the
// device-wide radix sort's seven, and the run-head flags fall out of the same searches.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxRuns = 64;
struct MergeRuns {
  int n;
  int off[kMaxRuns + 1];
};

__global__ void __launch_bounds__(kBlock)
emb_owner_merge_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, MergeRuns r,
                       uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t* __restrict__ flags) {
  // (the run offsets are read with wave-uniform indices only: scalar loads from the kernarg segment - a per-lane
  // read of it goes to host memory over PCIe, ~1 us per workgroup and serialised: measured 140 us per launch)
  const int N = r.off[r.n];
  const int i = static_cast<int>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= N) return;
  const uint32_t k = keys_in[i];
  int own = 0;  // the run of entry i: the last one starting at or before i (empty runs share their offset)
  for (int q = 1; q < r.n; ++q) own += (r.off[q] <= i) ? 1 : 0;
  uint32_t pos = 0;
  bool first = true;
  for (int q = 0; q < r.n; ++q) {
    const int b = r.off[q], e = r.off[q + 1];
    int lo = b, hi = e;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys_in[mid] < k) lo = mid + 1; else hi = mid;
    }
    pos += static_cast<uint32_t>(lo - b);
    if (q < own && lo < e && keys_in[lo] == k) {
      ++pos;
      first = false;
    }
  }
  keys_out[pos] = k;
  vals_out[pos] = vals_in[i];
  flags[pos] = first ? 1u : 0u;
}

// Padded form (no host-visible counts: the exchange has fixed per-peer capacity `cap`, so the step needs no host
// synchronisation): run q occupies [q * cap, q * cap + counts[q]) of the received buffer, counts on the device.
// Entries past a run's count are padding: they get the invalid key and the positions after the N real ones.
__global__ void __launch_bounds__(kBlock)
emb_owner_ids_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ counts, int n_runs, int cap, int hdr,
                     int64_t key_sub, int64_t* __restrict__ ids, int32_t* __restrict__ counts_out) {
  const int i = static_cast<int>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n_runs * cap) return;
  const int q = i / cap, j = i - q * cap;
  // hdr: run q arrives as [count, keys ...] in cap + 1 slots; else the counts came in their own array
  const uint32_t* run = keys + static_cast<int64_t>(q) * (cap + hdr);
  const int cnt = hdr ? static_cast<int>(run[0]) : counts[q];
  ids[i] = j < cnt ? static_cast<int64_t>(run[hdr + j]) - key_sub : -1;
  if (j == 0 && counts_out) counts_out[q] = cnt;
}

__global__ void __launch_bounds__(kBlock)
emb_owner_merge_padded_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                              const int32_t* __restrict__ counts, int n_runs, int cap, uint32_t* __restrict__ keys_out,
                              uint32_t* __restrict__ vals_out, uint32_t* __restrict__ flags) {
  const int i = static_cast<int>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n_runs * cap) return;
  const int own = i / cap;
  const int j = i - own * cap;
  uint32_t pos = 0;
  bool first = true;
  const uint32_t k = keys_in[i];
  const bool valid = j < min(counts[own], cap);
  if (valid) {
    for (int q = 0; q < n_runs; ++q) {  // (q and counts[q] are wave-uniform: scalar loads)
      const int b = q * cap, e = b + min(counts[q], cap);
      int lo = b, hi = e;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys_in[mid] < k) lo = mid + 1; else hi = mid;
      }
      pos += static_cast<uint32_t>(lo - b);
      if (q < own && lo < e && keys_in[lo] == k) {
        ++pos;
        first = false;
      }
    }
  } else {
    int n_valid = 0, pad_before = 0;
    for (int q = 0; q < n_runs; ++q) {
      const int c = min(counts[q], cap);
      n_valid += c;
      if (q < own) pad_before += cap - c;
    }
    pos = static_cast<uint32_t>(n_valid + pad_before + (j - min(counts[own], cap)));
    first = false;
  }
  keys_out[pos] = valid ? k : kInvalidKey;
  vals_out[pos] = vals_in[i];
  flags[pos] = first ? 1u : 0u;
}

// The owner's side between the key exchange and the serve launch as ONE launch (er_emb_owner_ids_merge): emb_owner_ids_kernel,
// the entry build of the owner group and of the groups that share its sort (build_body) and emb_owner_merge_padded_kernel were
// three launches of n_runs * cap threads each, one chained to the next only through arrays every thread can derive for ANY
// position from the received keys themselves: the key of a received slot p is what the build writes for it - key_base + id,
// or the invalid key for an id outside the table - so the merge's binary searches read the RECEIVED runs and no thread waits
// for another workgroup's build.  Workgroups behind the n_main of the merge build the closed-form replay's lag-1 table for
// the serve launch that follows (decay_tables_kernel's body, one wavefront per k).  Same arithmetic per entry: every array
// this launch leaves is bit for bit what the three (four) launches leave.
__global__ void __launch_bounds__(kBlock)
emb_owner_ids_merge_kernel(const uint32_t* recv, const int32_t* counts, int n_runs, int cap, int hdr, int64_t key_sub,
                           int64_t* ids, int32_t* counts_out, BuildMulti bm, int n_main, uint32_t* __restrict__ keys_out,
                           uint32_t* __restrict__ vals_out, uint32_t* __restrict__ flags, DecayTabDev tabs,
                           const float* __restrict__ hist, const int64_t* __restrict__ counter) {
  if (static_cast<int>(blockIdx.x) >= n_main) {
    const int k = static_cast<int>(((blockIdx.x - n_main) * kBlock + threadIdx.x) >> 6) + 1;
    if (k > tabs.K) return;
    const int64_t s_end = *counter - 1;  // lag 1 (decay_tables_kernel)
    const float mine = decay_sum_wave(tabs, hist, s_end - 1 - k, k);
    const int lane = threadIdx.x & 63;
    float* A = tabs.A + static_cast<int64_t>(kDecayKMax) * kDecayLd;
    if (lane < kDecayLd) A[static_cast<int64_t>(k - 1) * kDecayLd + lane] = mine;
    return;
  }
  const int i = static_cast<int>(blockIdx.x) * kBlock + threadIdx.x;
  const bool live = i < n_runs * cap;
  const int own = live ? i / cap : 0;
  const int j = i - own * cap;
  const int stride = cap + hdr;
  // (q and the run counts are wave-uniform: scalar loads)
  auto run_count = [&](int q) { return hdr ? static_cast<int>(recv[static_cast<int64_t>(q) * stride]) : counts[q]; };
  // emb_owner_ids_kernel
  int64_t my_id = -1;
  if (live) {
    const uint32_t* run = recv + static_cast<int64_t>(own) * stride;
    const int cnt = run_count(own);
    my_id = j < cnt ? static_cast<int64_t>(run[hdr + j]) - key_sub : -1;
    ids[i] = my_id;
    if (j == 0 && counts_out) counts_out[own] = cnt;
  }
  // the entry build of the group and of its followers (their lookups read ids[] - this thread's own store)
  for (int g = 0; g < bm.n; ++g) {
    const BuildArgs& a = bm.a[g];
    build_body(blockIdx.x, a.descs, a.blk_start, a.ent_base, a.n_lookups, a.rt, a.n_active, a.keys, a.vals, a.ent_gptr,
               a.ent_scale);
  }
  if (!live) return;
  // emb_owner_merge_padded_kernel on keys derived from the received runs
  const er_lookup_desc d0 = bm.a[0].descs[0];
  auto key_of_id = [&](int64_t id) {
    return (id < 0 || id >= d0.rows) ? kInvalidKey : static_cast<uint32_t>(d0.key_base + id);
  };
  auto key_at = [&](int q, int jj) {  // the key the build leaves at slot q * cap + jj (jj < that run's count)
    return key_of_id(static_cast<int64_t>(recv[static_cast<int64_t>(q) * stride + hdr + jj]) - key_sub);
  };
  uint32_t pos = 0;
  bool first = true;
  const uint32_t k = key_of_id(my_id);
  const bool valid = j < min(run_count(own), cap);
  if (valid) {
    for (int q = 0; q < n_runs; ++q) {
      const int e = min(run_count(q), cap);
      int lo = 0, hi = e;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key_at(q, mid) < k) lo = mid + 1; else hi = mid;
      }
      pos += static_cast<uint32_t>(lo);
      if (q < own && lo < e && key_at(q, lo) == k) {
        ++pos;
        first = false;
      }
    }
  } else {
    int n_valid = 0, pad_before = 0;
    for (int q = 0; q < n_runs; ++q) {
      const int c = min(run_count(q), cap);
      n_valid += c;
      if (q < own) pad_before += cap - c;
    }
    pos = static_cast<uint32_t>(n_valid + pad_before + (j - min(run_count(own), cap)));
    first = false;
  }
  keys_out[pos] = valid ? k : kInvalidKey;
  vals_out[pos] = static_cast<uint32_t>(i);  // (= vals_in[i]: a dense-mode lookup's entry index)
  flags[pos] = first ? 1u : 0u;
}

// Serve the received keys: one lane group per sorted position that heads a run brings the row up to date (lazy
// dense decay of TF-exact Adam: the catch-up of catch_up_body) and writes it to the reply slot of EVERY entry of
// the run (at most one per requester), so each distinct row is read once.
struct ServeArgs {
  const uint32_t* skeys;
  const uint32_t* svals;
  const uint32_t* flags;
  int64_t n;
  RowUpdate tab;         // last_step == nullptr: no catch-up
  const float* lr_hist;
  DecayAux aux;
  float* out;            // [n, ld] reply rows in entry order (ld >= dim: several groups' rows side by side)
  int dim, G, V, ld;
};
struct ServeMulti {
  int n;
  int start[kMaxMulti + 1];
  const er_opt_hyper* hyper;
  ServeArgs a[kMaxMulti];
};

template <int V>
__device__ __forceinline__ void serve_body(int bid, const ServeArgs& a, const er_opt_hyper* __restrict__ hyper) {
  const int64_t p = (static_cast<int64_t>(bid) * kBlock + threadIdx.x) / a.G;
  const int c = (static_cast<int>(threadIdx.x) % a.G) * V;
  if (p >= a.n || c >= a.dim || !a.flags[p]) return;
  const uint32_t key = a.skeys[p];
  if (key == kInvalidKey) return;
  const int64_t off = a.tab.off(key, c);
  float var[V];
  ld_vec<V>(var, a.tab.var + off);
  if (a.tab.last_step) {
    const int32_t t = static_cast<int32_t>(*a.tab.step_counter - 1);
    const int32_t last = a.tab.ls(key);
    if (last + 1 < t) {
      float m[V], v[V];
      ld_vec<V>(m, a.tab.m + off);
      ld_vec<V>(v, a.tab.v + off);
      bool live = false;
#pragma unroll
      for (int j = 0; j < V; ++j) live = live || (m[j] != 0.f) || (v[j] != 0.f);
      if (live) {
        if (a.aux.A != nullptr) replay_closed<V>(var, m, v, a.aux, last + 1, t, pin_scalar(hyper->eps));
        else replay_decay<V>(var, m, v, LrHist{a.lr_hist, nullptr, 0}, last + 1, t, pin_decay_consts(*hyper), lr_cap_twice(a.aux.lr_max, t));
        st_vec<V>(a.tab.var + off, var);
        st_vec<V>(a.tab.m + off, m);
        st_vec<V>(a.tab.v + off, v);
      }
      // current up to t-1 from here on (see catch_up_body: the rolling flush of the step may run next to the step).
      // The G lanes of the row sit in one wavefront (G <= 64, a power of two): all have read last_step above.
      if (c == 0) a.tab.ls(key) = t - 1;
    }
  }
  for (int64_t q = p; q < a.n && a.skeys[q] == key; ++q)
    st_vec<V>(a.out + static_cast<int64_t>(a.svals[q]) * a.ld + c, var);
}

__global__ void __launch_bounds__(kBlock)
emb_owner_serve_kernel(ServeMulti ma) {
  int i = 0;
  while (i + 1 < ma.n && static_cast<int>(blockIdx.x) >= ma.start[i + 1]) ++i;
  const ServeArgs& a = ma.a[i];
  if (a.V == 4) serve_body<4>(blockIdx.x - ma.start[i], a, ma.hyper);
  else serve_body<1>(blockIdx.x - ma.start[i], a, ma.hyper);
}

// dense[key, 0:dim] = grads[i, :], dense[key, dim] = 1 for the n (device scalar) unique keys of a
// replicated table group: the rows are then all-reduced across ranks like dense parameters.
__global__ void __launch_bounds__(kBlock)
scatter_unique_kernel(const uint32_t* __restrict__ keys, const float* __restrict__ grads,
                      const int32_t* __restrict__ n_unique, int dim, float* __restrict__ dense, int dense_stride) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t i = t / (dim + 1);
  const int c = static_cast<int>(t % (dim + 1));
  if (i >= *n_unique) return;
  const int64_t row = keys[i];
  dense[row * dense_stride + c] = (c < dim) ? grads[i * dim + c] : 1.f;
}

// ------------------------------------------------------------------------------------------------
// TF-exact Adam: rows not touched this step still decay (m*=b1, v*=b2) and move.
// Pure streaming kernel: 3 arrays read + 3 written, float4, grid-stride, 4 units in flight per lane.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void decay_elem(float& var, float& m, float& v, const er_opt_hyper& h) {
  const float mt = m * h.beta1;
  const float vt = v * h.beta2;
  m = mt;
  v = vt;
  var = var - (h.lr_t * mt) / (sqrtf(vt) + h.eps);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void decay_vec(f32x4& var, f32x4& m, f32x4& v, const er_opt_hyper& h, uint32_t skip) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!((skip >> i) & 1u)) {
      float a = var[i], b = m[i], c = v[i];
      decay_elem(a, b, c, h);
      var[i] = a; m[i] = b; v[i] = c;
    }
  }
}

template <int UNROLL>
__global__ void __launch_bounds__(kBlock)
adam_decay_sweep_vec4_kernel(float* __restrict__ var, float* __restrict__ m, float* __restrict__ v,
                             const uint32_t* __restrict__ bitmap, int64_t n_units, int dim,
                             const er_opt_hyper* __restrict__ hyper) {
  const er_opt_hyper h = *hyper;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  f32x4* var4 = reinterpret_cast<f32x4*>(var);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  const int upr = dim >> 2;  // float4 units per row
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i0 < n_units; i0 += stride * UNROLL) {
    f32x4 a[UNROLL], b[UNROLL], c[UNROLL];
    bool live[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = i0 + u * stride;
      live[u] = false;
      if (i < n_units) {
        const int64_t row = i / upr;
        const uint32_t word = bitmap ? bitmap[row >> 5] : 0u;
        live[u] = ((word >> (row & 31)) & 1u) == 0u;
        if (live[u]) {
          a[u] = __builtin_nontemporal_load(&var4[i]);
          b[u] = __builtin_nontemporal_load(&m4[i]);
          c[u] = __builtin_nontemporal_load(&v4[i]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (live[u]) {
        const int64_t i = i0 + u * stride;
        decay_vec(a[u], b[u], c[u], h, 0u);
        __builtin_nontemporal_store(a[u], &var4[i]);
        __builtin_nontemporal_store(b[u], &m4[i]);
        __builtin_nontemporal_store(c[u], &v4[i]);
      }
    }
  }
}

// dim == 1 tables (wide columns): a float4 covers 4 consecutive rows.
template <int UNROLL>
__global__ void __launch_bounds__(kBlock)
adam_decay_sweep_dim1_kernel(float* __restrict__ var, float* __restrict__ m, float* __restrict__ v,
                             const uint32_t* __restrict__ bitmap, int64_t n_rows,
                             const er_opt_hyper* __restrict__ hyper) {
  const er_opt_hyper h = *hyper;
  const int64_t n_units = n_rows >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  f32x4* var4 = reinterpret_cast<f32x4*>(var);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i0 < n_units; i0 += stride * UNROLL) {
    f32x4 a[UNROLL], b[UNROLL], c[UNROLL];
    uint32_t bits[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = i0 + u * stride;
      bits[u] = 0xFu;
      if (i < n_units) {
        const int64_t row = i << 2;
        const uint32_t word = bitmap ? bitmap[row >> 5] : 0u;
        bits[u] = (word >> (row & 31)) & 0xFu;
        a[u] = __builtin_nontemporal_load(&var4[i]);
        b[u] = __builtin_nontemporal_load(&m4[i]);
        c[u] = __builtin_nontemporal_load(&v4[i]);
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n_units && bits[u] != 0xFu) {
        decay_vec(a[u], b[u], c[u], h, bits[u]);
        __builtin_nontemporal_store(a[u], &var4[i]);
        __builtin_nontemporal_store(b[u], &m4[i]);
        __builtin_nontemporal_store(c[u], &v4[i]);
      }
    }
  }
  // tail rows (n_rows % 4)
  const int64_t t = (n_units << 2) + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t < n_rows) {
    const uint32_t word = bitmap ? bitmap[t >> 5] : 0u;
    if (((word >> (t & 31)) & 1u) == 0u) {
      float a = var[t], b = m[t], c = v[t];
      decay_elem(a, b, c, h);
      var[t] = a; m[t] = b; v[t] = c;
    }
  }
}

// any other dim: scalar elements
__global__ void __launch_bounds__(kBlock)
adam_decay_sweep_scalar_kernel(float* __restrict__ var, float* __restrict__ m, float* __restrict__ v,
                               const uint32_t* __restrict__ bitmap, int64_t n_elems, int dim,
                               const er_opt_hyper* __restrict__ hyper) {
  const er_opt_hyper h = *hyper;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_elems; i += stride) {
    const int64_t row = i / dim;
    const uint32_t word = bitmap ? bitmap[row >> 5] : 0u;
    if (((word >> (row & 31)) & 1u) == 0u) {
      float a = var[i], b = m[i], c = v[i];
      decay_elem(a, b, c, h);
      var[i] = a; m[i] = b; v[i] = c;
    }
  }
}

// the same sweep over rows that lie at a pitch of ld floats (row records: er_emb_group_set_row_pitch); a lane owns V
// consecutive columns of a row (V = 4 when dim and ld are multiples of 4)
template <int V>
__global__ void __launch_bounds__(kBlock)
adam_decay_sweep_pitched_kernel(float* __restrict__ var, float* __restrict__ m, float* __restrict__ v,
                                const uint32_t* __restrict__ bitmap, int64_t n_rows, int dim, int64_t ld,
                                const er_opt_hyper* __restrict__ hyper) {
  const er_opt_hyper h = *hyper;
  const int upr = dim / V;  // lane units per row
  const int64_t n_units = n_rows * upr;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_units; i += stride) {
    const int64_t row = i / upr;
    const int c = static_cast<int>(i - row * upr) * V;
    const uint32_t word = bitmap ? bitmap[row >> 5] : 0u;
    if ((word >> (row & 31)) & 1u) continue;
    const int64_t off = row * ld + c;
    float a[V], b[V], cc[V];
    ld_vec<V>(a, var + off);
    ld_vec<V>(b, m + off);
    ld_vec<V>(cc, v + off);
#pragma unroll
    for (int j = 0; j < V; ++j) decay_elem(a[j], b[j], cc[j], h);
    st_vec<V>(var + off, a);
    st_vec<V>(m + off, b);
    st_vec<V>(v + off, cc);
  }
}

// Replicated (small, data-parallel) tables of the embedding-parallel path: after the all-reduce of the dense
// gradient buffer every rank applies the same update.  Row r of `dense` holds [grad (dim floats), count]; count > 0
// means some rank had a gradient for the row.  Touched rows take the optimizer step on grad * grad_scale; under
// TF-exact Adam the untouched rows take the decay-only step (what er_adam_decay_sweep does) - one pass over the
// table instead of build + sort + reduce + sweep + bitmap clear.  Up to kMaxMulti tables side by side.
struct DenseApplyArgs {
  float *var, *m, *v;
  const float* dense;
  int64_t tld;  // floats between consecutive rows of var / m / v (er_dense_apply_desc.table_ld, 0 = dim)
  int ld, dim, V, G;
  int64_t rows;
};
struct DenseApplyMulti {
  int n;
  int start[kMaxMulti + 1];
  int opt_kind;
  const er_opt_hyper* hyper;
  DenseApplyArgs a[kMaxMulti];
};

template <int V>
__device__ __forceinline__ void dense_apply_body(int bid, const DenseApplyArgs& a, int opt_kind,
                                                 const er_opt_hyper* __restrict__ hyper) {
  const int64_t t = static_cast<int64_t>(bid) * kBlock + threadIdx.x;
  const int64_t row = t / a.G;
  const int c = static_cast<int>(t % a.G) * V;
  if (row >= a.rows || c >= a.dim) return;
  const er_opt_hyper h = *hyper;
  const float* d = a.dense + row * a.ld;
  const int64_t off = row * a.tld + c;
  if (d[a.dim] > 0.f) {
    float g[V];
#pragma unroll
    for (int i = 0; i < V; ++i) g[i] = d[c + i] * h.grad_scale;
    if (h.clip_scale != 0.f) {
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] = g[i] * h.clip_scale;
    }
    update_row<V>(RowUpdate{a.var, a.m, a.v, nullptr, nullptr, nullptr, a.tld, 1}, opt_kind, h, off, g);
  } else if (opt_kind == ER_OPT_ADAM) {
    float var[V], m[V], v[V];
    ld_vec<V>(var, a.var + off);
    ld_vec<V>(m, a.m + off);
    ld_vec<V>(v, a.v + off);
#pragma unroll
    for (int i = 0; i < V; ++i) decay_elem(var[i], m[i], v[i], h);
    st_vec<V>(a.var + off, var);
    st_vec<V>(a.m + off, m);
    st_vec<V>(a.v + off, v);
  }
}

__device__ __forceinline__ void dense_apply_block(int b, const DenseApplyMulti& ma) {
  int i = 0;
  while (i + 1 < ma.n && b >= ma.start[i + 1]) ++i;
  const DenseApplyArgs& a = ma.a[i];
  if (a.V == 4) dense_apply_body<4>(b - ma.start[i], a, ma.opt_kind, ma.hyper);
  else dense_apply_body<1>(b - ma.start[i], a, ma.opt_kind, ma.hyper);
}

__global__ void __launch_bounds__(kBlock)
emb_dense_apply_kernel(DenseApplyMulti ma) {
  dense_apply_block(blockIdx.x, ma);
}

// The end of an embedding-parallel step as ONE launch (er_emb_owner_update_tail): the owner update's cross-tile fix
// (workgroups [0, n_fix)), the replicated tables' apply ([n_fix, n_fix + n_apply)) and the dense optimizer behind them -
// three launches that depend on nothing of each other (owned rows | replicated tables | dense variables), each at the
// 4-5 us floor of a launch.  Same bodies: bit-identical to the launches apart.
__global__ void __launch_bounds__(kBlock)
emb_bwd_fix_apply_opt_kernel(RunMulti fx, DenseApplyMulti am, DenseOptArgs da, int n_fix, int n_apply) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  if (b < n_fix) {
    fix_block(b, fx);
  } else if (b < n_fix + n_apply) {
    dense_apply_block(b - n_fix, am);
  } else {
    GroupedReduceArgs none;
    none.n = 0;  // (no k-split weight gradient is finished here: the requester's tail did that before the all-reduce)
    dense_opt_block(da, none, b - n_fix - n_apply, red);
  }
}

// Word fill used instead of hipMemsetAsync: inside a captured hipGraph a memset NODE was observed to lose its
// ordering against the neighbouring kernel nodes on replay (the touched-row bitmap stayed dirty into the next
// replay: eager and graph runs of TF-exact Adam diverged, tools/dbg_determinism.py); a kernel node keeps it.
__global__ void __launch_bounds__(kBlock)
fill_u32_kernel(uint32_t* __restrict__ p, uint32_t value, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) p[i] = value;
}

// Calibration / achievable-bandwidth probe: float4 copy with the sweep's access pattern (nontemporal,
// grid-stride, 4 units in flight).  Moves exactly 2*bytes; used to calibrate rocprofv3's FETCH_SIZE /
// WRITE_SIZE on a known byte count and to report the copy bandwidth next to the spec peak.
template <int UNROLL>
__global__ void __launch_bounds__(kBlock)
stream_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n_units) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i0 < n_units; i0 += stride * UNROLL) {
    f32x4 a[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n_units) a[u] = __builtin_nontemporal_load(&src[i]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n_units) __builtin_nontemporal_store(a[u], &dst[i]);
    }
  }
}

}  // namespace er

// ------------------------------------------------------------------------------------------------
// host side: handles
// ------------------------------------------------------------------------------------------------
struct er_emb_plan {
  int n = 0;
  int n_blocks = 0;
  er_lookup_desc* d_descs = nullptr;
  int32_t* d_blk_start = nullptr;
  std::vector<er_lookup_desc> h_descs;
  // er_emb_fwd_lazy: which table group (index into the call's group list) each lookup reads, -1: none of them
  int8_t* d_lookup_group = nullptr;
  std::vector<int8_t> h_lookup_group;
};

struct er_emb_group {
  int n = 0;
  int32_t dim = 0;
  int V = 4, G = 4;
  int64_t total_rows = 0;
  int64_t n_entries = 0;
  int key_bits = 32;
  bool has_ragged = false;
  // embedding-parallel routing (world == 1: plain single-GPU keys) and the active prefix of a
  // single dense-mode lookup (owner-side groups whose entry count changes per step)
  // lazy dense decay (er_emb_group_enable_lazy_decay)
  int32_t* last_step = nullptr;
  const float* lr_hist = nullptr;
  er::DecayAux aux;               // lr_max: prefix maximum of lr_hist (er_emb_group_set_lr_max); A / C: closed form
  er_decay_tables* tabs = nullptr;  // er_emb_group_set_decay_tables
  const int64_t* step_counter = nullptr;
  int32_t world = 1;
  int64_t shard_stride = 0;
  int64_t* d_local_base = nullptr;
  int64_t n_active = -1;  // -1: all entries
  int seg_sort_pow2 = 0;   // > 0: per-lookup LDS sort (emb_segment_sort_kernel) with this padded size
  uint32_t* seg_count = nullptr;  // [n] distinct valid keys per lookup (fused heads of the segmented path)
  bool seg_narrow = false;        // 32-bit sort composites: (rows + 2) * P <= 2^32 for every lookup
  int seg_caps_pow2 = 0;          // padded per-lookup size if every lookup has at most kSegSortMax entries
  bool seg_routed = false;        // routed keys (er_emb_group_set_routing) keep per-lookup disjoint, increasing ranges
  bool seg_routed_narrow = false;
  int peer_hdr = 0;               // 1: the keys of owner w are preceded by their count, in peer_cap + 1 slots
  int64_t peer_cap = 0;           // > 0: er_emb_route writes owner w's keys at [w * peer_cap, ...) (padded exchange)
  int32_t* d_overflow = nullptr;  // set to 1 by er_emb_route when an owner's keys exceed peer_cap
  uint64_t merged_epoch = ~0ull;  // sort_epoch of the last er_emb_owner_merge (head_flags hold its run heads)
  bool sorted_valid = false;
  // er_emb_group_share_sort: `src` owns the sorted keys / entry permutation / run heads this group reduces over
  // (itself, or the leader whose keys are identical); the epochs tell a fresh leader sort from a stale one
  er_emb_group* leader = nullptr;
  er_emb_group* src = nullptr;
  uint64_t sort_epoch = 0, adopted_epoch = 0, heads_epoch = 0;
  std::vector<er_emb_group*> followers;  // groups that share this group's sort: their entries are built with ours
  uint64_t built_epoch = ~0ull;          // follower: leader sort epoch its entry arrays were built for
  std::vector<er_lookup_desc> h_descs;
  std::vector<int64_t> h_local_base;
  float *var = nullptr, *m = nullptr, *v = nullptr;
  int64_t ld = 0;     // floats between consecutive rows of var / m / v (er_emb_group_set_row_pitch; dim by default)
  int64_t ls_ld = 1;  // int32 words between consecutive rows' last_step
  uint32_t* bitmap = nullptr;
  int n_build_blocks = 0;
  er_lookup_desc* d_descs = nullptr;
  int32_t* d_blk_start = nullptr;
  int64_t* d_ent_base = nullptr;
  // workspace (owned)
  uint32_t *keys_in = nullptr, *keys_out = nullptr, *vals_in = nullptr, *vals_out = nullptr;
  const float** ent_gptr = nullptr;
  float* ent_scale = nullptr;
  float *tile_first = nullptr, *tile_last = nullptr;
  int tile_entries = 0;
  uint32_t *head_flags = nullptr, *head_index = nullptr;
  // the hand-written device-wide sort (emb_chunk_sort_kernel + emb_merge_pass_kernel) and head scan
  int chunk_pow2 = 0;                 // entries per sorted chunk (a power of two <= kSegSortMax)
  int64_t* d_chunk_base = nullptr;    // [chunks + 1]
  uint32_t* d_tile_sum = nullptr;     // [ceil(N / kMergeTile)] run heads per tile
  // er_emb_front / er_emb_bwd_fused (the fused single-GPU step)
  uint64_t front_epoch = ~0ull;   // sort_epoch of the last sort made by er_emb_front
  bool front_skip = false;        // ... which kept the one-row tables' entries out of the sort
  bool front_deferred = false;    // ... and left the rows' catch-up to er_emb_fwd_lazy / er_emb_bwd_fused (in registers)
  uint32_t* d_extra_keys = nullptr;  // keys of the one-row tables (always brought current by the fused catch-up)
  int32_t* d_proj_lookup = nullptr;  // their lookup indices
  int n_proj = -1;                   // -1: not collected yet
  float* d_proj_partial = nullptr;
  uint32_t* d_proj_ticket = nullptr;
  void* d_own_lookups = nullptr;           // er::OwnLookup[n]: the fused backward's per-lookup gather records
  std::vector<unsigned char> h_own_lookups;  // host shadow (re-uploaded when the finish descriptors' geometry changes)
};

namespace {

er::RowUpdate tab_of(const er_emb_group* g) {
  return er::RowUpdate{g->var, g->m, g->v, g->bitmap, g->last_step, g->step_counter, g->ld, g->ls_ld};
}

int validate_descs(const er_lookup_desc* descs, int n, const char* who) {
  for (int i = 0; i < n; ++i) {
    const er_lookup_desc& d = descs[i];
    ER_REQUIRE(d.dim > 0 && d.dim <= 1024, "%s: lookup %d: dim %d out of range (1..1024)", who, i, d.dim);
    ER_REQUIRE(d.dim % 4 == 0 || d.dim <= 64, "%s: lookup %d: dim %d must be a multiple of 4 or <= 64", who, i, d.dim);
    ER_REQUIRE(d.n_rows >= 0 && d.rows > 0, "%s: lookup %d: bad n_rows/rows", who, i);
    ER_REQUIRE(d.table && d.ids && d.out, "%s: lookup %d: null table/ids/out pointer", who, i);
    ER_REQUIRE(d.combiner >= 0 && d.combiner <= 2, "%s: lookup %d: bad combiner %d", who, i, d.combiner);
    if (d.dim % 4 == 0) {
      ER_REQUIRE((reinterpret_cast<uintptr_t>(d.table) & 15) == 0, "%s: lookup %d: table must be 16-byte aligned", who, i);
    }
  }
  return 0;
}

}  // namespace

extern "C" {

int er_emb_plan_create(const er_lookup_desc* descs, int n, er_emb_plan** out) {
  ER_REQUIRE(descs && n > 0 && out, "er_emb_plan_create: bad arguments");
  if (int rc = validate_descs(descs, n, "er_emb_plan_create")) return rc;
  auto* p = new er_emb_plan();
  p->n = n;
  std::vector<int32_t> blk(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    int V, G;
    er::lane_geom(descs[i].dim, V, G);
    const int rpb = er::kBlock / G;
    blk[i + 1] = blk[i] + static_cast<int32_t>(er::ceil_div(descs[i].n_rows > 0 ? descs[i].n_rows : 1, rpb));
  }
  p->n_blocks = blk[n];
  ER_CHECK_HIP(hipMalloc(&p->d_descs, sizeof(er_lookup_desc) * n));
  ER_CHECK_HIP(hipMalloc(&p->d_blk_start, sizeof(int32_t) * (n + 1)));
  ER_CHECK_HIP(hipMemcpy(p->d_descs, descs, sizeof(er_lookup_desc) * n, hipMemcpyHostToDevice));
  ER_CHECK_HIP(hipMemcpy(p->d_blk_start, blk.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
  p->h_descs.assign(descs, descs + n);
  *out = p;
  return 0;
}

int er_emb_plan_update(er_emb_plan* p, const er_lookup_desc* descs, int n) {
  ER_REQUIRE(p && descs && n == p->n, "er_emb_plan_update: lookup count changed");
  if (int rc = validate_descs(descs, n, "er_emb_plan_update")) return rc;
  ER_CHECK_HIP(hipMemcpy(p->d_descs, descs, sizeof(er_lookup_desc) * n, hipMemcpyHostToDevice));
  p->h_descs.assign(descs, descs + n);
  return 0;
}

int er_emb_plan_destroy(er_emb_plan* p) {
  if (!p) return 0;
  (void)hipFree(p->d_descs);
  (void)hipFree(p->d_blk_start);
  (void)hipFree(p->d_lookup_group);
  delete p;
  return 0;
}

int er_emb_plan_num_blocks(const er_emb_plan* p) { return p ? p->n_blocks : -1; }

int er_emb_fwd(const er_emb_plan* p, float* sumsq_partials, er_stream_t stream) {
  ER_REQUIRE(p, "er_emb_fwd: null plan");
  hipLaunchKernelGGL(er::emb_fwd_kernel, dim3(p->n_blocks), dim3(er::kBlock), 0, er::as_stream(stream), p->d_descs,
                     p->d_blk_start, p->n, sumsq_partials);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_emb_group_create(const er_lookup_desc* descs, int n, int32_t dim, int64_t total_rows, float* var, float* m,
                        float* v, uint32_t* bitmap, er_emb_group** out) {
  ER_REQUIRE(descs && n > 0 && out && var, "er_emb_group_create: bad arguments");
  if (int rc = validate_descs(descs, n, "er_emb_group_create")) return rc;
  ER_REQUIRE(total_rows > 0 && total_rows < 0xFFFFFFFFLL, "er_emb_group_create: total_rows %lld does not fit 32-bit keys",
             (long long)total_rows);
  auto* g = new er_emb_group();
  g->n = n;
  g->dim = dim;
  er::lane_geom(dim, g->V, g->G);
  g->total_rows = total_rows;
  g->var = var; g->m = m; g->v = v; g->bitmap = bitmap;
  g->ld = dim;
  g->key_bits = 1;
  while ((1LL << g->key_bits) <= total_rows) ++g->key_bits;  // 2^bits > total_rows: invalid key sorts last
  std::vector<int32_t> blk(n + 1, 0);
  std::vector<int64_t> base(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    ER_REQUIRE(descs[i].dim == dim, "er_emb_group_create: lookup %d has dim %d, group dim %d", i, descs[i].dim, dim);
    ER_REQUIRE(descs[i].key_base >= 0 && descs[i].key_base + descs[i].rows <= total_rows,
               "er_emb_group_create: lookup %d table outside the group", i);
    blk[i + 1] = blk[i] + static_cast<int32_t>(er::ceil_div(descs[i].n_rows > 0 ? descs[i].n_rows : 1, er::kBlock));
    const int64_t cap = descs[i].offsets ? descs[i].max_nnz : descs[i].n_rows;
    ER_REQUIRE(cap >= 0, "er_emb_group_create: lookup %d: max_nnz not set", i);
    base[i + 1] = base[i] + cap;
    if (descs[i].offsets) g->has_ragged = true;
  }
  g->n_build_blocks = blk[n];
  g->n_entries = base[n];
  g->src = g;
  g->h_descs.assign(descs, descs + n);
  {
    int64_t mc = 0;
    for (int i = 0; i < n; ++i) mc = std::max<int64_t>(mc, base[i + 1] - base[i]);
    if (mc > 0 && mc <= er::kSegSortMax) {
      int p2 = er::kSegSortMin;
      while (p2 < mc) p2 <<= 1;
      g->seg_caps_pow2 = p2;
    }
  }
  {
    // segmented-sort fast path: dense or small lookups, each on its own table, key ranges increasing in lookup order
    bool ok = true;
    int64_t max_cap = 0;
    for (int i = 0; i < n && ok; ++i) {
      const int64_t cap = base[i + 1] - base[i];
      if (cap > max_cap) max_cap = cap;
      if (cap > er::kSegSortMax) ok = false;
      if (i + 1 < n && descs[i].key_base + descs[i].rows > descs[i + 1].key_base) ok = false;
    }
    if (ok && max_cap > 0) {
      int p2 = er::kSegSortMin;
      while (p2 < max_cap) p2 <<= 1;
      g->seg_sort_pow2 = p2;
      g->seg_narrow = true;
      for (int i = 0; i < n; ++i)
        if ((static_cast<uint64_t>(descs[i].rows) + 2) * static_cast<uint64_t>(p2) > (1ull << 32)) g->seg_narrow = false;
    }
  }
  ER_REQUIRE(g->n_entries > 0 && g->n_entries < 0x7FFFFFFFLL, "er_emb_group_create: entry count %lld out of range",
             (long long)g->n_entries);
  const int64_t N = g->n_entries;
  ER_CHECK_HIP(hipMalloc(&g->d_descs, sizeof(er_lookup_desc) * n));
  ER_CHECK_HIP(hipMalloc(&g->d_blk_start, sizeof(int32_t) * (n + 1)));
  ER_CHECK_HIP(hipMalloc(&g->d_ent_base, sizeof(int64_t) * (n + 1)));
  ER_CHECK_HIP(hipMemcpy(g->d_descs, descs, sizeof(er_lookup_desc) * n, hipMemcpyHostToDevice));
  ER_CHECK_HIP(hipMemcpy(g->d_blk_start, blk.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
  ER_CHECK_HIP(hipMemcpy(g->d_ent_base, base.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
  ER_CHECK_HIP(hipMalloc(&g->keys_in, sizeof(uint32_t) * N));
  ER_CHECK_HIP(hipMalloc(&g->keys_out, sizeof(uint32_t) * N));
  ER_CHECK_HIP(hipMalloc(&g->vals_in, sizeof(uint32_t) * N));
  ER_CHECK_HIP(hipMalloc(&g->vals_out, sizeof(uint32_t) * N));
  ER_CHECK_HIP(hipMalloc(&g->ent_gptr, sizeof(float*) * N));
  ER_CHECK_HIP(hipMalloc(&g->ent_scale, sizeof(float) * N));
  g->tile_entries = er::tile_passes(g->V) * (er::kBlock / g->G);
  {
    const size_t nt = static_cast<size_t>(er::ceil_div(N, g->tile_entries)) + 1;
    ER_CHECK_HIP(hipMalloc(&g->tile_first, sizeof(float) * nt * dim));
    ER_CHECK_HIP(hipMalloc(&g->tile_last, sizeof(float) * nt * dim));
  }
  ER_CHECK_HIP(hipMalloc(&g->head_flags, sizeof(uint32_t) * N));
  ER_CHECK_HIP(hipMalloc(&g->head_index, sizeof(uint32_t) * N));
  ER_CHECK_HIP(hipMalloc(&g->seg_count, sizeof(uint32_t) * (static_cast<size_t>(n) * 64 + 1)));  // [n][64 owners]
  ER_CHECK_HIP(hipMemset(g->keys_in, 0xFF, sizeof(uint32_t) * N));
  {  // the device-wide sort's chunks (all but the last hold chunk_pow2 entries) and the head scan's tile totals
    int p2 = er::kSegSortMin;
    while (p2 < N && p2 < er::kChunkSortMax) p2 <<= 1;
    g->chunk_pow2 = p2;
    const int64_t chunks = er::ceil_div(N, p2);
    std::vector<int64_t> cb(static_cast<size_t>(chunks) + 1);
    for (int64_t c = 0; c <= chunks; ++c) cb[static_cast<size_t>(c)] = std::min<int64_t>(c * p2, N);
    ER_CHECK_HIP(hipMalloc(&g->d_chunk_base, sizeof(int64_t) * cb.size()));
    ER_CHECK_HIP(hipMemcpy(g->d_chunk_base, cb.data(), sizeof(int64_t) * cb.size(), hipMemcpyHostToDevice));
    ER_CHECK_HIP(hipMalloc(&g->d_tile_sum, sizeof(uint32_t) * static_cast<size_t>(er::ceil_div(N, er::kMergeTile) + 1)));
  }
  *out = g;
  return 0;
}

int er_emb_group_update(er_emb_group* g, const er_lookup_desc* descs, int n) {
  ER_REQUIRE(g && descs && n == g->n, "er_emb_group_update: lookup count changed");
  if (int rc = validate_descs(descs, n, "er_emb_group_update")) return rc;
  for (int i = 0; i < n; ++i)
    ER_REQUIRE(descs[i].rows == g->h_descs[i].rows && descs[i].key_base == g->h_descs[i].key_base &&
                   descs[i].dim == g->h_descs[i].dim,
               "er_emb_group_update: lookup %d: the table geometry (rows, key_base, dim) cannot change", i);
  ER_CHECK_HIP(hipMemcpy(g->d_descs, descs, sizeof(er_lookup_desc) * n, hipMemcpyHostToDevice));
  g->h_descs.assign(descs, descs + n);
  return 0;
}

int er_emb_group_destroy(er_emb_group* g) {
  if (!g) return 0;
  if (g->leader) {
    auto& fl = g->leader->followers;
    fl.erase(std::remove(fl.begin(), fl.end(), g), fl.end());
  }
  for (er_emb_group* f : g->followers) {  // followers fall back to their own sort
    f->leader = nullptr;
    f->src = f;
  }
  void* ptrs[] = {g->d_descs, g->d_blk_start, g->d_ent_base, g->keys_in, g->keys_out, g->vals_in, g->vals_out,
                  g->ent_gptr, g->ent_scale, g->tile_first, g->tile_last, g->head_flags, g->head_index, g->d_chunk_base, g->d_tile_sum,
                  g->d_local_base, g->seg_count, g->d_overflow, g->d_extra_keys, g->d_proj_lookup, g->d_proj_partial,
                  g->d_proj_ticket, g->d_own_lookups};
  for (void* q : ptrs) (void)hipFree(q);
  delete g;
  return 0;
}

int64_t er_emb_group_num_entries(const er_emb_group* g) { return g ? g->n_entries : -1; }

static int fill_u32(uint32_t* p, uint32_t value, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  int64_t blocks = er::ceil_div(n, er::kBlock);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(er::fill_u32_kernel, dim3(static_cast<int>(blocks)), dim3(er::kBlock), 0, s, p, value, n);
  ER_LAUNCH_CHECK();
  return 0;
}

static int64_t group_entries(const er_emb_group* g) { return g->n_active >= 0 ? g->n_active : g->n_entries; }

static bool emb_group_same_keys(const er_emb_group* g, const er_emb_group* l);

// Entry arrays (keys, gradient pointers, scales) of g - and, in the same launch, of the groups that share its sort
// (they are about to adopt it: er_emb_group_share_sort).
static int emb_group_build_prepare(er_emb_group* g, hipStream_t s, er::BuildMulti* out, er_emb_group** gs);

static int emb_group_build(er_emb_group* g, hipStream_t s) {
  const int64_t N = group_entries(g);
  if (N == 0) return 0;
  er_emb_group* gs[er::kMaxMulti];
  er::BuildMulti ma;
  if (int rc = emb_group_build_prepare(g, s, &ma, gs)) return rc;
  const int n = ma.n;
  if (n == 1) {
    hipLaunchKernelGGL(er::emb_bwd_build_kernel, dim3(g->n_build_blocks), dim3(er::kBlock), 0, s, ma.a[0].descs,
                       ma.a[0].blk_start, ma.a[0].ent_base, ma.a[0].n_lookups, ma.a[0].rt, ma.a[0].n_active, ma.a[0].keys,
                       ma.a[0].vals, ma.a[0].ent_gptr, ma.a[0].ent_scale);
  } else {
    hipLaunchKernelGGL(er::emb_bwd_build_multi_kernel, dim3(ma.start[n]), dim3(er::kBlock), 0, s, ma);
  }
  ER_LAUNCH_CHECK();
  for (int i = 1; i < n; ++i) gs[i]->built_epoch = g->sort_epoch + 1;  // the sort that follows bumps the epoch
  return 0;
}

// the build launch's record for g and the groups that share its sort (gs: the groups, g first); ragged groups' key fills
// are launched here
static int emb_group_build_prepare(er_emb_group* g, hipStream_t s, er::BuildMulti* out, er_emb_group** gs) {
  gs[0] = g;
  int n = 1;
  for (er_emb_group* f : g->followers)
    if (n < er::kMaxMulti && group_entries(f) > 0 && emb_group_same_keys(f, g)) gs[n++] = f;
  er::BuildMulti& ma = *out;
  ma.n = n;
  ma.start[0] = 0;
  for (int i = 0; i < n; ++i) {
    er_emb_group* q = gs[i];
    if (q->has_ragged)
      if (int rc = fill_u32(q->keys_in, 0xFFFFFFFFu, group_entries(q), s)) return rc;
    ma.a[i] = er::BuildArgs{q->d_descs, q->d_blk_start, q->d_ent_base, q->n,
                            er::Route{q->world, q->shard_stride, q->d_local_base},
                            q->n_active >= 0 ? q->n_active : INT64_MAX, q->keys_in, q->vals_in, q->ent_gptr, q->ent_scale};
    ma.start[i + 1] = ma.start[i] + q->n_build_blocks;
  }
  return 0;
}

// true when the keys of g are, by construction, those of its leader: every lookup reads the same ids with the same
// table geometry and routing (re-checked on the host at every call: er_emb_group_update may have changed either)
static bool emb_group_same_keys(const er_emb_group* g, const er_emb_group* l) {
  if (!l || g->n != l->n || g->n_active != l->n_active) return false;  // (owner groups: the same rows received)
  if (g->world != l->world || g->shard_stride != l->shard_stride || g->h_local_base != l->h_local_base) return false;
  if ((g->d_local_base == nullptr) != (l->d_local_base == nullptr)) return false;
  for (int i = 0; i < g->n; ++i) {
    const er_lookup_desc &a = g->h_descs[i], &b = l->h_descs[i];
    if (a.ids != b.ids || a.offsets != b.offsets || a.rows != b.rows || a.key_base != b.key_base ||
        a.n_rows != b.n_rows || a.max_nnz != b.max_nnz)
      return false;
  }
  return true;
}

// Follower of a shared sort: only the per-entry gradient pointers / scales are built; the sorted keys, the entry
// permutation and (after the leader's er_emb_route) the run heads are the leader's.
static int emb_group_adopt(er_emb_group* g, hipStream_t s, bool* adopted) {
  *adopted = false;
  er_emb_group* l = g->leader;
  if (!emb_group_same_keys(g, l)) return 0;
  ER_REQUIRE(l->sort_epoch != g->adopted_epoch,
             "shared sort: the leader group has not been processed since this group last used its sort "
             "(call the leader first in every step)");
  if (g->built_epoch != l->sort_epoch)  // normally built together with the leader's entries (emb_group_build)
    if (int rc = emb_group_build(g, s)) return rc;
  g->adopted_epoch = l->sort_epoch;
  g->src = l;
  *adopted = true;
  return 0;
}

// keys_in / vals_in (the built entries; N of them active) -> keys_out / vals_out in the order of a stable sort by key:
// chunk sorts, then log2(chunks) merge passes that ping-pong between the two array pairs (the chunk sort writes the pair
// from which an even number of passes ends in keys_out / vals_out; it may sort in place: a workgroup holds its whole
// chunk in registers before it stores)
static int emb_group_wide_sort(er_emb_group* g, int64_t N, hipStream_t s) {
  const int P = g->chunk_pow2;
  const int64_t chunks = er::ceil_div(N, P);
  int passes = 0;
  for (int64_t r = P; r < N; r <<= 1) ++passes;
  uint32_t *ka = g->keys_out, *va = g->vals_out, *kb = g->keys_in, *vb = g->vals_in;
  if (passes & 1) { std::swap(ka, kb); std::swap(va, vb); }
  const size_t lds = sizeof(unsigned long long) * static_cast<size_t>(P);
  // (d_chunk_base was built for n_entries; an owner group may hold fewer this step: the kernel bounds its last chunk by N)
  if (P > 4096) {
    hipLaunchKernelGGL((er::emb_chunk_sort_kernel<8>), dim3(static_cast<unsigned>(chunks)), dim3(P / 8), lds, s, g->keys_in,
                       g->d_chunk_base, P, N, ka, va);
  } else {
    hipLaunchKernelGGL((er::emb_chunk_sort_kernel<4>), dim3(static_cast<unsigned>(chunks)), dim3(P / 4), lds, s, g->keys_in,
                       g->d_chunk_base, P, N, ka, va);
  }
  ER_LAUNCH_CHECK();
  const unsigned tiles = static_cast<unsigned>(er::ceil_div(N, er::kMergeTile));
  for (int64_t run = P; run < N; run <<= 1) {
    hipLaunchKernelGGL(er::emb_merge_pass_kernel, dim3(tiles), dim3(er::kBlock), 0, s, ka, va, N, run, kb, vb);
    ER_LAUNCH_CHECK();
    std::swap(ka, kb);
    std::swap(va, vb);
  }
  ER_REQUIRE(ka == g->keys_out, "wide sort: pass parity");
  return 0;
}

// build keys (routed) + stable sort.  Leaves keys_out/vals_out valid for this step.
static bool emb_group_segmented(const er_emb_group* g) {
  if (g->n_active >= 0) return false;
  return g->d_local_base ? g->seg_routed : g->seg_sort_pow2 > 0;
}

// with_heads (segmented path only): the sort kernel also leaves head flags, per-lookup head indices and counts
static int emb_group_build_sort(er_emb_group* g, hipStream_t s, bool with_heads = false) {
  const int64_t N = group_entries(g);
  if (N == 0) return 0;
  if (int rc = emb_group_build(g, s)) return rc;
  g->src = g;
  ++g->sort_epoch;
  if (emb_group_segmented(g)) {
    const bool routed = g->d_local_base != nullptr;
    const int P = routed ? g->seg_caps_pow2 : g->seg_sort_pow2;
    const bool narrow = routed ? g->seg_routed_narrow : g->seg_narrow;
    const er::Route srt{g->world, g->shard_stride, g->d_local_base};
    const size_t lds = sizeof(unsigned long long) * static_cast<size_t>(P);
#define ER_SEG_SORT(E, H, NRW)                                                                                     \
  hipLaunchKernelGGL((er::emb_segment_sort_kernel<E, H, NRW>), dim3(g->n), dim3(P / E), lds, s, g->keys_in,        \
                     g->d_ent_base, g->d_descs, P, srt, g->keys_out, g->vals_out, g->head_flags, g->head_index,     \
                     g->seg_count)
#define ER_SEG_SORT_E(E)                                                       \
  if (narrow) {                                                                \
    if (with_heads) ER_SEG_SORT(E, true, true); else ER_SEG_SORT(E, false, true);   \
  } else {                                                                     \
    if (with_heads) ER_SEG_SORT(E, true, false); else ER_SEG_SORT(E, false, false); \
  }
    if (P > 4096) { ER_SEG_SORT_E(8) } else { ER_SEG_SORT_E(4) }
#undef ER_SEG_SORT_E
#undef ER_SEG_SORT
    ER_LAUNCH_CHECK();
    return 0;
  }
  return emb_group_wide_sort(g, N, s);
}

static int emb_group_sort(er_emb_group* g, hipStream_t s) { return emb_group_build_sort(g, s); }

// head flags + exclusive scan + unique count over the sorted keys
static int emb_group_heads(er_emb_group* g, int32_t* n_unique, hipStream_t s) {
  const int64_t N = group_entries(g);
  const int tiles = static_cast<int>(er::ceil_div(N, er::kMergeTile));
  hipLaunchKernelGGL(er::emb_head_scan_kernel, dim3(tiles), dim3(er::kBlock), 0, s, g->src->keys_out, N, g->head_flags,
                     g->head_index, g->d_tile_sum);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::emb_head_offsets_kernel, dim3(tiles), dim3(er::kBlock), 0, s, g->d_tile_sum, tiles, N, g->head_index,
                     n_unique);
  ER_LAUNCH_CHECK();
  return 0;
}

// a sort made by er_emb_front with skip_one_row leaves the one-row tables' entries out: only er_emb_bwd_fused may reduce it
static int front_sort_guard(const er_emb_group* g, const char* who) {
  const er_emb_group* src = g->src ? g->src : g;
  ER_REQUIRE(!(src->front_skip && src->front_epoch == src->sort_epoch),
             "%s: this step's sort was made by er_emb_front(skip_one_row = 1): reduce it with er_emb_bwd_fused", who);
  return 0;
}

static int emb_group_run(er_emb_group* g, int opt_kind, const er_opt_hyper* hyper, int mode, uint32_t* out_keys,
                         float* out_grads, hipStream_t s, int out_ld = 0) {
  const int64_t N = group_entries(g);
  if (N == 0) return 0;
  if (int rc = front_sort_guard(g, "er_emb_bwd_*")) return rc;
  const int T = g->tile_entries;
  const int n_tiles = static_cast<int>(er::ceil_div(N, T));
  const er::RowUpdate tab = tab_of(g);
  const er_emb_group* src = g->src;  // whose sort this group reduces over (itself unless er_emb_group_share_sort)
  er::ReduceOut ro{mode, src->head_flags, src->head_index, out_keys, out_grads, out_ld};
  const size_t lds = sizeof(float) * static_cast<size_t>(T) * g->dim + sizeof(uint32_t) * (T + 2);
  const int fix_blocks = static_cast<int>(er::ceil_div(static_cast<int64_t>(n_tiles) * g->G, er::kBlock));
  if (g->V == 4) {
    hipLaunchKernelGGL(er::emb_bwd_tile_kernel<4>, dim3(n_tiles), dim3(er::kBlock), lds, s, src->keys_out, src->vals_out,
                       g->ent_gptr, g->ent_scale, N, g->dim, g->G, tab, opt_kind, hyper, ro, g->tile_first, g->tile_last);
    ER_LAUNCH_CHECK();
    if (n_tiles > 1)
      hipLaunchKernelGGL(er::emb_bwd_fix_kernel<4>, dim3(fix_blocks), dim3(er::kBlock), 0, s, src->keys_out, N, g->dim, g->G,
                         T, n_tiles, tab, opt_kind, hyper, ro, g->tile_first, g->tile_last);
  } else {
    hipLaunchKernelGGL(er::emb_bwd_tile_kernel<1>, dim3(n_tiles), dim3(er::kBlock), lds, s, src->keys_out, src->vals_out,
                       g->ent_gptr, g->ent_scale, N, g->dim, g->G, tab, opt_kind, hyper, ro, g->tile_first, g->tile_last);
    ER_LAUNCH_CHECK();
    if (n_tiles > 1)
      hipLaunchKernelGGL(er::emb_bwd_fix_kernel<1>, dim3(fix_blocks), dim3(er::kBlock), 0, s, src->keys_out, N, g->dim, g->G,
                         T, n_tiles, tab, opt_kind, hyper, ro, g->tile_first, g->tile_last);
  }
  ER_LAUNCH_CHECK();
  return 0;
}

// Workgroups per CU of the dense-decay sweep.  8 (the default) fills every wave slot and is right
// when the sweep runs alone; when it is overlapped with the latency-bound forward/backward kernels on
// another stream it must leave wave slots free or those kernels queue behind the sweep's
// long-running grid-stride blocks (measured).
static int g_sweep_blocks_per_cu = 8;

int er_config_set(const char* key, int64_t value) {
  ER_REQUIRE(key, "er_config_set: null key");
  if (strcmp(key, "sweep_blocks_per_cu") == 0) {
    ER_REQUIRE(value >= 1 && value <= 8, "er_config_set: sweep_blocks_per_cu must be in 1..8");
    g_sweep_blocks_per_cu = static_cast<int>(value);
    return 0;
  }
  ER_REQUIRE(false, "er_config_set: unknown key %s", key);
  return 2;
}

int er_adam_decay_sweep(float* var, float* m, float* v, uint32_t* bitmap, int64_t total_rows, int32_t dim,
                        const er_opt_hyper* hyper, er_stream_t stream) {
  return er_adam_decay_sweep_ld(var, m, v, bitmap, total_rows, dim, dim, hyper, stream);
}

int er_adam_decay_sweep_ld(float* var, float* m, float* v, uint32_t* bitmap, int64_t total_rows, int32_t dim, int64_t ld,
                           const er_opt_hyper* hyper, er_stream_t stream) {
  ER_REQUIRE(var && m && v && hyper && total_rows > 0 && dim > 0 && ld >= dim, "er_adam_decay_sweep: bad arguments");
  hipStream_t s = er::as_stream(stream);
  constexpr int U = 4;
  const int max_blocks = 256 * g_sweep_blocks_per_cu;
  if (ld != dim) {
    const bool vec = dim % 4 == 0 && ld % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(var) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const int64_t units = total_rows * (vec ? dim / 4 : dim);
    int64_t blocks = er::ceil_div(units, er::kBlock);
    if (blocks > max_blocks) blocks = max_blocks;
    if (vec)
      hipLaunchKernelGGL(er::adam_decay_sweep_pitched_kernel<4>, dim3(static_cast<int>(blocks)), dim3(er::kBlock), 0, s, var, m,
                         v, bitmap, total_rows, dim, ld, hyper);
    else
      hipLaunchKernelGGL(er::adam_decay_sweep_pitched_kernel<1>, dim3(static_cast<int>(blocks)), dim3(er::kBlock), 0, s, var, m,
                         v, bitmap, total_rows, dim, ld, hyper);
  } else if (dim % 4 == 0) {
    const int64_t units = total_rows * (dim / 4);
    int64_t blocks = er::ceil_div(units, static_cast<int64_t>(er::kBlock) * U);
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(er::adam_decay_sweep_vec4_kernel<U>, dim3(static_cast<int>(blocks)), dim3(er::kBlock), 0, s, var,
                       m, v, bitmap, units, dim, hyper);
  } else if (dim == 1) {
    const int64_t units = total_rows / 4;
    int64_t blocks = er::ceil_div(units > 0 ? units : 1, static_cast<int64_t>(er::kBlock) * U);
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(er::adam_decay_sweep_dim1_kernel<U>, dim3(static_cast<int>(blocks)), dim3(er::kBlock), 0, s, var,
                       m, v, bitmap, total_rows, hyper);
  } else {
    const int64_t elems = total_rows * dim;
    int64_t blocks = er::ceil_div(elems, er::kBlock);
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(er::adam_decay_sweep_scalar_kernel, dim3(static_cast<int>(blocks)), dim3(er::kBlock), 0, s, var,
                       m, v, bitmap, elems, dim, hyper);
  }
  ER_LAUNCH_CHECK();
  return 0;
}

int er_emb_bwd_update(er_emb_group* g, int opt_kind, const er_opt_hyper* hyper, er_stream_t stream) {
  ER_REQUIRE(g && hyper, "er_emb_bwd_update: null argument");
  ER_REQUIRE(opt_kind >= ER_OPT_SGD && opt_kind <= ER_OPT_ADAGRAD, "er_emb_bwd_update: unknown optimizer %d", opt_kind);
  if (opt_kind == ER_OPT_ADAM || opt_kind == ER_OPT_LAZY_ADAM)
    ER_REQUIRE(g->m && g->v, "er_emb_bwd_update: Adam needs m and v");
  if (opt_kind == ER_OPT_ADAGRAD) ER_REQUIRE(g->v, "er_emb_bwd_update: Adagrad needs the accumulator in v");
  const bool lazy_decay = (opt_kind == ER_OPT_ADAM) && g->last_step;
  if (opt_kind == ER_OPT_ADAM && !lazy_decay)
    ER_REQUIRE(g->bitmap, "er_emb_bwd_update: ER_OPT_ADAM needs touched_bitmap (or er_emb_group_enable_lazy_decay)");
  hipStream_t s = er::as_stream(stream);
  // a sort left by this step's er_emb_route (lazy dense decay, embedding-parallel requester) is reused
  if (!g->sorted_valid) {
    bool adopted = false;
    if (g->leader)
      if (int rc = emb_group_adopt(g, s, &adopted)) return rc;
    if (!adopted)
      if (int rc = emb_group_sort(g, s)) return rc;
  }
  g->sorted_valid = false;
  if (int rc = emb_group_run(g, opt_kind, hyper, 0, nullptr, nullptr, s)) return rc;
  if (opt_kind == ER_OPT_ADAM && !lazy_decay) {
    if (int rc = er_adam_decay_sweep_ld(g->var, g->m, g->v, g->bitmap, g->total_rows, g->dim, g->ld, hyper, stream)) return rc;
    if (int rc = fill_u32(g->bitmap, 0u, er::ceil_div(g->total_rows, 32), s)) return rc;
  }
  return 0;
}

// The reduce kernels of up to kMaxMulti groups side by side in one launch.  dense == nullptr: apply the optimizer
// (er_emb_bwd_update_multi); else: the row sums go to dense[i][key * ld[i] ...] (er_emb_bwd_reduce_dense).
static int emb_groups_run_multi(er_emb_group* const* groups, int n, int opt_kind, const er_opt_hyper* hyper,
                                float* const* dense, const int32_t* ld, er_stream_t stream,
                                const er::DenseApplyMulti* rider_apply = nullptr, const er::DenseOptArgs* rider_opt = nullptr) {
  hipStream_t s = er::as_stream(stream);
  er::RunMulti ma;
  ma.n = 0;
  ma.start[0] = 0;
  ma.opt_kind = opt_kind;
  ma.hyper = hyper;
  int fix_start[er::kMaxMulti + 1] = {0};
  size_t lds = 0;
  bool any_fix = false;
  for (int i = 0; i < n; ++i) {
    er_emb_group* g = groups[i];
    ER_REQUIRE(g, "er_emb_bwd_update_multi: null group %d", i);
    if (!dense) {
      if (opt_kind == ER_OPT_ADAM || opt_kind == ER_OPT_LAZY_ADAM)
        ER_REQUIRE(g->m && g->v, "er_emb_bwd_update_multi: Adam needs m and v (group %d)", i);
      if (opt_kind == ER_OPT_ADAGRAD) ER_REQUIRE(g->v, "er_emb_bwd_update_multi: Adagrad needs the accumulator in v");
      const bool lazy_decay = (opt_kind == ER_OPT_ADAM) && g->last_step;
      if (opt_kind == ER_OPT_ADAM && !lazy_decay)
        ER_REQUIRE(g->bitmap, "er_emb_bwd_update_multi: ER_OPT_ADAM needs touched_bitmap or lazy decay (group %d)", i);
    } else {
      ER_REQUIRE(dense[i] && ld[i] > g->dim, "er_emb_bwd_reduce_dense: group %d: null buffer or ld <= dim", i);
    }
    if (!g->sorted_valid) {  // as er_emb_bwd_update: reuse this step's er_emb_route, adopt the leader's sort, or sort
      bool adopted = false;
      if (g->leader)
        if (int rc = emb_group_adopt(g, s, &adopted)) return rc;
      if (!adopted)
        if (int rc = emb_group_sort(g, s)) return rc;
    }
    g->sorted_valid = false;
    const int64_t N = group_entries(g);
    if (N == 0) continue;
    if (int rc = front_sort_guard(g, "er_emb_bwd_update_multi")) return rc;
    const er_emb_group* src = g->src;
    er::RunArgs& a = ma.a[ma.n];
    a.skeys = src->keys_out; a.svals = src->vals_out; a.ent_gptr = g->ent_gptr; a.ent_scale = g->ent_scale;
    a.n = N; a.dim = g->dim; a.G = g->G; a.V = g->V; a.T = g->tile_entries;
    a.n_tiles = static_cast<int>(er::ceil_div(N, a.T));
    a.tab = tab_of(g);
    a.aux = er::DecayAux{};
    a.ro = dense ? er::ReduceOut{2, nullptr, nullptr, nullptr, dense[i], ld[i]}
                 : er::ReduceOut{0, src->head_flags, src->head_index, nullptr, nullptr, 0};
    a.tile_first = g->tile_first; a.tile_last = g->tile_last;
    ma.start[ma.n + 1] = ma.start[ma.n] + a.n_tiles;
    const int fb = a.n_tiles > 1 ? static_cast<int>(er::ceil_div(static_cast<int64_t>(a.n_tiles) * g->G, er::kBlock)) : 0;
    fix_start[ma.n + 1] = fix_start[ma.n] + fb;
    any_fix = any_fix || fb > 0;
    const size_t need = sizeof(float) * static_cast<size_t>(a.T) * g->dim + sizeof(uint32_t) * (a.T + 2);
    if (need > lds) lds = need;
    ++ma.n;
  }
  if (ma.n > 0) {
    hipLaunchKernelGGL(er::emb_bwd_tile_multi_kernel, dim3(ma.start[ma.n]), dim3(er::kBlock), lds, s, ma);
    ER_LAUNCH_CHECK();
  }
  if (rider_opt) {  // (er_emb_owner_update_tail) the fix launch also carries the replicated tables' apply and the dense optimizer
    for (int i = 0; i <= ma.n; ++i) ma.start[i] = fix_start[i];
    const int n_fix = (ma.n > 0 && any_fix) ? ma.start[ma.n] : 0;
    const int n_apply = (rider_apply && rider_apply->n > 0) ? rider_apply->start[rider_apply->n] : 0;
    const int n_opt = static_cast<int>(er::ceil_div(rider_opt->n, er::kBlock));
    er::DenseApplyMulti am;
    if (n_apply > 0) am = *rider_apply;
    else am.n = 0;
    hipLaunchKernelGGL(er::emb_bwd_fix_apply_opt_kernel, dim3(n_fix + n_apply + n_opt), dim3(er::kBlock), 0, s, ma, am,
                       *rider_opt, n_fix, n_apply);
    ER_LAUNCH_CHECK();
  } else if (ma.n > 0 && any_fix) {
    for (int i = 0; i <= ma.n; ++i) ma.start[i] = fix_start[i];
    hipLaunchKernelGGL(er::emb_bwd_fix_multi_kernel, dim3(ma.start[ma.n]), dim3(er::kBlock), 0, s, ma);
    ER_LAUNCH_CHECK();
  }
  for (int i = 0; i < n && !dense; ++i) {  // TF-exact Adam with the streaming sweep: per group, as er_emb_bwd_update
    er_emb_group* g = groups[i];
    if (opt_kind == ER_OPT_ADAM && !g->last_step) {
      if (int rc = er_adam_decay_sweep_ld(g->var, g->m, g->v, g->bitmap, g->total_rows, g->dim, g->ld, hyper, stream)) return rc;
      if (int rc = fill_u32(g->bitmap, 0u, er::ceil_div(g->total_rows, 32), s)) return rc;
    }
  }
  return 0;
}

int er_emb_bwd_update_multi(er_emb_group* const* groups, int n, int opt_kind, const er_opt_hyper* hyper,
                            er_stream_t stream) {
  ER_REQUIRE(groups && hyper && n >= 1 && n <= er::kMaxMulti, "er_emb_bwd_update_multi: bad arguments (1 <= n <= %d)",
             er::kMaxMulti);
  if (n == 1) return er_emb_bwd_update(groups[0], opt_kind, hyper, stream);
  ER_REQUIRE(opt_kind >= ER_OPT_SGD && opt_kind <= ER_OPT_ADAGRAD, "er_emb_bwd_update_multi: unknown optimizer %d", opt_kind);
  return emb_groups_run_multi(groups, n, opt_kind, hyper, nullptr, nullptr, stream);
}

int er_emb_bwd_reduce_dense(er_emb_group* const* groups, float* const* dense, const int32_t* ld, int n,
                            er_stream_t stream) {
  ER_REQUIRE(groups && dense && ld && n >= 1 && n <= er::kMaxMulti, "er_emb_bwd_reduce_dense: bad arguments (1 <= n <= %d)",
             er::kMaxMulti);
  return emb_groups_run_multi(groups, n, ER_OPT_SGD, nullptr, dense, ld, stream);
}

int er_emb_group_enable_lazy_decay(er_emb_group* g, int32_t* last_step, const float* lr_t_history,
                                   const int64_t* step_counter) {
  ER_REQUIRE(g && last_step && lr_t_history && step_counter, "er_emb_group_enable_lazy_decay: null argument");
  ER_REQUIRE(g->m && g->v, "er_emb_group_enable_lazy_decay: the group has no Adam slots");
  g->last_step = last_step;
  g->lr_hist = lr_t_history;
  g->step_counter = step_counter;
  return 0;
}

int er_emb_group_set_row_pitch(er_emb_group* g, int64_t ld, int64_t last_step_ld) {
  ER_REQUIRE(g && ld >= g->dim && last_step_ld >= 1, "er_emb_group_set_row_pitch: bad arguments (ld >= dim, last_step_ld >= 1)");
  if (g->V == 4) {
    ER_REQUIRE(ld % 4 == 0, "er_emb_group_set_row_pitch: ld %lld must be a multiple of 4 floats for 16-byte lanes", (long long)ld);
    ER_REQUIRE(((reinterpret_cast<uintptr_t>(g->var) | reinterpret_cast<uintptr_t>(g->m) | reinterpret_cast<uintptr_t>(g->v)) & 15) == 0,
               "er_emb_group_set_row_pitch: var / m / v must be 16-byte aligned");
  }
  g->ld = ld;
  g->ls_ld = last_step_ld;
  return 0;
}

// The closed form's per-launch table A depends on the step the launch brings its rows to (*counter - lag).  Consumers with
// DIFFERENT lags may run on different streams (er_emb_flush_decay: lag 0; catch-up, owner serve: lag 1; the rolling window:
// either), so each lag owns a table; concurrent consumers of ONE lag rebuild identical contents (same counter, same
// history), which is benign.  lag is 0 or 1 everywhere (checked by the callers).
static er::DecayAux decay_aux_for(const er_emb_group* g, int lag) {
  er::DecayAux a = g->aux;
  if (a.A != nullptr) a.A += static_cast<int64_t>(lag) * er::kDecayKMax * er::kDecayLd;
  return a;
}

// A[k] for the consumer launch that follows on `s` (rows brought to step *counter - lag); one launch per distinct table set
static int launch_decay_tables(er_emb_group* const* groups, int n, int lag, hipStream_t s) {
  const er_decay_tables* done[er::kMaxMulti];
  int n_done = 0;
  for (int i = 0; i < n; ++i) {
    const er_decay_tables* t = groups[i] ? groups[i]->tabs : nullptr;
    bool seen = t == nullptr;
    for (int j = 0; j < n_done; ++j) seen = seen || done[j] == t;
    if (seen) continue;
    if (n_done < er::kMaxMulti) done[n_done++] = t;
    if (lag == 1 && t->lag1_built) {  // (a rider of an earlier launch of this step built it: er_emb_owner_ids_merge)
      const_cast<er_decay_tables*>(t)->lag1_built = false;
      continue;
    }
    const int blocks = static_cast<int>(er::ceil_div(static_cast<int64_t>(t->dev.K) * er::kWave, er::kBlock));
    hipLaunchKernelGGL(er::decay_tables_kernel, dim3(blocks), dim3(er::kBlock), 0, s, t->dev, t->hist, t->counter, lag);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// The fused single-GPU step (round 4): er_emb_front = build + sort + run heads (+ the closed-form replay's table) in ONE
// launch per sort leader, then the catch-up straight from the heads; er_emb_bwd_fused = finish + reduce + row update in
// ONE launch.  Returns 3 ("not eligible", er_last_error says why) when a group needs what only the general path offers.
// ------------------------------------------------------------------------------------------------
#define ER_ELIGIBLE(cond, ...)       \
  do {                               \
    if (!(cond)) {                   \
      er::set_error(__VA_ARGS__);    \
      return 3;                      \
    }                                \
  } while (0)

// >= 0 when every lookup of the group holds the same power-of-two number of entries
static int group_cap_shift(const er_emb_group* g) {
  const int64_t cap = g->h_descs[0].offsets ? g->h_descs[0].max_nnz : g->h_descs[0].n_rows;
  if (cap <= 0 || (cap & (cap - 1)) != 0) return -1;
  for (int i = 1; i < g->n; ++i)
    if ((g->h_descs[i].offsets ? g->h_descs[i].max_nnz : g->h_descs[i].n_rows) != cap) return -1;
  int sh = 0;
  while ((1LL << sh) < cap) ++sh;
  return sh;
}

static int front_collect_one_row(er_emb_group* g) {
  if (g->n_proj >= 0) return 0;
  std::vector<uint32_t> keys;
  std::vector<int32_t> idx;
  for (int i = 0; i < g->n; ++i)
    if (g->h_descs[i].rows == 1) {
      keys.push_back(static_cast<uint32_t>(g->h_descs[i].key_base));
      idx.push_back(i);
    }
  ER_REQUIRE(keys.size() <= 64, "er_emb_front: more than 64 one-row tables in a table group");
  g->n_proj = static_cast<int>(keys.size());
  if (g->n_proj > 0) {
    ER_CHECK_HIP(hipMalloc(&g->d_extra_keys, sizeof(uint32_t) * keys.size()));
    ER_CHECK_HIP(hipMalloc(&g->d_proj_lookup, sizeof(int32_t) * idx.size()));
    ER_CHECK_HIP(hipMalloc(&g->d_proj_partial, sizeof(float) * keys.size() * er::kProjParts * (g->dim + 1)));
    ER_CHECK_HIP(hipMalloc(&g->d_proj_ticket, sizeof(uint32_t) * keys.size()));
    ER_CHECK_HIP(hipMemcpy(g->d_extra_keys, keys.data(), sizeof(uint32_t) * keys.size(), hipMemcpyHostToDevice));
    ER_CHECK_HIP(hipMemcpy(g->d_proj_lookup, idx.data(), sizeof(int32_t) * idx.size(), hipMemcpyHostToDevice));
    ER_CHECK_HIP(hipMemset(g->d_proj_ticket, 0, sizeof(uint32_t) * keys.size()));
  }
  return 0;
}

// what er_emb_front_fwd hands to the front: the step's lookup, to ride on the last sort launch
struct FrontFwd {
  er_emb_plan* plan;
  float* sumsq;
  er::FwdLazy lz;
  bool launched;
};

static int fwd_lazy_prepare(er_emb_plan* p, er_emb_group* const* groups, int n, const er_opt_hyper* hyper, hipStream_t s,
                            er::FwdLazy* lz);

static int emb_front_impl(er_emb_group* const* groups, int n, int flags, const er_opt_hyper* hyper, hipStream_t s,
                          FrontFwd* ff) {
  ER_REQUIRE(groups && n >= 1 && n <= er::kMaxMulti, "er_emb_front: bad arguments (1 <= n <= %d)", er::kMaxMulti);
  const int skip_one_row = flags & ER_FRONT_SKIP_ONE_ROW;
  const bool defer = (flags & ER_FRONT_DEFER_CATCH_UP) != 0;
  // eligibility first: nothing is launched for a set of groups the fused path does not cover
  for (int i = 0; i < n; ++i) {
    er_emb_group* g = groups[i];
    ER_REQUIRE(g, "er_emb_front: null group %d", i);
    ER_ELIGIBLE(g->d_local_base == nullptr && g->world == 1 && g->n_active < 0, "er_emb_front: group %d is routed / partly active", i);
    ER_ELIGIBLE(!g->has_ragged && g->seg_sort_pow2 > 0, "er_emb_front: group %d has ragged lookups or no per-lookup sort", i);
    ER_ELIGIBLE(!g->last_step || (g->tabs && g->G <= er::kWave), "er_emb_front: group %d replays its decay step by step", i);
    ER_ELIGIBLE(g->n <= er::kOwnMaxLookups, "er_emb_front: group %d has more than %d lookups", i, er::kOwnMaxLookups);
    if (g->last_step) ER_REQUIRE(hyper, "er_emb_front: the catch-up needs the step's er_opt_hyper");
    if (int rc = front_collect_one_row(g)) return rc;
  }
  // the lag-1 replay table of this step was built by the step prologue (er_decay_tables_set_prologue_build)?
  bool tables_done = false;
  for (int i = 0; i < n; ++i)
    if (groups[i]->tabs && groups[i]->tabs->prologue_build) tables_done = true;
  if (tables_done)
    for (int i = 0; i < n; ++i)
      ER_REQUIRE(!groups[i]->tabs || groups[i]->tabs->prologue_build, "er_emb_front: the groups of one call must share their decay tables");
  // the fused lookup rides on the LAST sort launch of the call
  int last_sort = -1;
  if (ff) {
    if (int rc = fwd_lazy_prepare(ff->plan, groups, n, hyper, s, &ff->lz)) return rc;
    for (int i = 0; i < n; ++i) {
      er_emb_group* g = groups[i];
      er_emb_group* l = g->leader;
      bool follows = false;
      if (l && emb_group_same_keys(g, l))
        for (int j = 0; j < i; ++j) follows = follows || groups[j] == l;
      if (!follows) last_sort = i;
    }
  }
  for (int i = 0; i < n; ++i) {
    er_emb_group* g = groups[i];
    er_emb_group* l = g->leader;
    bool follows = false;
    if (l && emb_group_same_keys(g, l))
      for (int j = 0; j < i; ++j) follows = follows || groups[j] == l;
    if (follows && l->front_epoch == l->sort_epoch && l->front_skip == (skip_one_row != 0)) {
      g->src = l;
      g->adopted_epoch = l->sort_epoch;
      g->sorted_valid = true;
      continue;
    }
    g->src = g;
    ++g->sort_epoch;
    const int P = g->seg_sort_pow2;
    const bool with_tables = g->tabs != nullptr && !tables_done;
    er::DecayTabDev tabs{};
    if (with_tables) tabs = g->tabs->dev;
    const int E = P > 4096 ? 8 : 4;
    const int threads = P / E;
    const int table_blocks = with_tables ? static_cast<int>(er::ceil_div(static_cast<int64_t>(tabs.K) * er::kWave, threads)) : 0;
    const size_t lds = sizeof(unsigned long long) * static_cast<size_t>(P);
    if (ff && i == last_sort && !with_tables && threads % er::kBlock == 0 && !ff->launched) {
      // sort + lookup in one launch (emb_front_fwd_kernel): the lookup's blocks behind the sort's workgroups
      er_emb_plan* pl = ff->plan;
      const int per_wg = threads / er::kBlock;
      const int fwd_wgs = static_cast<int>(er::ceil_div(pl->n_blocks, per_wg));
#define ER_FRONT_FWD(EE, NRW)                                                                                             \
  hipLaunchKernelGGL((er::emb_front_fwd_kernel<EE, NRW>), dim3(g->n + fwd_wgs), dim3(threads), lds, s, g->d_ent_base,      \
                     g->d_descs, g->n, P, g->keys_out, g->vals_out, g->head_flags, g->head_index, g->seg_count,          \
                     skip_one_row ? 1 : 0, g->keys_in, pl->d_descs, pl->d_blk_start, pl->n, pl->n_blocks, ff->sumsq, ff->lz)
      if (E == 8) { if (g->seg_narrow) ER_FRONT_FWD(8, true); else ER_FRONT_FWD(8, false); }
      else { if (g->seg_narrow) ER_FRONT_FWD(4, true); else ER_FRONT_FWD(4, false); }
#undef ER_FRONT_FWD
      ER_LAUNCH_CHECK();
      ff->launched = true;
    } else {
#define ER_FRONT_SORT(EE, NRW)                                                                                           \
  hipLaunchKernelGGL((er::emb_front_sort_kernel<EE, NRW>), dim3(g->n + table_blocks), dim3(threads), lds, s, g->d_ent_base, \
                     g->d_descs, g->n, P, g->keys_out, g->vals_out, g->head_flags, g->head_index, g->seg_count,         \
                     skip_one_row ? 1 : 0, tabs, with_tables ? g->tabs->hist : nullptr, with_tables ? g->tabs->counter : nullptr, \
                     g->keys_in)
    if (E == 8) { if (g->seg_narrow) ER_FRONT_SORT(8, true); else ER_FRONT_SORT(8, false); }
    else { if (g->seg_narrow) ER_FRONT_SORT(4, true); else ER_FRONT_SORT(4, false); }
#undef ER_FRONT_SORT
    ER_LAUNCH_CHECK();
    }
    tables_done = tables_done || with_tables;
    g->heads_epoch = g->sort_epoch;
    g->front_epoch = g->sort_epoch;
    g->front_skip = skip_one_row != 0;
    g->sorted_valid = true;
  }
  for (int i = 0; i < n; ++i) groups[i]->front_deferred = defer && groups[i]->last_step != nullptr;
  if (defer) {  // the lookup (er_emb_fwd_lazy) and the row update (er_emb_bwd_fused) catch the rows up in registers
    for (int i = 0; i < n; ++i)
      if (groups[i]->last_step) ER_REQUIRE(tables_done, "er_emb_front: no sort leader carried the decay tables");
    return 0;
  }
  // catch-up from the heads
  er::CatchHeadsMulti cm;
  cm.n = 0;
  cm.start[0] = 0;
  cm.hyper = hyper;
  const er_decay_tables* first_tabs = nullptr;
  for (int i = 0; i < n; ++i) {
    er_emb_group* g = groups[i];
    if (!g->last_step) continue;
    if (!first_tabs) first_tabs = g->tabs;
    ER_REQUIRE(g->tabs == first_tabs, "er_emb_front: the groups of one call must share their decay tables");
    const er_emb_group* src = g->src;
    er::CatchHeadsArgs& a = cm.a[cm.n];
    a.ukeys_seg = src->keys_in;  // (the unsorted key array is free in the fused front: the sort builds its entries itself)
    a.seg_count = src->seg_count; a.ent_base = src->d_ent_base; a.n_lookups = src->n; a.n = group_entries(g);
    a.tab = tab_of(g);
    a.aux = decay_aux_for(g, 1);
    a.dim = g->dim; a.G = g->G; a.V = g->V;
    a.extra_keys = skip_one_row ? g->d_extra_keys : nullptr;
    a.n_extra = skip_one_row ? g->n_proj : 0;
    a.cap_shift = group_cap_shift(src);
    // (one lane group per sorted position; + one workgroup for the one-row tables: at most 64 of them, G <= 4 lanes each...)
    ER_REQUIRE(a.n_extra * g->G <= er::kBlock, "er_emb_front: group %d: %d one-row tables of %d lanes exceed a workgroup", i, a.n_extra, g->G);
    cm.start[cm.n + 1] = cm.start[cm.n] + static_cast<int>(er::ceil_div(a.n, er::kBlock / g->G)) + (a.n_extra > 0 ? 1 : 0);
    ++cm.n;
  }
  if (cm.n > 0) {
    ER_REQUIRE(tables_done, "er_emb_front: no sort leader carried the decay tables");
    hipLaunchKernelGGL(er::emb_catch_up_heads_kernel, dim3(cm.start[cm.n]), dim3(er::kBlock), 0, s, cm);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

int er_emb_front(er_emb_group* const* groups, int n, int flags, const er_opt_hyper* hyper, er_stream_t stream) {
  return emb_front_impl(groups, n, flags, hyper, er::as_stream(stream), nullptr);
}

// probe hook: 16 wall-clock stamps (100 MHz) per workgroup of the next er_emb_bwd_fused launches into `p` (nullptr: off);
// tools/own_probe.py reads them.  Not part of the product path.
static unsigned long long* g_own_dbg = nullptr;
int er_debug_stamps(unsigned long long* p) { g_own_dbg = p; return 0; }

static int emb_bwd_fused_impl(er_emb_group* const* groups, int n, const er_grad_group* finish, int n_finish, int opt_kind,
                              const er_opt_hyper* hyper, const er_gemm_problem* wgrads, int n_wgrads, int wgrad_blocks,
                              const er_loss_tail_job* loss_tail, const er_dense_opt_job* dense_opt, er_stream_t stream) {
  ER_REQUIRE(groups && finish && hyper && n >= 1 && n <= er::kMaxMulti && n_finish >= 1 && n_finish <= er::kOwnMaxGG,
             "er_emb_bwd_fused: bad arguments (1 <= n <= %d groups, 1 <= n_finish <= %d)", er::kMaxMulti, er::kOwnMaxGG);
  ER_REQUIRE(opt_kind >= ER_OPT_SGD && opt_kind <= ER_OPT_ADAGRAD, "er_emb_bwd_fused: unknown optimizer %d", opt_kind);
  hipStream_t s = er::as_stream(stream);
  er::GroupedPlan plan;  // (n_wgrads > 0) the weight gradients' grouped launch: planned before any group state changes
  if (n_wgrads > 0) {
    ER_REQUIRE(wgrads && n_wgrads <= er::kMaxGroup, "er_emb_bwd_fused_wgrad: 1 <= n_wgrads <= %d", er::kMaxGroup);
    for (int i = 0; i < n_wgrads; ++i)
      ER_REQUIRE(!wgrads[i].col_stats && !wgrads[i].bn_partial,
                 "er_emb_bwd_fused_wgrad: problem %d: plain contractions only (no column statistics or BatchNorm epilogue)", i);
    if (int rc = er::plan_grouped(ER_GEMM_TN, wgrads, n_wgrads, false, &plan, wgrad_blocks)) return rc;
  }
  er::LossTailArgs lt;
  std::memset(&lt, 0, sizeof(lt));
  er::DenseOptArgs da;
  std::memset(&da, 0, sizeof(da));
  if (loss_tail || dense_opt) ER_REQUIRE(n_wgrads > 0, "er_emb_bwd_fused_tail: the riders need the weight gradients' launch (n_wgrads > 0)");
  if (loss_tail)
    if (int rc = er::make_loss_tail_args(loss_tail, &lt)) return rc;
  if (dense_opt) {
    if (int rc = er::make_dense_opt_args(dense_opt, &da)) return rc;
    for (int j = 0; j < plan.ra.n; ++j) {  // the k-split outputs the optimizer finishes while it reads them
      const er::ReduceItem& r = plan.ra.r[j];
      ER_REQUIRE(r.ldc == r.N && r.C >= da.grad && r.C + r.mn <= da.grad + da.n,
                 "er_emb_bwd_fused_tail: weight gradient %d is not a contiguous block of dense_opt->grad", j);
    }
    for (int i = 0; i < n_wgrads; ++i)  // (an unsplit problem writes C itself: it only has to be finished before the optimizer)
      ER_REQUIRE(wgrads[i].C, "er_emb_bwd_fused_tail: problem %d has no output", i);
  }
  er::OwnMulti ma;
  er::RunMulti fx;
  fx.n = 0;
  fx.start[0] = 0;
  fx.opt_kind = opt_kind;
  fx.hyper = hyper;
  ma.n = 0;
  ma.start[0] = 0;
  ma.proj_start[0] = 0;
  ma.opt_kind = opt_kind;
  ma.hyper = hyper;
  ma.n_gg = n_finish;
  ma.dbg = g_own_dbg;
  for (int k = 0; k < n_finish; ++k)
    ER_REQUIRE(finish[k].dout && finish[k].out && finish[k].n_terms >= 0 && finish[k].n_terms <= 4,
               "er_emb_bwd_fused: finish descriptor %d: bad arguments", k);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    er_emb_group* g = groups[i];
    ER_REQUIRE(g, "er_emb_bwd_fused: null group %d", i);
    const er_emb_group* src = g->src;
    ER_REQUIRE(g->sorted_valid && src && src->front_epoch == src->sort_epoch,
               "er_emb_bwd_fused: call er_emb_front for this step first (group %d)", i);
    if (opt_kind == ER_OPT_ADAM || opt_kind == ER_OPT_LAZY_ADAM) ER_REQUIRE(g->m && g->v, "er_emb_bwd_fused: Adam needs m and v");
    if (opt_kind == ER_OPT_ADAGRAD) ER_REQUIRE(g->v, "er_emb_bwd_fused: Adagrad needs the accumulator in v");
    ER_REQUIRE(opt_kind != ER_OPT_ADAM || g->last_step, "er_emb_bwd_fused: ER_OPT_ADAM needs lazy dense decay (the sweep form: er_emb_bwd_update)");
    // the lookups' gather records: which descriptor finishes each lookup's output block, and which of its terms cover it
    std::vector<er::OwnLookup> recs(g->n);
    for (int l = 0; l < g->n; ++l) {
      const er_lookup_desc& d = g->h_descs[l];
      int k_found = -1;
      for (int k = 0; k < n_finish; ++k)
        if (finish[k].dout == d.out && finish[k].ld == d.out_stride) k_found = k;
      ER_REQUIRE(k_found >= 0, "er_emb_bwd_fused: group %d lookup %d: no finish descriptor for its gradient buffer", i, l);
      const er_grad_group& fg = finish[k_found];
      ER_REQUIRE(d.n_rows <= fg.batch && d.out_col + d.dim <= fg.width,
                 "er_emb_bwd_fused: group %d lookup %d lies outside its gradient buffer", i, l);
      unsigned mask = 0;
      for (int t = 0; t < fg.n_terms; ++t) {
        const er_grad_term& q = fg.terms[t];
        const bool inside = d.out_col >= q.col0 && d.out_col + d.dim <= q.col0 + q.width;
        const bool outside = d.out_col + d.dim <= q.col0 || d.out_col >= q.col0 + q.width;
        ER_REQUIRE(inside || outside, "er_emb_bwd_fused: group %d lookup %d: a deferred term covers part of its columns", i, l);
        if (!inside) continue;
        if (q.kind == ER_GRAD_TERM_FM) {
          ER_REQUIRE(q.dim == d.dim && (d.out_col - q.col0) % q.dim == 0,
                     "er_emb_bwd_fused: group %d lookup %d is not one field of the FM term over its columns", i, l);
          mask |= 0x10u << t;
        }
        mask |= 1u << t;
      }
      er::OwnLookup& r = recs[l];
      std::memset(&r, 0, sizeof(r));
      r.weights = d.weights;
      r.base = static_cast<int32_t>(l == 0 ? 0 : 0);
      r.out_col = d.out_col;
      r.combiner = d.combiner;
      r.gg = static_cast<int8_t>(k_found);
      r.tmask = static_cast<uint8_t>(mask);
    }
    {  // entry offsets (dense-mode lookups: n_rows entries each)
      int64_t base = 0;
      for (int l = 0; l < g->n; ++l) {
        recs[l].base = static_cast<int32_t>(base);
        base += g->h_descs[l].offsets ? g->h_descs[l].max_nnz : g->h_descs[l].n_rows;
      }
    }
    const size_t rec_bytes = sizeof(er::OwnLookup) * recs.size();
    if (g->h_own_lookups.size() != rec_bytes || std::memcmp(g->h_own_lookups.data(), recs.data(), rec_bytes) != 0) {
      ER_REQUIRE(cap == hipStreamCaptureStatusNone, "er_emb_bwd_fused: run one step eagerly before capturing (the plan is uploaded on first use)");
      if (!g->d_own_lookups) ER_CHECK_HIP(hipMalloc(&g->d_own_lookups, sizeof(er::OwnLookup) * er::kOwnMaxLookups));
      ER_CHECK_HIP(hipStreamSynchronize(s));  // (an earlier launch may still read the old records)
      ER_CHECK_HIP(hipMemcpy(g->d_own_lookups, recs.data(), rec_bytes, hipMemcpyHostToDevice));
      g->h_own_lookups.assign(reinterpret_cast<const unsigned char*>(recs.data()), reinterpret_cast<const unsigned char*>(recs.data()) + rec_bytes);
    }
    er::OwnArgs& a = ma.a[ma.n];
    a.skeys = src->keys_out; a.svals = src->vals_out; a.n = group_entries(g);
    a.descs = g->d_descs; a.ent_base = g->d_ent_base; a.lookups = static_cast<const er::OwnLookup*>(g->d_own_lookups);
    a.n_lookups = g->n;
    a.cap_shift = group_cap_shift(g);
    a.dim = g->dim; a.G = g->G; a.V = g->V;
    const int T = g->tile_entries;
    a.n_tiles = static_cast<int>(er::ceil_div(a.n, T));
    a.tab = tab_of(g);
    a.aux = g->front_deferred ? decay_aux_for(g, 1) : er::DecayAux{};
    a.tile_first = g->tile_first; a.tile_last = g->tile_last;
    const bool proj = src->front_skip && g->n_proj > 0;
    a.n_proj = proj ? g->n_proj : 0;
    {  // the fix launch's arguments (emb_bwd_fix_multi_kernel: the three-launch path's kernel)
      er::RunArgs& f = fx.a[fx.n];
      f.skeys = src->keys_out; f.svals = src->vals_out; f.ent_gptr = nullptr; f.ent_scale = nullptr;
      f.n = a.n; f.dim = g->dim; f.G = g->G; f.V = g->V; f.T = T; f.n_tiles = a.n_tiles;
      f.tab = a.tab;
      f.aux = a.aux;
      f.ro = er::ReduceOut{0, nullptr, nullptr, nullptr, nullptr, 0};
      f.tile_first = g->tile_first; f.tile_last = g->tile_last;
      const int fb = a.n_tiles > 1 ? static_cast<int>(er::ceil_div(static_cast<int64_t>(a.n_tiles) * g->G, er::kBlock)) : 0;
      fx.start[fx.n + 1] = fx.start[fx.n] + fb;
      ++fx.n;
    }
    a.proj_lookup = g->d_proj_lookup; a.proj_partial = g->d_proj_partial; a.proj_ticket = g->d_proj_ticket;
    size_t need = sizeof(float) * static_cast<size_t>(T) * g->dim + sizeof(uint32_t) * (T + 4) +
                  sizeof(er::OwnLookup) * static_cast<size_t>(g->n) + sizeof(er_grad_group) * static_cast<size_t>(n_finish);
    const size_t rpp = er::kBlock / g->G;
    const size_t need_proj = sizeof(float) * (rpp * g->dim + rpp);
    if (need_proj > need) need = need_proj;
    if (need > lds) lds = need;
    g->sorted_valid = false;
    ma.start[ma.n + 1] = ma.start[ma.n] + a.n_tiles;
    ma.proj_start[ma.n + 1] = ma.proj_start[ma.n] + a.n_proj * er::kProjParts;
    g->front_deferred = false;
    ++ma.n;
  }
  for (int k = 0; k < n_finish; ++k) ma.gg[k] = finish[k];
  const int grid = ma.start[ma.n] + ma.proj_start[ma.n];
  if (n_wgrads > 0) {
    const int n_gemm = er::grouped_grid(plan.ga);
    const size_t gemm_lds = sizeof(float) * 2 * 2 * er::kOpTile;
    hipLaunchKernelGGL(er::emb_bwd_own_wgrad_kernel, dim3(n_gemm + grid + (loss_tail ? 1 : 0)), dim3(er::kBlock),
                       lds > gemm_lds ? lds : gemm_lds, s, ma, plan.ga, n_gemm, grid, lt);
    ER_LAUNCH_CHECK();
    const int n_fix = grid > 0 ? fx.start[fx.n] : 0;
    if (dense_opt) {
      const int n_opt = static_cast<int>(er::ceil_div(da.n, er::kBlock));
      hipLaunchKernelGGL(er::emb_bwd_fix_opt_kernel, dim3(n_fix + n_opt), dim3(er::kBlock), 0, s, fx, plan.ra, da, n_fix);
      ER_LAUNCH_CHECK();
      return 0;
    }
    const int n_red = plan.ra.n > 0 ? plan.ra.start[plan.ra.n] : 0;
    if (n_fix + n_red > 0) {
      hipLaunchKernelGGL(er::emb_bwd_fix_reduce_kernel, dim3(n_fix + n_red), dim3(er::kBlock), 0, s, fx, plan.ra, n_fix);
      ER_LAUNCH_CHECK();
    }
    return 0;
  }
  if (grid > 0) {
    hipLaunchKernelGGL(er::emb_bwd_own_kernel, dim3(grid), dim3(er::kBlock), lds, s, ma);
    ER_LAUNCH_CHECK();
    if (fx.start[fx.n] > 0) {
      hipLaunchKernelGGL(er::emb_bwd_fix_multi_kernel, dim3(fx.start[fx.n]), dim3(er::kBlock), 0, s, fx);
      ER_LAUNCH_CHECK();
    }
  }
  return 0;
}

int er_emb_bwd_fused(er_emb_group* const* groups, int n, const er_grad_group* finish, int n_finish, int opt_kind,
                     const er_opt_hyper* hyper, er_stream_t stream) {
  return emb_bwd_fused_impl(groups, n, finish, n_finish, opt_kind, hyper, nullptr, 0, 0, nullptr, nullptr, stream);
}

int er_emb_bwd_fused_wgrad(er_emb_group* const* groups, int n, const er_grad_group* finish, int n_finish, int opt_kind,
                           const er_opt_hyper* hyper, const er_gemm_problem* wgrads, int n_wgrads, int32_t wgrad_blocks,
                           er_stream_t stream) {
  ER_REQUIRE(wgrads && n_wgrads >= 1 && wgrad_blocks >= 0, "er_emb_bwd_fused_wgrad: no weight-gradient problems (er_emb_bwd_fused)");
  return emb_bwd_fused_impl(groups, n, finish, n_finish, opt_kind, hyper, wgrads, n_wgrads, wgrad_blocks, nullptr, nullptr, stream);
}

int er_emb_bwd_fused_tail(er_emb_group* const* groups, int n, const er_grad_group* finish, int n_finish, int opt_kind,
                          const er_opt_hyper* hyper, const er_gemm_problem* wgrads, int n_wgrads, int32_t wgrad_blocks,
                          const er_loss_tail_job* loss_tail, const er_dense_opt_job* dense_opt, er_stream_t stream) {
  ER_REQUIRE(wgrads && n_wgrads >= 1 && wgrad_blocks >= 0, "er_emb_bwd_fused_tail: no weight-gradient problems (er_emb_bwd_fused)");
  return emb_bwd_fused_impl(groups, n, finish, n_finish, opt_kind, hyper, wgrads, n_wgrads, wgrad_blocks, loss_tail, dense_opt, stream);
}
#undef ER_ELIGIBLE

// the lazy lookup's arguments: which group every lookup of the plan reads (uploaded when it changes), the groups' records.
// deferred_now: the groups are being deferred by the front of this very call (er_emb_front_fwd)
static int fwd_lazy_prepare_impl(er_emb_plan* p, er_emb_group* const* groups, int n, const er_opt_hyper* hyper, hipStream_t s,
                                 er::FwdLazy* out, bool deferred_now) {
  ER_REQUIRE(p && groups && n >= 1 && n <= er::kMaxMulti, "er_emb_fwd_lazy: bad arguments (1 <= n <= %d)", er::kMaxMulti);
  er::FwdLazy& lz = *out;
  lz.n = 0;
  lz.hyper = hyper;
  lz.lag = nullptr;
  lz.counter = nullptr;
  // which group a lookup reads: its table lies inside the group's var rows (and has the group's dim)
  std::vector<int8_t> map(p->n, -1);
  int slot_of[er::kMaxMulti];
  for (int i = 0; i < n; ++i) {
    er_emb_group* g = groups[i];
    ER_REQUIRE(g, "er_emb_fwd_lazy: null group %d", i);
    slot_of[i] = -1;
    if (!g->last_step) continue;
    ER_REQUIRE(hyper && g->tabs && g->G <= er::kWave && (g->front_deferred || deferred_now),
               "er_emb_fwd_lazy: group %d: call er_emb_front(ER_FRONT_DEFER_CATCH_UP) for this step first (closed-form replay only)", i);
    slot_of[i] = lz.n;
    lz.tab[lz.n] = tab_of(g);
    lz.aux[lz.n] = decay_aux_for(g, 1);
    if (g->tabs->prologue_build) {  // (the lookup launch keeps the prologue's counter word and checks the table's stamp)
      lz.lag = g->tabs->dev.lag;
      lz.counter = g->tabs->counter;
    }
    ++lz.n;
  }
  for (int l = 0; l < p->n; ++l) {
    const er_lookup_desc& d = p->h_descs[l];
    for (int i = 0; i < n; ++i) {
      const er_emb_group* g = groups[i];
      if (slot_of[i] < 0 || d.dim != g->dim) continue;
      const float* lo = g->var;
      const float* hi = g->var + g->total_rows * g->ld;
      if (d.table < lo || d.table >= hi) continue;
      ER_REQUIRE(d.table == g->var + d.key_base * g->ld && (d.table_ld ? d.table_ld : d.dim) == g->ld,
                 "er_emb_fwd_lazy: lookup %d does not address group %d's rows by its key_base / row pitch", l, i);
      map[l] = static_cast<int8_t>(slot_of[i]);
    }
  }
  if (map != p->h_lookup_group) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    ER_REQUIRE(cap == hipStreamCaptureStatusNone, "er_emb_fwd_lazy: run one step eagerly before capturing (the plan is uploaded on first use)");
    if (!p->d_lookup_group) ER_CHECK_HIP(hipMalloc(&p->d_lookup_group, sizeof(int8_t) * p->n));
    ER_CHECK_HIP(hipStreamSynchronize(s));
    ER_CHECK_HIP(hipMemcpy(p->d_lookup_group, map.data(), sizeof(int8_t) * p->n, hipMemcpyHostToDevice));
    p->h_lookup_group = map;
  }
  lz.lookup_group = p->d_lookup_group;
  return 0;
}

static int fwd_lazy_prepare(er_emb_plan* p, er_emb_group* const* groups, int n, const er_opt_hyper* hyper, hipStream_t s,
                            er::FwdLazy* lz) {
  return fwd_lazy_prepare_impl(p, groups, n, hyper, s, lz, true);
}

int er_emb_fwd_lazy(er_emb_plan* p, er_emb_group* const* groups, int n, const er_opt_hyper* hyper, float* sumsq_partials,
                    er_stream_t stream) {
  hipStream_t s = er::as_stream(stream);
  er::FwdLazy lz;
  if (int rc = fwd_lazy_prepare_impl(p, groups, n, hyper, s, &lz, false)) return rc;
  hipLaunchKernelGGL(er::emb_fwd_lazy_kernel, dim3(p->n_blocks), dim3(er::kBlock), 0, s, p->d_descs, p->d_blk_start, p->n,
                     sumsq_partials, lz);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_emb_front_fwd(er_emb_group* const* groups, int n, int flags, er_emb_plan* plan, const er_opt_hyper* hyper,
                     float* sumsq_partials, er_stream_t stream) {
  ER_REQUIRE(plan && (flags & ER_FRONT_DEFER_CATCH_UP), "er_emb_front_fwd: needs a plan and ER_FRONT_DEFER_CATCH_UP");
  hipStream_t s = er::as_stream(stream);
  FrontFwd ff;
  ff.plan = plan;
  ff.sumsq = sumsq_partials;
  ff.launched = false;
  if (int rc = emb_front_impl(groups, n, flags, hyper, s, &ff)) return rc;
  if (!ff.launched) {  // (no sort launch could carry it: the lookup as a launch of its own)
    hipLaunchKernelGGL(er::emb_fwd_lazy_kernel, dim3(plan->n_blocks), dim3(er::kBlock), 0, s, plan->d_descs, plan->d_blk_start,
                       plan->n, sumsq_partials, ff.lz);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

int er_emb_catch_up(er_emb_group* g, const uint32_t* unique_keys, const int32_t* n_unique, const er_opt_hyper* hyper,
                    er_stream_t stream) {
  ER_REQUIRE(g && unique_keys && n_unique && hyper, "er_emb_catch_up: null argument");
  ER_REQUIRE(g->last_step, "er_emb_catch_up: call er_emb_group_enable_lazy_decay first");
  if (g->tabs) return er_emb_catch_up_multi(&g, &unique_keys, &n_unique, 1, hyper, stream);
  const int64_t cap = group_entries(g);
  if (cap == 0) return 0;
  const er::RowUpdate tab = tab_of(g);
  const int blocks = static_cast<int>(er::ceil_div(cap * g->G, er::kBlock));
  if (g->V == 4) {
    hipLaunchKernelGGL(er::emb_catch_up_kernel<4>, dim3(blocks), dim3(er::kBlock), 0, er::as_stream(stream), unique_keys,
                       n_unique, cap, tab, g->lr_hist, hyper, g->dim, g->G, decay_aux_for(g, 1));
  } else {
    hipLaunchKernelGGL(er::emb_catch_up_kernel<1>, dim3(blocks), dim3(er::kBlock), 0, er::as_stream(stream), unique_keys,
                       n_unique, cap, tab, g->lr_hist, hyper, g->dim, g->G, decay_aux_for(g, 1));
  }
  ER_LAUNCH_CHECK();
  return 0;
}

int er_emb_catch_up_multi(er_emb_group* const* groups, const uint32_t* const* unique_keys,
                          const int32_t* const* n_unique, int n, const er_opt_hyper* hyper, er_stream_t stream) {
  ER_REQUIRE(groups && unique_keys && n_unique && hyper && n >= 1 && n <= er::kMaxMulti,
             "er_emb_catch_up_multi: bad arguments (1 <= n <= %d)", er::kMaxMulti);
  if (n == 1 && !(groups[0] && groups[0]->tabs)) return er_emb_catch_up(groups[0], unique_keys[0], n_unique[0], hyper, stream);
  er::CatchUpMulti ma;
  bool closed = true;  // every group evaluates the closed form and keeps a row's lanes in one wavefront
  ma.n = 0;
  ma.start[0] = 0;
  ma.hyper = hyper;
  for (int i = 0; i < n; ++i) {
    er_emb_group* g = groups[i];
    ER_REQUIRE(g && unique_keys[i] && n_unique[i], "er_emb_catch_up_multi: null argument (group %d)", i);
    ER_REQUIRE(g->last_step, "er_emb_catch_up_multi: call er_emb_group_enable_lazy_decay first (group %d)", i);
    const int64_t cap = group_entries(g);
    if (cap == 0) continue;
    closed = closed && g->tabs != nullptr && g->G <= er::kWave;
    er::CatchUpArgs& a = ma.a[ma.n];
    a.ukeys = unique_keys[i]; a.n_unique = n_unique[i]; a.capacity = cap;
    a.tab = tab_of(g);
    a.lr_hist = g->lr_hist; a.aux = decay_aux_for(g, 1); a.dim = g->dim; a.G = g->G; a.V = g->V;
    ma.start[ma.n + 1] = ma.start[ma.n] + static_cast<int>(er::ceil_div(cap * g->G, er::kBlock));
    ++ma.n;
  }
  if (ma.n == 0) return 0;
  if (int rc = launch_decay_tables(groups, n, 1, er::as_stream(stream))) return rc;
  if (closed) hipLaunchKernelGGL(er::emb_catch_up_closed_kernel, dim3(ma.start[ma.n]), dim3(er::kBlock), 0, er::as_stream(stream), ma);
  else hipLaunchKernelGGL(er::emb_catch_up_multi_kernel, dim3(ma.start[ma.n]), dim3(er::kBlock), 0, er::as_stream(stream), ma);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_emb_flush_decay(er_emb_group* g, const er_opt_hyper* hyper, er_stream_t stream) {
  ER_REQUIRE(g && hyper, "er_emb_flush_decay: null argument");
  ER_REQUIRE(g->last_step, "er_emb_flush_decay: call er_emb_group_enable_lazy_decay first");
  hipStream_t s = er::as_stream(stream);
  const er::RowUpdate tab = tab_of(g);
  const int64_t blocks = er::ceil_div(g->total_rows * g->G, er::kBlock);
  ER_REQUIRE(blocks < 0x7FFFFFFFLL, "er_emb_flush_decay: table group too large for one launch");
  if (int rc = launch_decay_tables(&g, 1, 0, s)) return rc;
  if (g->V == 4) {
    hipLaunchKernelGGL(er::emb_flush_decay_kernel<4>, dim3(static_cast<unsigned>(blocks)), dim3(er::kBlock), 0, s, tab,
                       g->total_rows, g->lr_hist, hyper, g->dim, g->G, g->aux);
  } else {
    hipLaunchKernelGGL(er::emb_flush_decay_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(er::kBlock), 0, s, tab,
                       g->total_rows, g->lr_hist, hyper, g->dim, g->G, g->aux);
  }
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::emb_flush_mark_kernel, dim3(1024), dim3(er::kBlock), 0, s, g->last_step, g->ls_ld, g->total_rows,
                     g->step_counter);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_emb_group_set_lr_max(er_emb_group* g, const float* lr_max_history) {
  ER_REQUIRE(g, "er_emb_group_set_lr_max: null group");
  g->aux.lr_max = lr_max_history;
  return 0;
}

// ---- closed-form replay tables (er_decay.h) ----
static int decay_terms(double beta1) {  // smallest multiple of 64 with beta1^K <= 2e-9
  if (!(beta1 > 0.0 && beta1 < 1.0)) return 0;
  const double k = std::ceil(std::log(2e-9) / std::log(beta1));
  const int K = static_cast<int>((static_cast<int64_t>(k) + 63) / 64 * 64);
  return K < 64 ? 64 : K;
}

int er_decay_tables_supported(float beta1, float beta2) {
  const int K = decay_terms(static_cast<double>(beta1));
  if (K <= 0 || K > er::kDecayKMax || !(beta2 > 0.f && beta2 < 1.f)) return 0;
  // remainder of the w-expansion relative to the whole update: sum_s beta1^s w_s^N / sum_s beta1^s, w_s = 1 - sqrt(beta2)^s
  const double q = std::sqrt(static_cast<double>(beta2)), b1 = static_cast<double>(beta1);
  double num = 0.0, den = 0.0;
  for (int s = 1; s <= K; ++s) {
    const double w = -std::expm1(s * std::log(q)), p = std::pow(b1, s);
    num += p * std::pow(w, er::kDecayN) / (1.0 - w);
    den += p;
  }
  return num / den <= 3e-8 ? K : 0;
}

int64_t er_decay_tables_bytes(int64_t history_capacity) {
  if (history_capacity <= 0) return -1;
  return static_cast<int64_t>(er::kDecayKMax) * er::kDecayLd * (sizeof(double) + 2 * sizeof(float)) +  // coef, A[lag 0 | 1]
         (history_capacity + 1) * er::kDecayLd * static_cast<int64_t>(sizeof(float)) + 64;  // + the lag words
}

int er_decay_tables_create(void* buffer, int64_t history_capacity, const float* lr_t_history, const int64_t* step_counter,
                           float beta1, float beta2, er_decay_tables** out) {
  ER_REQUIRE(buffer && lr_t_history && step_counter && out && history_capacity > 0, "er_decay_tables_create: bad arguments");
  ER_REQUIRE((reinterpret_cast<uintptr_t>(buffer) & 15) == 0, "er_decay_tables_create: buffer must be 16-byte aligned");
  const int K = er_decay_tables_supported(beta1, beta2);
  ER_REQUIRE(K > 0, "er_decay_tables_create: beta1 %g / beta2 %g are outside the closed form's range (use the exact replay)",
             static_cast<double>(beta1), static_cast<double>(beta2));
  std::vector<double> coef(static_cast<size_t>(er::kDecayKMax) * er::kDecayLd, 0.0);
  const double b1 = static_cast<double>(beta1), lq = 0.5 * std::log(static_cast<double>(beta2));
  for (int s = 1; s <= K; ++s) {
    const double p = std::exp(s * std::log(b1)), w = -std::expm1(s * lq);
    double wn = 1.0;
    for (int n = 0; n < er::kDecayN; ++n, wn *= w) coef[static_cast<size_t>(s - 1) * er::kDecayLd + n] = p * wn;
  }
  er_decay_tables* t = new er_decay_tables();
  char* base = static_cast<char*>(buffer);
  t->dev.coef = reinterpret_cast<const double*>(base);
  base += static_cast<size_t>(er::kDecayKMax) * er::kDecayLd * sizeof(double);
  t->dev.A = reinterpret_cast<float*>(base);  // two tables back to back: lag 0, lag 1 (er::decay_aux_for)
  base += 2 * static_cast<size_t>(er::kDecayKMax) * er::kDecayLd * sizeof(float);
  t->dev.C = reinterpret_cast<float*>(base);
  base += (static_cast<size_t>(history_capacity) + 1) * er::kDecayLd * sizeof(float);
  t->dev.lag = reinterpret_cast<int64_t*>(base);  // (8-byte aligned: every block before it is a multiple of 32 bytes)
  t->dev.K = K;
  t->dev.capacity = history_capacity;
  t->hist = lr_t_history;
  t->counter = step_counter;
  t->beta1 = beta1;
  t->beta2 = beta2;
  t->ln_b1 = std::log(b1);
  t->ln_b2 = std::log(static_cast<double>(beta2));
  hipError_t e = hipMemcpy(buffer, coef.data(), coef.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemset(t->dev.A, 0, (2 * static_cast<size_t>(er::kDecayKMax) + static_cast<size_t>(history_capacity) + 1) * er::kDecayLd * sizeof(float));
  if (e == hipSuccess) {
    const int64_t init[3] = {0, -1, 0};  // (lag[0] follows the counter from the first er_decay_tables_sync / lookup on)
    e = hipMemcpy(t->dev.lag, init, sizeof(init), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipMemcpy(t->dev.lag, step_counter, sizeof(int64_t), hipMemcpyDeviceToDevice);
  if (e != hipSuccess) {
    delete t;
    er::set_error("er_decay_tables_create: %s", hipGetErrorString(e));
    return 1;
  }
  *out = t;
  return 0;
}

int er_decay_tables_set_prologue_build(er_decay_tables* t, int on) {
  ER_REQUIRE(t, "er_decay_tables_set_prologue_build: null tables");
  t->prologue_build = on != 0;
  return 0;
}

int er_decay_tables_sync(er_decay_tables* t, er_stream_t stream) {
  ER_REQUIRE(t, "er_decay_tables_sync: null tables");
  hipLaunchKernelGGL(er::decay_lag_sync_kernel, dim3(1), dim3(64), 0, er::as_stream(stream), t->dev.lag, t->counter);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_decay_tables_error(er_decay_tables* t, int32_t* error_host) {
  ER_REQUIRE(t && error_host, "er_decay_tables_error: null argument");
  int64_t v = 0;
  ER_CHECK_HIP(hipMemcpy(&v, t->dev.lag + 2, sizeof(v), hipMemcpyDeviceToHost));
  *error_host = v != 0 ? 1 : 0;
  return 0;
}

int er_decay_tables_destroy(er_decay_tables* t) {
  delete t;
  return 0;
}

int er_emb_group_set_decay_tables(er_emb_group* g, er_decay_tables* t) {
  ER_REQUIRE(g, "er_emb_group_set_decay_tables: null group");
  ER_REQUIRE(!t || (g->last_step && g->lr_hist == t->hist && g->step_counter == t->counter),
             "er_emb_group_set_decay_tables: call er_emb_group_enable_lazy_decay with the tables' history and counter first");
  g->tabs = t;
  g->aux.A = t ? t->dev.A : nullptr;
  g->aux.C = t ? t->dev.C : nullptr;
  g->aux.K = t ? t->dev.K : 0;
  g->aux.ln_b1 = t ? t->ln_b1 : 0.0;
  g->aux.ln_b2 = t ? t->ln_b2 : 0.0;
  return 0;
}

int er_emb_flush_window(er_emb_group* const* groups, int n, int32_t n_windows, int32_t lag, int32_t max_blocks,
                        const er_opt_hyper* hyper, er_stream_t stream) {
  ER_REQUIRE(groups && hyper && n >= 1 && n <= er::kMaxMulti && n_windows >= 1 && (lag == 0 || lag == 1),
             "er_emb_flush_window: bad arguments (1 <= n <= %d, n_windows >= 1, lag 0 or 1)", er::kMaxMulti);
  er::FlushWindowMulti ma;
  ma.n = 0;
  ma.n_windows = n_windows;
  ma.lag = lag;
  ma.start[0] = 0;
  ma.hyper = hyper;
  for (int i = 0; i < n; ++i) {
    er_emb_group* g = groups[i];
    ER_REQUIRE(g && g->last_step, "er_emb_flush_window: call er_emb_group_enable_lazy_decay first (group %d)", i);
    ER_REQUIRE(g->G <= er::kWave, "er_emb_flush_window: a row must fit one wavefront");
    er::FlushWindowArgs& a = ma.a[ma.n];
    a.tab = tab_of(g);
    a.total_rows = g->total_rows;
    a.chunk = er::ceil_div(g->total_rows, n_windows);
    a.lr_hist = g->lr_hist; a.aux = decay_aux_for(g, lag); a.dim = g->dim; a.G = g->G; a.V = g->V;
    ma.start[ma.n + 1] = ma.start[ma.n] + static_cast<int>(er::ceil_div(a.chunk * g->G, er::kBlock));
    ++ma.n;
  }
  if (ma.start[ma.n] == 0) return 0;
  if (int rc = launch_decay_tables(groups, n, lag, er::as_stream(stream))) return rc;
  const int grid = max_blocks > 0 && max_blocks < ma.start[ma.n] ? max_blocks : ma.start[ma.n];
  hipLaunchKernelGGL(er::emb_flush_window_kernel, dim3(grid), dim3(er::kBlock), 0, er::as_stream(stream), ma);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_stream_copy(const void* src, void* dst, int64_t bytes, er_stream_t stream) {
  ER_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0, "er_stream_copy: bytes must be a positive multiple of 16");
  ER_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0,
             "er_stream_copy: pointers must be 16-byte aligned");
  const int64_t units = bytes / 16;
  int64_t blocks = er::ceil_div(units, static_cast<int64_t>(er::kBlock) * 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(er::stream_copy_kernel<4>, dim3(static_cast<int>(blocks)), dim3(er::kBlock), 0,
                     er::as_stream(stream), reinterpret_cast<const er::f32x4*>(src), reinterpret_cast<er::f32x4*>(dst),
                     units);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_emb_mark_touched(er_emb_group* g, er_stream_t stream) {
  ER_REQUIRE(g && g->bitmap, "er_emb_mark_touched: group has no touched bitmap");
  const int64_t n_act = g->n_active >= 0 ? g->n_active : INT64_MAX;
  hipLaunchKernelGGL(er::emb_mark_kernel, dim3(g->n_build_blocks), dim3(er::kBlock), 0, er::as_stream(stream),
                     g->d_descs, g->d_blk_start, g->n, n_act, g->bitmap);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_emb_sweep_untouched(er_emb_group* g, const er_opt_hyper* hyper, er_stream_t stream) {
  ER_REQUIRE(g && hyper && g->bitmap && g->m && g->v, "er_emb_sweep_untouched: needs bitmap, m and v");
  if (int rc = er_adam_decay_sweep_ld(g->var, g->m, g->v, g->bitmap, g->total_rows, g->dim, g->ld, hyper, stream)) return rc;
  return fill_u32(g->bitmap, 0u, er::ceil_div(g->total_rows, 32), er::as_stream(stream));
}

int er_emb_bwd_reduce(er_emb_group* g, uint32_t* unique_keys, float* unique_grads, int32_t* n_unique,
                      er_stream_t stream) {
  ER_REQUIRE(g && unique_keys && unique_grads && n_unique, "er_emb_bwd_reduce: null argument");
  hipStream_t s = er::as_stream(stream);
  if (group_entries(g) == 0) {
    return fill_u32(reinterpret_cast<uint32_t*>(n_unique), 0u, 1, s);
  }
  if (int rc = emb_group_sort(g, s)) return rc;
  if (int rc = emb_group_heads(g, n_unique, s)) return rc;
  return emb_group_run(g, ER_OPT_SGD, nullptr, 1, unique_keys, unique_grads, s);
}

int er_emb_apply_unique(er_emb_group* g, const uint32_t* unique_keys, const float* unique_grads, int32_t ld,
                        const int32_t* n_unique, int opt_kind, const er_opt_hyper* hyper, er_stream_t stream) {
  ER_REQUIRE(g && unique_keys && unique_grads && n_unique && hyper, "er_emb_apply_unique: null argument");
  ER_REQUIRE(opt_kind >= ER_OPT_SGD && opt_kind <= ER_OPT_ADAGRAD, "er_emb_apply_unique: unknown optimizer %d", opt_kind);
  ER_REQUIRE(ld == 0 || ld >= g->dim, "er_emb_apply_unique: ld < dim");
  if (opt_kind == ER_OPT_ADAM || opt_kind == ER_OPT_LAZY_ADAM) ER_REQUIRE(g->m && g->v, "er_emb_apply_unique: Adam needs m and v");
  if (opt_kind == ER_OPT_ADAGRAD) ER_REQUIRE(g->v, "er_emb_apply_unique: Adagrad needs the accumulator in v");
  const bool lazy_decay = (opt_kind == ER_OPT_ADAM) && g->last_step;
  if (opt_kind == ER_OPT_ADAM && !lazy_decay)
    ER_REQUIRE(g->bitmap, "er_emb_apply_unique: ER_OPT_ADAM needs touched_bitmap (or er_emb_group_enable_lazy_decay)");
  hipStream_t s = er::as_stream(stream);
  const int64_t cap = group_entries(g);
  if (cap > 0) {
    const er::RowUpdate tab = tab_of(g);
    const int blocks = static_cast<int>(er::ceil_div(cap * g->G, er::kBlock));
    const int row_ld = ld ? ld : g->dim;
    if (g->V == 4) {
      hipLaunchKernelGGL(er::emb_apply_unique_kernel<4>, dim3(blocks), dim3(er::kBlock), 0, s, unique_keys, unique_grads,
                         row_ld, n_unique, cap, tab, opt_kind, hyper, g->dim, g->G);
    } else {
      hipLaunchKernelGGL(er::emb_apply_unique_kernel<1>, dim3(blocks), dim3(er::kBlock), 0, s, unique_keys, unique_grads,
                         row_ld, n_unique, cap, tab, opt_kind, hyper, g->dim, g->G);
    }
    ER_LAUNCH_CHECK();
  }
  g->sorted_valid = false;  // (a sort left by this step's er_emb_route has served its purpose)
  if (opt_kind == ER_OPT_ADAM && !lazy_decay) {  // TF-exact Adam with the streaming sweep, as er_emb_bwd_update
    if (int rc = er_adam_decay_sweep_ld(g->var, g->m, g->v, g->bitmap, g->total_rows, g->dim, g->ld, hyper, stream)) return rc;
    if (int rc = fill_u32(g->bitmap, 0u, er::ceil_div(g->total_rows, 32), s)) return rc;
  }
  return 0;
}

int er_emb_group_set_routing(er_emb_group* g, int32_t world, int64_t shard_stride, const int64_t* local_base_host) {
  ER_REQUIRE(g && world >= 1 && local_base_host, "er_emb_group_set_routing: bad arguments");
  ER_REQUIRE(shard_stride > 0 && static_cast<int64_t>(world) * shard_stride < 0xFFFFFFFFLL,
             "er_emb_group_set_routing: world * shard_stride = %lld does not fit 32-bit keys",
             (long long)(world * shard_stride));
  if (!g->d_local_base) ER_CHECK_HIP(hipMalloc(&g->d_local_base, sizeof(int64_t) * g->n));
  ER_CHECK_HIP(hipMemcpy(g->d_local_base, local_base_host, sizeof(int64_t) * g->n, hipMemcpyHostToDevice));
  g->h_local_base.assign(local_base_host, local_base_host + g->n);
  // segmented sort of routed keys: every lookup's shard range [local_base, local_base + ceil(rows / world)) must be
  // its own and the ranges increasing with the lookup (one table per lookup), world <= 64 owners
  g->seg_routed = g->seg_caps_pow2 > 0 && world <= 64;
  g->seg_routed_narrow = g->seg_routed;
  for (int i = 0; i < g->n && g->seg_routed; ++i) {
    const int64_t srows = (g->h_descs[i].rows + world - 1) / world;
    if (local_base_host[i] < 0 || local_base_host[i] + srows > shard_stride) g->seg_routed = false;
    if (i + 1 < g->n && local_base_host[i] + srows > local_base_host[i + 1]) g->seg_routed = false;
    if ((static_cast<uint64_t>(srows) * world + 2) * static_cast<uint64_t>(g->seg_caps_pow2) > (1ull << 32))
      g->seg_routed_narrow = false;
  }
  g->world = world;
  g->shard_stride = shard_stride;
  const int64_t span = static_cast<int64_t>(world) * shard_stride;
  g->key_bits = 1;
  while ((1LL << g->key_bits) <= span) ++g->key_bits;
  // the temporary storage of the radix sort does not depend on the bit range
  return 0;
}

int er_emb_group_share_sort(er_emb_group* g, er_emb_group* leader) {
  ER_REQUIRE(g && g != leader, "er_emb_group_share_sort: bad arguments");
  if (!leader) {
    if (g->leader) {
      auto& fl = g->leader->followers;
      fl.erase(std::remove(fl.begin(), fl.end(), g), fl.end());
    }
    g->leader = nullptr;
    g->src = g;
    return 0;
  }
  ER_REQUIRE(!leader->leader, "er_emb_group_share_sort: the leader itself follows another group");
  ER_REQUIRE(emb_group_same_keys(g, leader),
             "er_emb_group_share_sort: the groups do not read the same ids with the same table geometry");
  if (g->leader && g->leader != leader) {
    auto& fl = g->leader->followers;
    fl.erase(std::remove(fl.begin(), fl.end(), g), fl.end());
  }
  g->leader = leader;
  g->adopted_epoch = leader->sort_epoch;
  if (std::find(leader->followers.begin(), leader->followers.end(), g) == leader->followers.end())
    leader->followers.push_back(g);
  return 0;
}

int er_emb_group_set_active(er_emb_group* g, int64_t n_rows) {
  ER_REQUIRE(g, "er_emb_group_set_active: null group");
  ER_REQUIRE(g->n == 1 && !g->has_ragged, "er_emb_group_set_active: needs a group of ONE dense-mode lookup");
  ER_REQUIRE(n_rows >= 0 && n_rows <= g->n_entries, "er_emb_group_set_active: %lld rows exceed the capacity %lld",
             (long long)n_rows, (long long)g->n_entries);
  g->n_active = n_rows;
  return 0;
}

int er_emb_route(er_emb_group* g, uint32_t* unique_keys, int32_t* n_unique, int64_t* entry_unique_index,
                 int32_t* owner_counts, er_stream_t stream) {
  ER_REQUIRE(g, "er_emb_route: null argument");
  ER_REQUIRE(g->world <= 64, "er_emb_route: world %d > 64", g->world);
  hipStream_t s = er::as_stream(stream);
  const int64_t N = group_entries(g);
  ER_REQUIRE(N > 0, "er_emb_route: empty group");
  bool adopted = false, fused_route = false, counts_done = false;
  if (g->leader)
    if (int rc = emb_group_adopt(g, s, &adopted)) return rc;
  if (adopted) {
    // shared sort: the leader's er_emb_route already produced the run heads; the outputs are optional here
    // (they equal the leader's)
    const er_emb_group* l = g->src;
    ER_REQUIRE(l->heads_epoch == l->sort_epoch, "er_emb_route: shared sort: call er_emb_route on the leader first");
    ER_REQUIRE((unique_keys != nullptr) == (n_unique != nullptr) && (unique_keys || (!entry_unique_index && !owner_counts)),
               "er_emb_route: shared sort: pass unique_keys AND n_unique, or neither (then no other output)");
    if (unique_keys) {
      // (routed per-lookup sort: the leader's head indices are in send order, not a running count)
      ER_REQUIRE(!(l->d_local_base && emb_group_segmented(l)),
                 "er_emb_route: shared sort of routed keys: the de-duplicated keys are the leader's outputs, pass none");
      hipLaunchKernelGGL(er::emb_count_unique_kernel, dim3(1), dim3(64), 0, s, l->head_flags, l->head_index, N, n_unique);
      ER_LAUNCH_CHECK();
    }
  } else {
    ER_REQUIRE(unique_keys && n_unique, "er_emb_route: null argument");
    ER_REQUIRE(g->peer_cap == 0 || g->d_local_base, "er_emb_route: a per-peer capacity needs routed keys");
    if (emb_group_segmented(g)) {
      // two launches: sort + in-lookup heads, then the cross-lookup offsets + key list + per-entry index
      if (int rc = emb_group_build_sort(g, s, true)) return rc;
      if (g->d_local_base) {
        dim3 grid(static_cast<unsigned>(er::ceil_div(g->seg_caps_pow2, er::kBlock)), static_cast<unsigned>(g->n));
        hipLaunchKernelGGL(er::emb_route_seg_routed_kernel, grid, dim3(er::kBlock), 0, s, g->keys_out, g->vals_out,
                           g->head_flags, g->head_index, g->d_ent_base, g->seg_count, g->n, g->world,
                           static_cast<uint32_t>(g->shard_stride), unique_keys, entry_unique_index, n_unique,
                           owner_counts, static_cast<uint32_t>(g->peer_cap), static_cast<uint32_t>(g->peer_hdr),
                           g->d_overflow);
        counts_done = true;
      } else {
        dim3 grid(static_cast<unsigned>(er::ceil_div(g->seg_sort_pow2, er::kBlock)), static_cast<unsigned>(g->n));
        hipLaunchKernelGGL(er::emb_route_seg_kernel, grid, dim3(er::kBlock), 0, s, g->keys_out, g->vals_out, g->head_flags,
                           g->head_index, g->d_ent_base, g->seg_count, g->n, unique_keys, entry_unique_index, n_unique);
      }
      ER_LAUNCH_CHECK();
      g->heads_epoch = g->sort_epoch;
      fused_route = true;
    } else {
      if (int rc = emb_group_build_sort(g, s)) return rc;
      if (int rc = emb_group_heads(g, n_unique, s)) return rc;
      g->heads_epoch = g->sort_epoch;
    }
  }
  if (unique_keys && !fused_route && !adopted && g->peer_cap > 0) {
    // device-wide sort + fixed-capacity layout: compact list into scratch, counts per owner, then the padded layout
    uint32_t* compact = g->keys_in;  // (the unsorted keys are no longer needed)
    int32_t* cnt = reinterpret_cast<int32_t*>(g->seg_count);
    const int blocks = static_cast<int>(er::ceil_div(N, er::kBlock));
    hipLaunchKernelGGL(er::emb_route_kernel, dim3(blocks), dim3(er::kBlock), 0, s, g->keys_out, g->vals_out, g->head_flags,
                       g->head_index, N, compact, entry_unique_index);
    hipLaunchKernelGGL(er::emb_owner_counts_kernel, dim3(1), dim3(64), 0, s, compact, n_unique, g->world, g->shard_stride,
                       cnt);
    hipLaunchKernelGGL(er::emb_route_pad_kernel, dim3(blocks), dim3(er::kBlock), 0, s, g->keys_out, g->vals_out,
                       g->head_flags, g->head_index, N, cnt, g->world, static_cast<uint32_t>(g->shard_stride),
                       static_cast<uint32_t>(g->peer_cap), static_cast<uint32_t>(g->peer_hdr), unique_keys,
                       entry_unique_index, owner_counts, g->d_overflow);
    ER_LAUNCH_CHECK();
    counts_done = true;
  } else if (unique_keys && !fused_route) {
    const er_emb_group* src = g->src;
    hipLaunchKernelGGL(er::emb_route_kernel, dim3(static_cast<int>(er::ceil_div(N, er::kBlock))), dim3(er::kBlock), 0, s,
                       src->keys_out, src->vals_out, src->head_flags, src->head_index, N, unique_keys,
                       entry_unique_index);
    ER_LAUNCH_CHECK();
  }
  if (owner_counts && !counts_done) {
    const int64_t stride = g->d_local_base ? g->shard_stride : g->total_rows;
    hipLaunchKernelGGL(er::emb_owner_counts_kernel, dim3(1), dim3(64), 0, s, unique_keys, n_unique, g->world, stride,
                       owner_counts);
    ER_LAUNCH_CHECK();
  }
  g->sorted_valid = true;
  return 0;
}

int er_emb_reduce_local_tail(er_emb_group* const* groups, const int32_t* modes, float* const* outs, const int32_t* ld, int n,
                             const er_gemm_problem* wgrads, int n_wgrads, int32_t wgrad_blocks,
                             const er_loss_tail_job* loss_tail, er_stream_t stream) {
  ER_REQUIRE(groups && modes && outs && ld && n >= 1 && n <= er::kMaxMulti, "er_emb_reduce_local_tail: bad arguments (1 <= n <= %d)",
             er::kMaxMulti);
  ER_REQUIRE(n_wgrads >= 0 && n_wgrads <= er::kMaxGroup && (n_wgrads == 0 || wgrads) && wgrad_blocks >= 0,
             "er_emb_reduce_local_tail: 0 <= n_wgrads <= %d", er::kMaxGroup);
  ER_REQUIRE(!loss_tail || n_wgrads > 0, "er_emb_reduce_local_tail: the loss tail rides with the weight gradients' grid");
  hipStream_t s = er::as_stream(stream);
  er::GroupedPlan plan;
  plan.ra.n = 0;
  if (n_wgrads > 0) {
    for (int i = 0; i < n_wgrads; ++i)
      ER_REQUIRE(!wgrads[i].col_stats && !wgrads[i].bn_partial, "er_emb_reduce_local_tail: problem %d: plain contractions only", i);
    if (int rc = er::plan_grouped(ER_GEMM_TN, wgrads, n_wgrads, false, &plan, wgrad_blocks)) return rc;
  }
  er::LossTailArgs lt;
  std::memset(&lt, 0, sizeof(lt));
  if (loss_tail)
    if (int rc = er::make_loss_tail_args(loss_tail, &lt)) return rc;
  er::RunMulti ma;
  ma.n = 0;
  ma.start[0] = 0;
  ma.opt_kind = ER_OPT_SGD;
  ma.hyper = nullptr;
  int fix_start[er::kMaxMulti + 1] = {0};
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    er_emb_group* g = groups[i];
    ER_REQUIRE(g && outs[i] && (modes[i] == 1 || modes[i] == 2), "er_emb_reduce_local_tail: group %d: null argument or mode %d", i,
               g ? modes[i] : -1);
    if (modes[i] == 1) {  // sharded: this step's er_emb_route left the routed sort (er_emb_bwd_reduce_routed)
      ER_REQUIRE(ld[i] == 0 || ld[i] >= g->dim, "er_emb_reduce_local_tail: group %d: ld < dim", i);
      ER_REQUIRE(g->sorted_valid, "er_emb_reduce_local_tail: group %d: call er_emb_route for this step first", i);
    } else {              // replicated: sort now unless a leader's sort is adopted (er_emb_bwd_reduce_dense)
      ER_REQUIRE(ld[i] > g->dim, "er_emb_reduce_local_tail: group %d: dense buffer needs ld > dim", i);
      if (!g->sorted_valid) {
        bool adopted = false;
        if (g->leader)
          if (int rc = emb_group_adopt(g, s, &adopted)) return rc;
        if (!adopted)
          if (int rc = emb_group_sort(g, s)) return rc;
      }
      g->sorted_valid = false;
    }
    const int64_t N = group_entries(g);
    if (N == 0) continue;
    if (int rc = front_sort_guard(g, "er_emb_reduce_local_tail")) return rc;
    const er_emb_group* src = g->src;
    er::RunArgs& a = ma.a[ma.n];
    a.skeys = src->keys_out; a.svals = src->vals_out; a.ent_gptr = g->ent_gptr; a.ent_scale = g->ent_scale;
    a.n = N; a.dim = g->dim; a.G = g->G; a.V = g->V; a.T = g->tile_entries;
    a.n_tiles = static_cast<int>(er::ceil_div(N, a.T));
    a.tab = tab_of(g);
    a.aux = er::DecayAux{};
    a.ro = modes[i] == 2 ? er::ReduceOut{2, nullptr, nullptr, nullptr, outs[i], ld[i]}
                         : er::ReduceOut{1, src->head_flags, src->head_index, nullptr, outs[i], ld[i]};
    a.tile_first = g->tile_first; a.tile_last = g->tile_last;
    ma.start[ma.n + 1] = ma.start[ma.n] + a.n_tiles;
    const int fb = a.n_tiles > 1 ? static_cast<int>(er::ceil_div(static_cast<int64_t>(a.n_tiles) * g->G, er::kBlock)) : 0;
    fix_start[ma.n + 1] = fix_start[ma.n] + fb;
    const size_t need = sizeof(float) * static_cast<size_t>(a.T) * g->dim + sizeof(uint32_t) * (a.T + 2);
    if (need > lds) lds = need;
    ++ma.n;
  }
  const int n_tile = ma.n > 0 ? ma.start[ma.n] : 0;
  const int n_fix = ma.n > 0 ? fix_start[ma.n] : 0;
  if (n_wgrads > 0) {
    const int n_gemm = er::grouped_grid(plan.ga);
    const size_t gemm_lds = sizeof(float) * 2 * 2 * er::kOpTile;
    hipLaunchKernelGGL(er::emb_reduce_local_wgrad_kernel, dim3(n_gemm + n_tile + (loss_tail ? 1 : 0)), dim3(er::kBlock),
                       lds > gemm_lds ? lds : gemm_lds, s, ma, plan.ga, n_gemm, n_tile, lt);
    ER_LAUNCH_CHECK();
  } else if (n_tile > 0) {
    hipLaunchKernelGGL(er::emb_bwd_tile_multi_kernel, dim3(n_tile), dim3(er::kBlock), lds, s, ma);
    ER_LAUNCH_CHECK();
  }
  for (int i = 0; i <= ma.n; ++i) ma.start[i] = fix_start[i];
  const int n_red = plan.ra.n > 0 ? plan.ra.start[plan.ra.n] : 0;
  if (n_fix + n_red > 0) {
    hipLaunchKernelGGL(er::emb_bwd_fix_reduce_kernel, dim3(n_fix + n_red), dim3(er::kBlock), 0, s, ma, plan.ra, n_fix);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

int er_emb_bwd_reduce_routed(er_emb_group* g, float* unique_grads, int32_t ld, er_stream_t stream) {
  ER_REQUIRE(g && unique_grads && (ld == 0 || ld >= g->dim), "er_emb_bwd_reduce_routed: null argument or ld < dim");
  ER_REQUIRE(g->sorted_valid, "er_emb_bwd_reduce_routed: call er_emb_route for this step first");
  hipStream_t s = er::as_stream(stream);
  // unique keys were already written by er_emb_route (and with a per-peer capacity the run index exceeds the entries)
  return emb_group_run(g, ER_OPT_SGD, nullptr, 1, nullptr, unique_grads, s, ld);
}

int er_gather_rows(const float* table, int64_t table_rows, int32_t dim, const uint32_t* keys, int64_t n,
                   int64_t key_sub, float* out, er_stream_t stream) {
  return er_gather_rows_ld(table, dim, table_rows, dim, keys, n, key_sub, out, stream);
}

int er_gather_rows_ld(const float* table, int64_t table_ld, int64_t table_rows, int32_t dim, const uint32_t* keys, int64_t n,
                      int64_t key_sub, float* out, er_stream_t stream) {
  ER_REQUIRE(table && keys && out && dim > 0 && n >= 0 && table_ld >= dim, "er_gather_rows: bad arguments");
  if (n == 0) return 0;
  int V, G;
  er::lane_geom(dim, V, G);
  const int blocks = static_cast<int>(er::ceil_div(n * G, er::kBlock));
  if (V == 4) {
    hipLaunchKernelGGL(er::gather_rows_kernel<4>, dim3(blocks), dim3(er::kBlock), 0, er::as_stream(stream), table, table_ld, keys,
                       n, dim, G, key_sub, table_rows, out);
  } else {
    hipLaunchKernelGGL(er::gather_rows_kernel<1>, dim3(blocks), dim3(er::kBlock), 0, er::as_stream(stream), table, table_ld, keys,
                       n, dim, G, key_sub, table_rows, out);
  }
  ER_LAUNCH_CHECK();
  return 0;
}

int er_emb_owner_merge(er_emb_group* g, const int32_t* run_counts, int n_runs, er_stream_t stream) {
  ER_REQUIRE(g && run_counts && n_runs >= 1 && n_runs <= er::kMaxRuns, "er_emb_owner_merge: bad arguments (1 <= runs <= %d)",
             er::kMaxRuns);
  ER_REQUIRE(g->n == 1 && !g->has_ragged && !g->d_local_base && !g->leader,
             "er_emb_owner_merge: needs a group of ONE dense-mode lookup that follows no other group");
  hipStream_t s = er::as_stream(stream);
  const int64_t N = group_entries(g);
  er::MergeRuns r;
  r.n = n_runs;
  r.off[0] = 0;
  for (int i = 0; i < n_runs; ++i) {
    ER_REQUIRE(run_counts[i] >= 0, "er_emb_owner_merge: negative run length");
    r.off[i + 1] = r.off[i] + run_counts[i];
  }
  ER_REQUIRE(r.off[n_runs] == N, "er_emb_owner_merge: the runs hold %d keys, the group's active rows are %lld",
             r.off[n_runs], (long long)N);
  if (N == 0) return 0;
  if (int rc = emb_group_build(g, s)) return rc;
  g->src = g;
  ++g->sort_epoch;
  hipLaunchKernelGGL(er::emb_owner_merge_kernel, dim3(static_cast<unsigned>(er::ceil_div(N, er::kBlock))), dim3(er::kBlock),
                     0, s, g->keys_in, g->vals_in, r, g->keys_out, g->vals_out, g->head_flags);
  ER_LAUNCH_CHECK();
  g->merged_epoch = g->sort_epoch;
  g->sorted_valid = true;
  return 0;
}

int er_emb_group_set_peer_capacity(er_emb_group* g, int64_t peer_cap, int32_t count_header) {
  ER_REQUIRE(g && peer_cap >= 0 && (count_header == 0 || count_header == 1), "er_emb_group_set_peer_capacity: bad arguments");
  if (peer_cap > 0) {
    ER_REQUIRE(g->d_local_base && g->world <= 64, "er_emb_group_set_peer_capacity: needs er_emb_group_set_routing, world <= 64");
    ER_REQUIRE(peer_cap * g->world < 0x7FFFFFFFLL, "er_emb_group_set_peer_capacity: world * capacity too large");
    if (!g->d_overflow) {
      ER_CHECK_HIP(hipMalloc(&g->d_overflow, sizeof(int32_t)));
      ER_CHECK_HIP(hipMemset(g->d_overflow, 0, sizeof(int32_t)));
    }
  }
  g->peer_cap = peer_cap;
  g->peer_hdr = peer_cap > 0 ? count_header : 0;
  return 0;
}

int er_emb_route_overflow(er_emb_group* g, int32_t* overflow_host) {
  ER_REQUIRE(g && overflow_host, "er_emb_route_overflow: null argument");
  *overflow_host = 0;
  if (g->d_overflow) ER_CHECK_HIP(hipMemcpy(overflow_host, g->d_overflow, sizeof(int32_t), hipMemcpyDeviceToHost));
  return 0;
}

int er_emb_owner_ids(const uint32_t* recv_keys, const int32_t* counts, int n_runs, int64_t peer_cap, int64_t key_sub,
                     int64_t* ids, int32_t* counts_out, er_stream_t stream) {
  ER_REQUIRE(recv_keys && ids && (counts || counts_out) && n_runs >= 1 && peer_cap > 0 &&
             n_runs * (peer_cap + 1) < 0x7FFFFFFFLL, "er_emb_owner_ids: bad arguments");
  const int64_t n = n_runs * peer_cap;
  hipLaunchKernelGGL(er::emb_owner_ids_kernel, dim3(static_cast<unsigned>(er::ceil_div(n, er::kBlock))), dim3(er::kBlock), 0,
                     er::as_stream(stream), recv_keys, counts, n_runs, static_cast<int>(peer_cap), counts ? 0 : 1, key_sub,
                     ids, counts_out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_emb_owner_ids_merge(er_emb_group* g, const uint32_t* recv_keys, const int32_t* counts, int n_runs, int64_t peer_cap,
                           int64_t key_sub, int64_t* ids, int32_t* counts_out, int build_lag1_tables, er_stream_t stream) {
  ER_REQUIRE(g && recv_keys && ids && (counts || counts_out) && n_runs >= 1 && peer_cap > 0 &&
                 n_runs * (peer_cap + 1) < 0x7FFFFFFFLL,
             "er_emb_owner_ids_merge: bad arguments");
  ER_REQUIRE(g->n == 1 && !g->has_ragged && !g->d_local_base && !g->leader && g->n_active < 0,
             "er_emb_owner_ids_merge: needs a group of ONE dense-mode lookup, all rows active, following no group");
  ER_REQUIRE(g->n_entries == n_runs * peer_cap, "er_emb_owner_ids_merge: the group holds %lld rows, not %d x %lld",
             (long long)g->n_entries, n_runs, (long long)peer_cap);
  ER_REQUIRE(g->h_descs[0].ids == ids && !g->h_descs[0].offsets && !g->h_descs[0].weights,
             "er_emb_owner_ids_merge: the group's lookup must read `ids` (dense mode, unweighted)");
  hipStream_t s = er::as_stream(stream);
  er_emb_group* gs[er::kMaxMulti];
  er::BuildMulti bm;
  if (int rc = emb_group_build_prepare(g, s, &bm, gs)) return rc;
  const int n_main = static_cast<int>(er::ceil_div(g->n_entries, er::kBlock));
  for (int i = 0; i < bm.n; ++i)
    ER_REQUIRE(gs[i]->n_build_blocks == n_main && !gs[i]->has_ragged,
               "er_emb_owner_ids_merge: group %d of the shared sort does not have the leader's entries", i);
  for (int i = 1; i < bm.n; ++i) gs[i]->built_epoch = g->sort_epoch + 1;
  g->src = g;
  ++g->sort_epoch;
  er::DecayTabDev tabs{};
  const float* hist = nullptr;
  const int64_t* counter = nullptr;
  int n_tab = 0;
  if (build_lag1_tables && g->tabs) {
    tabs = g->tabs->dev;
    hist = g->tabs->hist;
    counter = g->tabs->counter;
    n_tab = static_cast<int>(er::ceil_div(static_cast<int64_t>(tabs.K) * er::kWave, er::kBlock));
    g->tabs->lag1_built = true;  // (consumed by the er_emb_owner_serve that follows on this stream)
  }
  hipLaunchKernelGGL(er::emb_owner_ids_merge_kernel, dim3(static_cast<unsigned>(n_main + n_tab)), dim3(er::kBlock), 0, s,
                     recv_keys, counts, n_runs, static_cast<int>(peer_cap), counts ? 0 : 1, key_sub, ids, counts_out, bm, n_main,
                     g->keys_out, g->vals_out, g->head_flags, tabs, hist, counter);
  ER_LAUNCH_CHECK();
  g->merged_epoch = g->sort_epoch;
  g->sorted_valid = true;
  return 0;
}

int er_emb_owner_merge_padded(er_emb_group* g, const int32_t* counts, int n_runs, int64_t peer_cap, er_stream_t stream) {
  ER_REQUIRE(g && counts && n_runs >= 1 && peer_cap > 0, "er_emb_owner_merge_padded: bad arguments");
  ER_REQUIRE(g->n == 1 && !g->has_ragged && !g->d_local_base && !g->leader && g->n_active < 0,
             "er_emb_owner_merge_padded: needs a group of ONE dense-mode lookup, all rows active, following no group");
  ER_REQUIRE(g->n_entries == n_runs * peer_cap, "er_emb_owner_merge_padded: the group holds %lld rows, not %d x %lld",
             (long long)g->n_entries, n_runs, (long long)peer_cap);
  hipStream_t s = er::as_stream(stream);
  if (int rc = emb_group_build(g, s)) return rc;
  g->src = g;
  ++g->sort_epoch;
  hipLaunchKernelGGL(er::emb_owner_merge_padded_kernel, dim3(static_cast<unsigned>(er::ceil_div(g->n_entries, er::kBlock))),
                     dim3(er::kBlock), 0, s, g->keys_in, g->vals_in, counts, n_runs, static_cast<int>(peer_cap), g->keys_out,
                     g->vals_out, g->head_flags);
  ER_LAUNCH_CHECK();
  g->merged_epoch = g->sort_epoch;
  g->sorted_valid = true;
  return 0;
}

int er_emb_owner_serve(er_emb_group* const* groups, float* const* rows_out, const int32_t* ld, int n,
                       const er_opt_hyper* hyper, er_stream_t stream) {
  ER_REQUIRE(groups && rows_out && n >= 1 && n <= er::kMaxMulti, "er_emb_owner_serve: bad arguments (1 <= n <= %d)",
             er::kMaxMulti);
  hipStream_t s = er::as_stream(stream);
  er::ServeMulti ma;
  ma.n = 0;
  ma.start[0] = 0;
  ma.hyper = hyper;
  for (int i = 0; i < n; ++i) {
    er_emb_group* g = groups[i];
    ER_REQUIRE(g && rows_out[i], "er_emb_owner_serve: null argument (group %d)", i);
    const int64_t N = group_entries(g);
    if (N == 0) continue;
    if (g->leader && !g->sorted_valid) {  // follower of a shared sort: take the leader's merge of this step
      bool adopted = false;
      if (int rc = emb_group_adopt(g, s, &adopted)) return rc;
      ER_REQUIRE(adopted, "er_emb_owner_serve: group %d no longer sees its leader's keys", i);
      g->sorted_valid = true;
    }
    const er_emb_group* src = g->src;
    ER_REQUIRE(g->sorted_valid && src->merged_epoch == src->sort_epoch,
               "er_emb_owner_serve: call er_emb_owner_merge (on the group or its leader) for this step first");
    er::ServeArgs& a = ma.a[ma.n];
    a.skeys = src->keys_out; a.svals = src->vals_out; a.flags = src->head_flags; a.n = N;
    // hyper == NULL: serve the rows as they are (inference after er_emb_flush_decay: nothing is pending and nothing
    // may be replayed, because no row update follows that would advance last_step)
    a.tab = tab_of(g);
    if (!hyper) a.tab.last_step = nullptr;
    ER_REQUIRE(a.tab.last_step == nullptr || g->G <= er::kWave, "er_emb_owner_serve: a lazily decayed row must fit one wavefront");
    a.lr_hist = g->lr_hist; a.aux = decay_aux_for(g, 1); a.out = rows_out[i]; a.dim = g->dim; a.G = g->G; a.V = g->V;
    a.ld = ld && ld[i] ? ld[i] : g->dim;
    ER_REQUIRE(a.ld >= g->dim && (g->V == 1 || (a.ld % 4 == 0 && (reinterpret_cast<uintptr_t>(rows_out[i]) & 15) == 0)),
               "er_emb_owner_serve: group %d: ld %d < dim, or 16-byte lanes on rows that are not 16-byte aligned", i, a.ld);
    ma.start[ma.n + 1] = ma.start[ma.n] + static_cast<int>(er::ceil_div(N * g->G, er::kBlock));
    ++ma.n;
  }
  if (ma.n == 0) return 0;
  if (hyper) {  // (serve_body brings the rows to step *counter - 1)
    if (int rc = launch_decay_tables(groups, n, 1, s)) return rc;
  } else {
    for (int i = 0; i < n; ++i)
      if (groups[i] && groups[i]->tabs) groups[i]->tabs->lag1_built = false;
  }
  hipLaunchKernelGGL(er::emb_owner_serve_kernel, dim3(ma.start[ma.n]), dim3(er::kBlock), 0, s, ma);
  ER_LAUNCH_CHECK();
  return 0;
}

// er_emb_dense_apply's argument checks and launch record
static int make_dense_apply_multi(const er_dense_apply_desc* descs, int n, int opt_kind, const er_opt_hyper* hyper,
                                  er::DenseApplyMulti* out) {
  ER_REQUIRE(descs && hyper && n >= 1 && n <= er::kMaxMulti, "er_emb_dense_apply: bad arguments (1 <= n <= %d)",
             er::kMaxMulti);
  ER_REQUIRE(opt_kind >= ER_OPT_SGD && opt_kind <= ER_OPT_ADAGRAD, "er_emb_dense_apply: unknown optimizer %d", opt_kind);
  er::DenseApplyMulti& ma = *out;
  ma.n = 0;
  ma.start[0] = 0;
  ma.opt_kind = opt_kind;
  ma.hyper = hyper;
  for (int i = 0; i < n; ++i) {
    const er_dense_apply_desc& d = descs[i];
    ER_REQUIRE(d.var && d.dense && d.dim > 0 && d.ld > d.dim && d.rows >= 0, "er_emb_dense_apply: table %d: bad arguments", i);
    if (opt_kind == ER_OPT_ADAM || opt_kind == ER_OPT_LAZY_ADAM)
      ER_REQUIRE(d.m && d.v, "er_emb_dense_apply: Adam needs m and v (table %d)", i);
    if (opt_kind == ER_OPT_ADAGRAD) ER_REQUIRE(d.v, "er_emb_dense_apply: Adagrad needs the accumulator in v (table %d)", i);
    if (d.rows == 0) continue;
    er::DenseApplyArgs& a = ma.a[ma.n];
    a.var = d.var; a.m = d.m; a.v = d.v; a.dense = d.dense; a.ld = d.ld; a.dim = d.dim; a.rows = d.rows;
    a.tld = d.table_ld ? d.table_ld : d.dim;
    ER_REQUIRE(a.tld >= d.dim, "er_emb_dense_apply: table %d: table_ld < dim", i);
    er::lane_geom(d.dim, a.V, a.G);
    ma.start[ma.n + 1] = ma.start[ma.n] + static_cast<int>(er::ceil_div(d.rows * a.G, er::kBlock));
    ++ma.n;
  }
  return 0;
}

int er_emb_dense_apply(const er_dense_apply_desc* descs, int n, int opt_kind, const er_opt_hyper* hyper,
                       er_stream_t stream) {
  er::DenseApplyMulti ma;
  if (int rc = make_dense_apply_multi(descs, n, opt_kind, hyper, &ma)) return rc;
  if (ma.n == 0) return 0;
  hipLaunchKernelGGL(er::emb_dense_apply_kernel, dim3(ma.start[ma.n]), dim3(er::kBlock), 0, er::as_stream(stream), ma);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_emb_owner_update_tail(er_emb_group* const* groups, int n, int opt_kind, const er_opt_hyper* hyper,
                             const er_dense_apply_desc* descs, int n_apply, const er_dense_opt_job* dense_opt,
                             er_stream_t stream) {
  ER_REQUIRE(groups && hyper && dense_opt && n >= 1 && n <= er::kMaxMulti && n_apply >= 0 && n_apply <= er::kMaxMulti &&
                 (n_apply == 0 || descs),
             "er_emb_owner_update_tail: bad arguments (1 <= n <= %d groups, <= %d replicated tables, a dense optimizer job)",
             er::kMaxMulti, er::kMaxMulti);
  ER_REQUIRE(opt_kind >= ER_OPT_SGD && opt_kind <= ER_OPT_ADAGRAD, "er_emb_owner_update_tail: unknown optimizer %d", opt_kind);
  er::DenseApplyMulti am;
  am.n = 0;
  if (n_apply > 0)
    if (int rc = make_dense_apply_multi(descs, n_apply, opt_kind, hyper, &am)) return rc;
  er::DenseOptArgs da;
  if (int rc = er::make_dense_opt_args(dense_opt, &da)) return rc;
  return emb_groups_run_multi(groups, n, opt_kind, hyper, nullptr, nullptr, stream, &am, &da);
}

int er_scatter_unique(const uint32_t* keys, const float* grads, const int32_t* n_unique, int64_t capacity, int32_t dim,
                      float* dense, int32_t dense_stride, er_stream_t stream) {
  ER_REQUIRE(keys && grads && n_unique && dense && dim > 0 && dense_stride >= dim + 1 && capacity >= 0,
             "er_scatter_unique: bad arguments");
  if (capacity == 0) return 0;
  const int blocks = static_cast<int>(er::ceil_div(capacity * (dim + 1), er::kBlock));
  hipLaunchKernelGGL(er::scatter_unique_kernel, dim3(blocks), dim3(er::kBlock), 0, er::as_stream(stream), keys, grads,
                     n_unique, dim, dense, dense_stride);
  ER_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
