// Closed-form replay of TF-Adam's decay-only steps (er_decay_tables_*; used by er_embedding.hip, maintained by the
// step prologue in er_dense.hip).
//
// A row that no lookup touches for k steps after step t0 receives, under tf.train.AdamOptimizer's sparse apply
// (builders/optimizer_builder.py:61-66 -> every row decays every step), with a = sqrt(v0), q = sqrt(beta2):
//     m_s = m0 beta1^s,   v_s = v0 beta2^s,   var -= sum_{s=1..k} lr_t(t0+s) m0 beta1^s / (a q^s + eps).
// With d = a + eps, z = a / d in [0, 1) and w_s = 1 - q^s:   a q^s + eps = d (1 - z w_s), so
//     1 / (a q^s + eps) = (1 / d) sum_n (z w_s)^n          (|z w_s| <= w_s: converges for EVERY a, eps)
//     update = (m0 / d) sum_n z^n T_n(t0, k),      T_n(t0, k) = sum_{s=1..k} lr_t(t0+s) beta1^s w_s^n.
// T_n depends on the row only through (t0, k): the per-element work is one sqrt, two divisions and a degree-5 Horner
// evaluation instead of k steps of sqrt + division.  beta1^s kills the series long before w_s grows: terms beyond
// s = K (beta1^K <= 2e-9) are dropped, and kDecayN = 6 powers of w leave a remainder below 1e-8 of the update for
// beta2 >= 0.99 (er_decay_tables_supported checks both; other betas keep the exact replay).
//   A[k-1][n] = T_n(s_end-1-k, k), k = 1..K : rebuilt from the lr_t history by decay_tables_kernel in front of every
//               consumer launch (all rows of a launch are brought to the same step s_end);
//   C[t0+1][n] = T_n(t0, K)                 : appended by the step prologue at step t0 + K, when its last term exists;
//               rows idle for more than K steps read it (their later terms are below the cut).
// Measured against an fp64 evaluation of the step-by-step recurrence the closed form is CLOSER than the fp32
// step-by-step replay (2e-7 against 1e-6 of the update, tools/decay_closed_form.py).
#pragma once
#include "er_common.h"

namespace er {

constexpr int kDecayKMax = 512;
constexpr int kDecayN = 6;   // powers of w kept
constexpr int kDecayLd = 8;  // floats per table entry (two 16-byte loads)

struct DecayTabDev {
  const double* coef;  // [K][kDecayLd]: beta1^s (1 - sqrt(beta2)^s)^n at row s-1
  float* A;            // [K][kDecayLd]
  float* C;            // [capacity + 1][kDecayLd]
  int K;
  int64_t capacity;    // of the lr_t history
  // The lag-1 table built one launch EARLY (er_decay_tables_set_prologue_build): lag[0] = the step counter as the last
  // lookup launch saw it (= the s_end of the NEXT step's lag-1 consumers; er_emb_fwd_lazy / er_decay_tables_sync write it -
  // a word no launch of the prologue writes, so the prologue's table workgroups can read it while workgroup 0 increments
  // the counter itself), lag[1] = the s_end the lag-1 table was last built for (its consumers check it), lag[2] = sticky
  // error: a consumer found the table built for another step (er_decay_tables_error).
  int64_t* lag;
};

// T_n(t0, k), n < kDecayN, by one wavefront (fixed combination order); lane n of the result holds T_n.
__device__ __forceinline__ float decay_sum_wave(const DecayTabDev& t, const float* __restrict__ hist, int64_t t0, int k) {
  const int lane = threadIdx.x & 63;
  double acc[kDecayN];
#pragma unroll
  for (int n = 0; n < kDecayN; ++n) acc[n] = 0.0;
  for (int s = lane + 1; s <= k; s += 64) {
    const int64_t idx = t0 + s;
    if (idx < 0) continue;
    const double lr = static_cast<double>(hist[idx]);
    const double* c = t.coef + static_cast<int64_t>(s - 1) * kDecayLd;
#pragma unroll
    for (int n = 0; n < kDecayN; ++n) acc[n] = acc[n] + lr * c[n];
  }
  float mine = 0.f;
#pragma unroll
  for (int n = 0; n < kDecayN; ++n) {
    double x = acc[n];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = x + __shfl_xor(x, off, 64);
    if (lane == n) mine = static_cast<float>(x);
  }
  return mine;
}

// the workgroups of a launch that build A (lag 1) for s_end: wave `w` (0-based, over all of them) handles k = w + 1
__device__ __forceinline__ void decay_build_lag1(const DecayTabDev& t, const float* __restrict__ hist, int64_t s_end, int w) {
  const int k = w + 1;
  if (k > t.K) return;
  const float mine = decay_sum_wave(t, hist, s_end - 1 - k, k);
  const int lane = threadIdx.x & 63;
  float* A = t.A + static_cast<int64_t>(kDecayKMax) * kDecayLd;  // (lag 1's table: decay_aux_for)
  if (lane < kDecayLd) A[static_cast<int64_t>(k - 1) * kDecayLd + lane] = mine;  // (lanes >= kDecayN hold 0)
  if (k == 1 && lane == 0) t.lag[1] = s_end;
}

}  // namespace er

struct er_decay_tables {
  er::DecayTabDev dev;
  const float* hist = nullptr;
  const int64_t* counter = nullptr;
  double ln_b1 = 0.0, ln_b2 = 0.0;
  float beta1 = 0.f, beta2 = 0.f;
  bool prologue_build = false;  // er_decay_tables_set_prologue_build
  // the lag-1 table of THIS step was already built by a rider of an earlier launch on the same stream
  // (er_emb_owner_ids_merge): the next er_emb_owner_serve consumes the note instead of launching decay_tables_kernel
  bool lag1_built = false;
};
