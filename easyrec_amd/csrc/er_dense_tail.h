// Device bodies of the step's scalar loss tail (er_loss_tail) and of the dense optimizer (er_dense_opt_step_l2), shared by
// their own launches (er_dense.hip) and by the step's fused tail (er_embedding.hip: er_emb_bwd_fused_tail), where the loss
// tail rides as ONE workgroup of the [weight gradients | embedding row update] grid and the optimizer - with the split-K
// reduce of the weight gradients folded into its gradient read - as the workgroups behind the cross-tile fix.
//
// Reference: the add_n over the loss dict + REGULARIZATION_LOSSES of model/easy_rec_estimator.py:166-184 (loss tail) and
// tf.train.AdamOptimizer / Adagrad / GradientDescent applied to the dense variables, builders/optimizer_builder.py:33-97
// (optimizer).  Every sum keeps the order of the stand-alone launch whatever the workgroup size it runs in: the results
// are bit-identical between the four-launch and the two-launch tail (tests/test_deepfm_gpu.py).
#pragma once
#include "er_common.h"
#include "er_gemm_core.h"

namespace er {

// one workgroup of 1024 threads (16 waves): the batch is a few thousand logits; sums combine in a fixed order
constexpr int kCeBlock = 1024;

struct LossPtrs {
  const float* src[8];
  float* dst[8];
};

// The scalar tail of the loss (reg_total_loss_kernel) for steps whose head ran as er_head_sigmoid_ce: a task loss may
// arrive as per-workgroup partial sums (loss = scale * sum of them, fixed order), and small column-sum jobs (the head's
// dW / db partials: dst[j] += sum_p partial[p][j]) ride along, so that the head costs no launch of its own for them.
constexpr int kTailJobs = 4;
struct TailJob { const float* partial; float* dst; int n_parts, n_cols, ld; };
struct LossTailArgs {
  const float* emb_partials; int n_partials; float emb_scale;
  const float* dense_partials; int n_dense;
  LossPtrs lp; int n_losses;
  int loss_parts[8];       // > 0: lp.src[i] holds that many partial sums ...
  float loss_scale[8];     // ... and the loss is loss_scale[i] * their sum / loss_div[i] (sigmoid_ce_kernel's expression),
  float loss_div[8];       //     also written back to loss_value[i]
  float* loss_value[8];
  float* reg_out; float* total_out;
  int n_jobs; TailJob jobs[kTailJobs];
};
constexpr int kLossTailLdsFloats = 16 * 64 + 16 + 8;  // s_job | red | s_loss

// The body is written for the 1024 LANES of the stand-alone launch; a workgroup of NT threads (1024, or 256 inside the
// fused tail) runs kCeBlock / NT of those lanes per thread - lane v = tid + NT * k keeps its column (v & 63 == tid & 63,
// NT is a multiple of 64) and its wave v >> 6 = (tid >> 6) + (NT / 64) * k - so every partial sum has the operands and the
// order of the 1024-thread launch.
template <int NT>
__device__ __forceinline__ float vblock_sum(const float (&v)[kCeBlock / NT], float* red16) {
  constexpr int VL = kCeBlock / NT;
#pragma unroll
  for (int k = 0; k < VL; ++k) {
    const float w = wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0) red16[(threadIdx.x >> 6) + (NT / 64) * k] = w;
  }
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < kCeBlock / 64; ++i) r = r + red16[i];
  }
  __syncthreads();
  return r;
}

template <int NT>
__device__ __forceinline__ void loss_tail_body(const LossTailArgs& a, float* lds /* kLossTailLdsFloats */) {
  constexpr int VL = kCeBlock / NT;
  float (*s_job)[64] = reinterpret_cast<float (*)[64]>(lds);
  float* red = lds + 16 * 64;
  float* s_loss = red + 16;
  const int tid = threadIdx.x;
  // column-sum jobs: 16 part lanes x 64 columns per round (every partial's load is in flight at once: a thread per column
  // walking its partials one after the other was a 20 us chain of dependent L2 round trips), lanes combined in a fixed order
  for (int j = 0; j < a.n_jobs; ++j) {
    const TailJob& jb = a.jobs[j];
    const int cl = tid & 63;
    for (int c0 = 0; c0 < jb.n_cols; c0 += 64) {
      const int col = c0 + cl;
#pragma unroll
      for (int k = 0; k < VL; ++k) {
        const int pl = (tid >> 6) + (NT / 64) * k;
        float s = 0.f;
        if (col < jb.n_cols) {
          float v[8];
          for (int p0 = pl; p0 < jb.n_parts; p0 += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int p = p0 + u * 16;
              v[u] = p < jb.n_parts ? jb.partial[static_cast<int64_t>(p) * jb.ld + col] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s = s + v[u];
          }
        }
        s_job[pl][cl] = s;
      }
      __syncthreads();
      if (tid < 64 && col < jb.n_cols) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t = t + s_job[q][cl];
        jb.dst[col] = jb.dst[col] + t;
      }
      __syncthreads();
    }
  }
  for (int i = 0; i < a.n_losses; ++i) {
    if (a.loss_parts[i] <= 0) continue;  // (uniform)
    float v[VL];
#pragma unroll
    for (int k = 0; k < VL; ++k) {
      v[k] = 0.f;
      for (int p = tid + NT * k; p < a.loss_parts[i]; p += kCeBlock) v[k] = v[k] + a.lp.src[i][p];
    }
    const float tot = vblock_sum<NT>(v, red);
    if (tid == 0) {
      const float l = a.loss_scale[i] * tot / a.loss_div[i];
      s_loss[i] = l;
      if (a.loss_value[i]) a.loss_value[i][0] = l;
    }
  }
  float e[VL], b[VL];
#pragma unroll
  for (int k = 0; k < VL; ++k) {
    e[k] = 0.f;
    for (int i = tid + NT * k; i < a.n_partials; i += kCeBlock) e[k] = e[k] + a.emb_partials[i];
  }
  const float emb = vblock_sum<NT>(e, red);
#pragma unroll
  for (int k = 0; k < VL; ++k) {
    b[k] = 0.f;
    for (int i = tid + NT * k; i < a.n_dense; i += kCeBlock) b[k] = b[k] + a.dense_partials[i];
  }
  const float dense = vblock_sum<NT>(b, red);
  if (tid != 0) return;
  const float reg = a.emb_scale * emb + dense;
  a.reg_out[0] = reg;
  float total = reg;
  for (int i = 0; i < a.n_losses; ++i) {
    const float v = a.loss_parts[i] > 0 ? s_loss[i] : a.lp.src[i][0];
    if (a.lp.dst[i]) a.lp.dst[i][0] = v;
    total = total + v;
  }
  a.total_out[0] = total;
}

// ------------------------------------------------------------------------------------------------
// one element of the dense optimizer on its raw gradient; returns 0.5 * coef * w_new^2 (its share of the next step's
// kernel-L2 loss)
__device__ __forceinline__ float dense_opt_elem_g(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                  float g_raw, const float* __restrict__ l2coef, int64_t i, int opt_kind,
                                                  const er_opt_hyper& h) {
  float wi = w[i];
  float g = g_raw * h.grad_scale;
  if (l2coef) {
    const float c = l2coef[i];
    if (c != 0.f) g = g + c * wi;
  }
  if (h.clip_scale != 0.f) g = g * h.clip_scale;  // clip_by_global_norm (er_clip_scale); 0 = no clipping
  if (opt_kind == ER_OPT_ADAM || opt_kind == ER_OPT_LAZY_ADAM) {
    // training_ops.apply_adam (dense): m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= m*alpha/(sqrt(v)+eps)
    float mi = m[i], vi = v[i];
    mi = mi + (g - mi) * h.one_minus_beta1;
    vi = vi + (g * g - vi) * h.one_minus_beta2;
    wi = wi - (mi * h.lr_t) / (sqrtf(vi) + h.eps);
    m[i] = mi;
    v[i] = vi;
  } else if (opt_kind == ER_OPT_ADAGRAD) {
    float vi = v[i] + g * g;
    v[i] = vi;
    wi = wi - (g * h.lr) / sqrtf(vi);
  } else {
    wi = wi - h.lr * g;
  }
  w[i] = wi;
  float l2 = 0.f;
  if (l2coef) {
    const float c = l2coef[i];
    if (c != 0.f) l2 = c * (0.5f * (wi * wi));
  }
  return l2;
}
__device__ __forceinline__ float dense_opt_elem(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                const float* __restrict__ grad, const float* __restrict__ l2coef,
                                                int64_t i, int opt_kind, const er_opt_hyper& h) {
  return dense_opt_elem_g(w, m, v, grad[i], l2coef, i, opt_kind, h);
}

struct DenseOptArgs {
  float* w; float* m; float* v;
  float* grad;             // the flat gradient buffer (the k-split weight gradients are finished INTO it: er_emb_bwd_fused_tail)
  const float* l2coef;
  int64_t n;
  int opt_kind;
  const er_opt_hyper* hyper;
  float* l2_partial;       // [ceil(n / 256)] or nullptr
};

// Workgroup b of the dense optimizer (dense_opt_kernel's 256 elements, its l2_partial[b]).  ra.n > 0: elements that a
// k-split weight gradient of the step's grouped launch covers (ra.r[p].C .. + mn inside `grad`, ldc == N) are first
// finished from the split-K workspace - splitk_reduce_elems' sum, split by split in order, + bias, (+)= - stored to grad[]
// and used at once: the reduce costs no pass of its own and the gradient no second read.
__device__ __forceinline__ void dense_opt_block(const DenseOptArgs& a, const GroupedReduceArgs& ra, int b, float* red4) {
  const int64_t i = static_cast<int64_t>(b) * kBlock + threadIdx.x;
  float l2 = 0.f;
  if (i < a.n) {
    float g = a.grad[i];
    for (int p = 0; p < ra.n; ++p) {  // (<= 16 ranges; a workgroup's 256 elements lie in one or two of them)
      const ReduceItem& r = ra.r[p];
      const int64_t j = i - (r.C - a.grad);
      if (j < 0 || j >= r.mn) continue;
      float s = 0.f;
#pragma unroll 4
      for (int z = 0; z < r.splits; ++z) s = s + r.ws[z * r.mn + j];
      if (r.bias) s = s + r.bias[static_cast<int>(j % r.N)];
      g = r.accumulate ? g + s : s;
      a.grad[i] = g;
      break;
    }
    l2 = dense_opt_elem_g(a.w, a.m, a.v, g, a.l2coef, i, a.opt_kind, *a.hyper);
  }
  if (a.l2_partial) {  // (uniform) sum over the block of 0.5 * coef * w_new^2: the next step's kernel-L2 loss term
    const float sum = block_sum_256(l2, red4);
    if (threadIdx.x == 0) a.l2_partial[b] = sum;
  }
}

// host side (er_dense.hip): er_loss_tail's / er_dense_opt_step_l2's argument checks and records without the launch
int make_loss_tail_args(const er_loss_tail_job* job, LossTailArgs* out);
int make_dense_opt_args(const er_dense_opt_job* job, DenseOptArgs* out);

}  // namespace er
