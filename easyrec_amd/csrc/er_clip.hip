// Gradient clipping by global norm: train_config.gradient_clipping_by_norm.
//
// Reference: compat/optimizers.py:365-376 (`_get_grad_norm` + `clip_ops.clip_by_global_norm(grads, clip, use_norm)`)
// and :453-481 (the norm: sqrt(2 * sum of tf.nn.l2_loss over every gradient; IndexedSlices contribute their `values`
// rows - one row per distinct id of a lookup; sharded tables' partial sums are all-reduced).  Every gradient is then
// multiplied by  clip_norm * min(1 / norm, 1 / clip_norm).
//
// Here the gradients never exist as separate tensors at that point: the dense ones sit in the flat gradient buffer
// (the kernel-L2 term l2 * w is added by the optimizer kernel), the embedding ones are the de-duplicated row sums
// of er_emb_bwd_reduce(_routed).  So: squared norms of both (one workgroup each, fixed order -> deterministic), one scalar kernel
// that turns the total into the multiplier and stores it in the er_opt_hyper records of the step (`clip_scale`),
// and the optimizer kernels (dense_opt_kernel, the embedding row updates) multiply by it.  All HBM-streaming fp32.
#include "er_common.h"

namespace er {

constexpr int kNormBlock = 1024;

// block-wide sum for kNormBlock threads (16 waves), fixed combination order; result valid in thread 0
__device__ __forceinline__ float block_sum_1024(float v, float* smem16) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) smem16[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < kNormBlock / 64; ++i) r = r + smem16[i];
  }
  return r;
}

// ONE workgroup per buffer, no library scratch: the sums are a few MB at most and clipping is an optional path; a
// single block keeps the order fixed (deterministic) and the call free of shared state (ranks simulated as threads of
// one process - the tests - would otherwise race on the reduction scratch between the partial and the final pass).
// acc[0] (+)= weight * sum of x[r * ld + c]^2 over the valid rows r, c < cols
__global__ void __launch_bounds__(kNormBlock)
gradsq_rows_kernel(const float* __restrict__ x, int64_t max_rows, int cols, int ld,
                   const int32_t* __restrict__ seg_counts, int n_seg, int64_t seg_stride, float weight,
                   float* __restrict__ acc, int accumulate) {
  __shared__ float red[kNormBlock / 64];
  float a = 0.f;
  // a thread owns a column slot and walks rows (no division per element: DIN's sequence groups have 204,800 rows): with
  // cols <= kNormBlock, kNormBlock / cols rows are in flight per pass; wider rows are walked in column chunks
  const int per_pass = cols <= kNormBlock ? kNormBlock / cols : 1;
  const int my_row = cols <= kNormBlock ? static_cast<int>(threadIdx.x) / cols : 0;
  const int my_col = cols <= kNormBlock ? static_cast<int>(threadIdx.x) % cols : static_cast<int>(threadIdx.x);
  const bool active = cols > kNormBlock || my_row < per_pass;
  if (active) {
    // (rows of one segment are consecutive: the segment cursor only moves forward)
    for (int64_t r = my_row; r < max_rows; r += per_pass) {
      if (seg_counts) {
        const int64_t sg = r / seg_stride;
        if (sg >= n_seg) break;
        if ((r - sg * seg_stride) >= seg_counts[sg]) continue;
      }
      for (int c = my_col; c < cols; c += kNormBlock) {
        const float v = x[r * ld + c];
        a = a + v * v;
      }
    }
  }
  const float s = block_sum_1024(a, red);
  if (threadIdx.x == 0) acc[0] = accumulate ? (acc[0] + weight * s) : weight * s;
}

// the dense variables' gradient as the optimizer will see it: grad_scale * grad + l2coef * w
__global__ void __launch_bounds__(kNormBlock)
gradsq_dense_kernel(const float* __restrict__ w, const float* __restrict__ grad, const float* __restrict__ l2coef,
                    int64_t n, const er_opt_hyper* __restrict__ hyper, float* __restrict__ acc, int accumulate) {
  __shared__ float red[kNormBlock / 64];
  const float gs = hyper->grad_scale;
  float a = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kNormBlock) {
    float g = grad[i] * gs;
    if (l2coef) {
      const float c = l2coef[i];
      if (c != 0.f) g = g + c * w[i];
    }
    a = a + g * g;
  }
  const float s = block_sum_1024(a, red);
  if (threadIdx.x == 0) acc[0] = accumulate ? (acc[0] + s) : s;
}

__global__ void clip_scale_kernel(const float* __restrict__ normsq, float clip_norm, er_opt_hyper* __restrict__ records,
                                  int n_records, float* __restrict__ norm_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float norm = sqrtf(normsq[0]);
  // clip_ops.clip_by_global_norm: scale = clip_norm * min(1 / norm, 1 / clip_norm).  A non-finite norm must not pass for
  // "no clipping" (0 in the record means that; inf would give it, and NaN compares false into 1.0): like TensorFlow,
  // which propagates NaN into every clipped gradient, the multiplier becomes NaN and the divergence shows at once.
  const float a = 1.0f / norm, b = 1.0f / clip_norm;
  float scale = clip_norm * (a < b ? a : b);
  if (!(norm <= 3.0e38f)) scale = __builtin_nanf("");
  for (int i = 0; i < n_records; ++i) records[i].clip_scale = scale;
  if (norm_out) norm_out[0] = norm;
}

}  // namespace er

extern "C" {

int er_gradsq_rows(const float* x, int64_t max_rows, int32_t cols, int32_t ld, const int32_t* seg_counts, int32_t n_seg,
                   int64_t seg_stride, float weight, float* acc, int accumulate, er_stream_t stream) {
  ER_REQUIRE(x && acc && max_rows >= 0 && cols > 0 && ld >= cols, "er_gradsq_rows: bad arguments");
  ER_REQUIRE(!seg_counts || (n_seg > 0 && seg_stride > 0), "er_gradsq_rows: seg_counts needs n_seg and seg_stride");
  hipLaunchKernelGGL(er::gradsq_rows_kernel, dim3(1), dim3(er::kNormBlock), 0, er::as_stream(stream), x, max_rows, cols,
                     ld, seg_counts, n_seg, seg_stride, weight, acc, accumulate);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_gradsq_dense(const float* w, const float* grad, const float* l2coef, int64_t n, const er_opt_hyper* hyper,
                    float* acc, int accumulate, er_stream_t stream) {
  ER_REQUIRE(w && grad && hyper && acc && n > 0, "er_gradsq_dense: bad arguments");
  hipLaunchKernelGGL(er::gradsq_dense_kernel, dim3(1), dim3(er::kNormBlock), 0, er::as_stream(stream), w, grad, l2coef, n,
                     hyper, acc, accumulate);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_clip_scale(const float* normsq, float clip_norm, er_opt_hyper* records, int32_t n_records, float* norm_out,
                  er_stream_t stream) {
  ER_REQUIRE(normsq && records && n_records > 0 && clip_norm > 0.f, "er_clip_scale: bad arguments");
  hipLaunchKernelGGL(er::clip_scale_kernel, dim3(1), dim3(64), 0, er::as_stream(stream), normsq, clip_norm, records,
                     n_records, norm_out);
  ER_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
