// K5-K8, K11: feature-interaction kernels (FM, wide sum, DCN cross v1 / v2 epilogue, DIN target
// attention, MMoE mixing).  All are HBM/L2-bound elementwise + small-reduction work on [B, d]
// activations that the embedding forward has just written; they are kept off the MFMA path on
// purpose and fuse what the reference issues as 4-10 separate TF ops each.
//
// Reference call sites: layers/fm.py:20-26, model/deepfm.py:62-63, model/dcn.py:32-45,
// layers/keras/interaction.py:276-286, model/multi_tower_din.py:62-97, layers/mmoe.py:73-82.
#include "er_common.h"
#include "er_fm_bodies.h"
#include "er_grad_finish.h"

namespace er {

// ------------------------------------------------------------------------------------------------
// FM
// ------------------------------------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(kBlock)
fm_fwd_kernel(const float* __restrict__ x, int B, int F, int D, int x_stride, float* __restrict__ fm_out,
              float* __restrict__ sum_out) {
  fm_fwd_body<V>(static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x, x, B, F, D, x_stride, fm_out, D, sum_out);
}

template <int V>
__global__ void __launch_bounds__(kBlock)
fm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ S, const float* __restrict__ g, int B, int F,
              int D, int x_stride, float* __restrict__ dx, int dx_stride, int accumulate) {
  // one lane per (b, f, V columns): fully coalesced over the [B, F*D] activation
  const int lanes_row = (F * D) / V;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t b = idx / lanes_row;
  if (b >= B) return;
  const int col = static_cast<int>(idx % lanes_row) * V;  // column in [0, F*D)
  const int c = col % D;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const float e = x[b * x_stride + col + i];
    const float v = g[b * D + c + i] * (S[b * D + c + i] - e);
    float* o = dx + b * dx_stride + col + i;
    *o = accumulate ? (*o + v) : v;
  }
}

__global__ void __launch_bounds__(kBlock)
rowsum_fwd_kernel(const float* __restrict__ x, int B, int n, int x_stride, float* __restrict__ out) {
  rowsum_fwd_body(static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x, x, B, n, x_stride, out, 1);
}

// DeepFM's final-DNN input [sum(wide) | FM(fields) | deep] (reference model/deepfm.py:60-83: reduce_sum, layers/fm.py, concat)
// in ONE launch: three workgroup ranges running the bodies of fm_fwd_kernel, rowsum_fwd_kernel and a copy - the same
// arithmetic in the same order as the three separate launches.
template <int V>
__global__ void __launch_bounds__(kBlock)
wide_fm_concat_kernel(const float* __restrict__ wide, int n_w, int ld_w, const float* __restrict__ fm_x, int F, int D, int ld_x,
                      const float* __restrict__ deep, int n_d, int ld_d, int B, float* __restrict__ out, int ld_out,
                      float* __restrict__ sum_out, int fm_blocks, int rs_blocks) {
  const int bid = blockIdx.x;
  if (bid < fm_blocks) {
    fm_fwd_body<V>(static_cast<int64_t>(bid) * kBlock + threadIdx.x, fm_x, B, F, D, ld_x, out + 1, ld_out, sum_out);
  } else if (bid < fm_blocks + rs_blocks) {
    rowsum_fwd_body(static_cast<int64_t>(bid - fm_blocks) * kBlock + threadIdx.x, wide, B, n_w, ld_w, out, ld_out);
  } else {
    const int64_t i = static_cast<int64_t>(bid - fm_blocks - rs_blocks) * kBlock + threadIdx.x;
    if (i >= static_cast<int64_t>(B) * n_d) return;
    const int64_t b = i / n_d;
    const int c = static_cast<int>(i % n_d);
    out[b * ld_out + 1 + D + c] = deep[b * ld_d + c];
  }
}

__global__ void __launch_bounds__(kBlock)
rowsum_bwd_kernel(const float* __restrict__ g, int B, int n, float* __restrict__ dx, int dx_stride, int accumulate) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t b = idx / n;
  if (b >= B) return;
  const int j = static_cast<int>(idx % n);
  float* o = dx + b * dx_stride + j;
  *o = accumulate ? (*o + g[b]) : g[b];
}

__global__ void __launch_bounds__(kBlock)
axpy2d_kernel(const float* __restrict__ x, int x_stride, float alpha, float* __restrict__ y, int y_stride, int rows,
              int cols, int accumulate) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t r = idx / cols;
  if (r >= rows) return;
  const int c = static_cast<int>(idx % cols);
  const float v = alpha * x[r * x_stride + c];
  float* o = y + r * y_stride + c;
  *o = accumulate ? (*o + v) : v;
}

// ------------------------------------------------------------------------------------------------
// DCN-v1 cross: one wave per row, the row lives in registers (d <= 64*kCrossE)
// ------------------------------------------------------------------------------------------------
constexpr int kCrossE = 16;

__global__ void __launch_bounds__(kBlock)
cross_v1_fwd_kernel(const float* __restrict__ x0g, const float* __restrict__ w, const float* __restrict__ bias,
                    int B, int d, int L, float* __restrict__ out, float* __restrict__ xl_dots) {
  const int lane = threadIdx.x & 63;
  const int64_t r = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (r >= B) return;
  float x0[kCrossE], x[kCrossE];
#pragma unroll
  for (int i = 0; i < kCrossE; ++i) {
    const int c = lane + i * 64;
    x0[i] = (c < d) ? x0g[r * d + c] : 0.f;
    x[i] = x0[i];
  }
  for (int l = 0; l < L; ++l) {
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < kCrossE; ++i) {
      const int c = lane + i * 64;
      if (c < d) part = part + x[i] * w[l * d + c];
    }
    const float dot = wave_sum(part);
    if (lane == 0) xl_dots[r * L + l] = dot;
#pragma unroll
    for (int i = 0; i < kCrossE; ++i) {
      const int c = lane + i * 64;
      if (c < d) x[i] = (x0[i] * dot + bias[l * d + c]) + x[i];
    }
  }
#pragma unroll
  for (int i = 0; i < kCrossE; ++i) {
    const int c = lane + i * 64;
    if (c < d) out[r * d + c] = x[i];
  }
}

// backward: one wave per block, rows strided over blocks; dw/db accumulated in LDS per block and
// written as partials [gridDim.x, L, d] (reduced by er_colsum) -> deterministic.
__global__ void __launch_bounds__(64)
cross_v1_bwd_kernel(const float* __restrict__ x0g, const float* __restrict__ w, const float* __restrict__ bias,
                    const float* __restrict__ xl_dots, const float* __restrict__ dout, int B, int d, int L,
                    float* __restrict__ dx0g, float* __restrict__ dw_part, float* __restrict__ db_part) {
  extern __shared__ float lds[];  // [2, L, d]
  float* dw = lds;
  float* db = lds + static_cast<size_t>(L) * d;
  const int lane = threadIdx.x;
  for (int i = lane; i < 2 * L * d; i += 64) lds[i] = 0.f;
  __syncthreads();
  for (int64_t r = blockIdx.x; r < B; r += gridDim.x) {
    float x0[kCrossE], g[kCrossE], dx0[kCrossE];
#pragma unroll
    for (int i = 0; i < kCrossE; ++i) {
      const int c = lane + i * 64;
      x0[i] = (c < d) ? x0g[r * d + c] : 0.f;
      g[i] = (c < d) ? dout[r * d + c] : 0.f;
      dx0[i] = 0.f;
    }
    for (int l = L - 1; l >= 0; --l) {
      // recompute x_l from x0 with the saved dots (x_{k+1} = x0*dot_k + b_k + x_k)
      float xl[kCrossE];
#pragma unroll
      for (int i = 0; i < kCrossE; ++i) xl[i] = x0[i];
      for (int k = 0; k < l; ++k) {
        const float dk = xl_dots[r * L + k];
#pragma unroll
        for (int i = 0; i < kCrossE; ++i) {
          const int c = lane + i * 64;
          if (c < d) xl[i] = (x0[i] * dk + bias[k * d + c]) + xl[i];
        }
      }
      const float dot = xl_dots[r * L + l];
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < kCrossE; ++i) part = part + g[i] * x0[i];
      const float ddot = wave_sum(part);
#pragma unroll
      for (int i = 0; i < kCrossE; ++i) {
        const int c = lane + i * 64;
        if (c < d) {
          dx0[i] = dx0[i] + g[i] * dot;
          db[l * d + c] = db[l * d + c] + g[i];
          dw[l * d + c] = dw[l * d + c] + ddot * xl[i];
          g[i] = g[i] + ddot * w[l * d + c];  // gradient w.r.t. x_l
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kCrossE; ++i) {
      const int c = lane + i * 64;
      if (c < d) dx0g[r * d + c] = dx0[i] + g[i];  // x_0 is also the first x_l
    }
  }
  __syncthreads();
  const size_t base = static_cast<size_t>(blockIdx.x) * L * d;
  for (int i = lane; i < L * d; i += 64) {
    dw_part[base + i] = dw[i];
    db_part[base + i] = db[i];
  }
}

// ------------------------------------------------------------------------------------------------
// DCN-v2 cross epilogue
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
cross_v2_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ x, const float* __restrict__ u,
                    const float* __restrict__ bias, float diag, int64_t n, int d, float* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int c = static_cast<int>(i % d);
  float t = u[i] + (bias ? bias[c] : 0.f);
  if (diag != 0.f) t = t + diag * x[i];
  out[i] = x0[i] * t + x[i];
}

__global__ void __launch_bounds__(kBlock)
cross_v2_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ x, const float* __restrict__ u,
                    const float* __restrict__ bias, float diag, const float* __restrict__ dout, int64_t n, int d,
                    float* __restrict__ dx0, int acc_dx0, float* __restrict__ dx, float* __restrict__ du) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int c = static_cast<int>(i % d);
  float t = u[i] + (bias ? bias[c] : 0.f);
  if (diag != 0.f) t = t + diag * x[i];
  const float g = dout[i];
  const float a = g * t;
  dx0[i] = acc_dx0 ? (dx0[i] + a) : a;
  du[i] = g * x0[i];
  dx[i] = g + (diag != 0.f ? g * x0[i] * diag : 0.f);
}

// ------------------------------------------------------------------------------------------------
// DIN
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
din_concat_fwd_kernel(const float* __restrict__ q, const float* __restrict__ h, int B, int L, int E,
                      float* __restrict__ out) {
  const int64_t n = static_cast<int64_t>(B) * L * E;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int e = static_cast<int>(i % E);
  const int64_t bt = i / E;
  const int64_t b = bt / L;
  const float qv = q[b * E + e];
  const float hv = h[i];
  float* o = out + bt * (4 * E);
  o[e] = qv;
  o[E + e] = hv;
  o[2 * E + e] = qv - hv;
  o[3 * E + e] = qv * hv;
}

// dhist: elementwise.  dquery: sum over t -> one lane per (b, e) loops over L (coalesced over e).
__global__ void __launch_bounds__(kBlock)
din_concat_bwd_hist_kernel(const float* __restrict__ q, const float* __restrict__ dout, int B, int L, int E,
                           float* __restrict__ dh, int acc_h) {
  const int64_t n = static_cast<int64_t>(B) * L * E;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int e = static_cast<int>(i % E);
  const int64_t bt = i / E;
  const int64_t b = bt / L;
  const float* g = dout + bt * (4 * E);
  const float v = (g[E + e] - g[2 * E + e]) + g[3 * E + e] * q[b * E + e];
  dh[i] = acc_h ? (dh[i] + v) : v;
}

__global__ void __launch_bounds__(kBlock)
din_concat_bwd_query_kernel(const float* __restrict__ h, const float* __restrict__ dout, int B, int L, int E,
                            float* __restrict__ dq, int acc_q) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * E) return;
  const int e = static_cast<int>(i % E);
  const int64_t b = i / E;
  float s = 0.f;
  for (int t = 0; t < L; ++t) {
    const int64_t bt = b * L + t;
    const float* g = dout + bt * (4 * E);
    s = s + ((g[e] + g[2 * E + e]) + g[3 * E + e] * h[bt * E + e]);
  }
  dq[i] = acc_q ? (dq[i] + s) : s;
}

// ------------------------------------------------------------------------------------------------
// DIN, one wave per example with 16-byte lanes: lane = (row r = lane / C, chunk c = lane % C), C = E / 4 chunks per
// history row, 64 / C rows per pass; a [L x E] history block is read ONCE, coalesced, and sums over t finish with
// xor-shuffles across the row lanes.  Needs L <= 64 (a lane holds position `lane`'s probability), E % 4 == 0 and
// E / 4 a power of two <= 64, 16-byte aligned tensors: DIN's L = 50, E = 32.  Anything else: the kernels below (one
// lane per position / per output column, 4-byte lanes).  Round 3: the four kernels were 1/11 of the DIN step.
// ------------------------------------------------------------------------------------------------
typedef float f32x4i __attribute__((ext_vector_type(4)));

__host__ __device__ inline bool din_fast_ok(int L, int E, const void* a, const void* b = nullptr, const void* c = nullptr) {
  const int C = E / 4;
  return L <= 64 && E % 4 == 0 && C >= 1 && C <= 64 && (C & (C - 1)) == 0 &&
         ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

// sum over the lanes that share `lane % C` (C a power of two): xor-shuffles over the row bits
__device__ __forceinline__ float row_lanes_sum(float v, int C) {
  for (int m = C; m < 64; m <<= 1) v = v + __shfl_xor(v, m, 64);
  return v;
}

__global__ void __launch_bounds__(kBlock)
din_pool_fwd_fast_kernel(const float* __restrict__ scores, const float* __restrict__ hist, const int32_t* __restrict__ len,
                         int B, int L, int E, float scale, float* __restrict__ probs, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (b >= B) return;
  const float kPad = -4294967295.0f;  // -2**32 + 1 (reference model/multi_tower_din.py:88)
  const int n = len[b];
  const float s = (lane < L) ? ((lane < n) ? scores[b * L + lane] * scale : kPad) : -INFINITY;
  const float mx = wave_max(s);
  const float ex = (lane < L) ? expf(s - mx) : 0.f;
  const float den = wave_sum(ex);
  const float p = ex / den;
  if (lane < L) probs[b * L + lane] = p;
  const int C = E / 4, c = lane % C, r0 = lane / C, R = 64 / C;
  f32x4i acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4i* h4 = reinterpret_cast<const f32x4i*>(hist + b * L * E);
  for (int t0 = 0; t0 < L; t0 += R) {
    const int t = t0 + r0;
    const float pt = __shfl(p, t < L ? t : 0, 64);
    if (t < L) {
      const f32x4i v = h4[t * C + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = acc[j] + pt * v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = row_lanes_sum(acc[j], C);
  if (r0 == 0) *reinterpret_cast<f32x4i*>(out + b * E + c * 4) = acc;
}

__global__ void __launch_bounds__(kBlock)
din_pool_bwd_fast_kernel(const float* __restrict__ probs, const float* __restrict__ hist, const int32_t* __restrict__ len,
                         const float* __restrict__ dout, int B, int L, int E, float scale, float* __restrict__ dscores,
                         float* __restrict__ dhist, int acc_h) {
  const int lane = threadIdx.x & 63;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int n = len[b];
  const int C = E / 4, c = lane % C, r0 = lane / C, R = 64 / C;
  const float p = (lane < L) ? probs[b * L + lane] : 0.f;
  const f32x4i g = *reinterpret_cast<const f32x4i*>(dout + b * E + c * 4);
  const f32x4i* h4 = reinterpret_cast<const f32x4i*>(hist + b * L * E);
  f32x4i* dh4 = reinterpret_cast<f32x4i*>(dhist + b * L * E);
  // dp[t] = sum_e dout[e] * hist[t, e] lands in lane t (the probabilities' layout); dhist[t, e] = p[t] * dout[e]
  float dp = 0.f;
  for (int t0 = 0; t0 < L; t0 += R) {
    const int t = t0 + r0;
    const float pt = __shfl(p, t < L ? t : 0, 64);
    float part = 0.f;
    if (t < L) {
      const f32x4i v = h4[t * C + c];
      part = (g[0] * v[0] + g[1] * v[1]) + (g[2] * v[2] + g[3] * v[3]);
      f32x4i o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pt * g[j];
      if (acc_h) {
        const f32x4i old = dh4[t * C + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = old[j] + o[j];
      }
      dh4[t * C + c] = o;
    }
    for (int m = 1; m < C; m <<= 1) part = part + __shfl_xor(part, m, 64);  // over the row's chunks
    // row t0 + r of this pass sits in lanes r * C ..: hand it to lane t0 + r
    const int src_row = lane - t0;
    const float got = __shfl(part, (src_row >= 0 && src_row < R) ? src_row * C : 0, 64);
    if (src_row >= 0 && src_row < R && lane < L) dp = got;
  }
  const float dot_pd = wave_sum(p * dp);
  // ds = p * (dp - sum_t p * dp); masked positions get no gradient
  if (lane < L) dscores[b * L + lane] = (lane < n) ? p * (dp - dot_pd) * scale : 0.f;
}

// dhist and dquery of the [q, h, q - h, q * h] concat in one pass over dout (one wave per example)
__global__ void __launch_bounds__(kBlock)
din_concat_bwd_fast_kernel(const float* __restrict__ q, const float* __restrict__ h, const float* __restrict__ dout, int B,
                           int L, int E, float* __restrict__ dq, int acc_q, float* __restrict__ dh, int acc_h) {
  const int lane = threadIdx.x & 63;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int C = E / 4, c = lane % C, r0 = lane / C, R = 64 / C;
  const f32x4i qv = *reinterpret_cast<const f32x4i*>(q + b * E + c * 4);
  const f32x4i* h4 = reinterpret_cast<const f32x4i*>(h + b * L * E);
  const f32x4i* g4 = reinterpret_cast<const f32x4i*>(dout + b * L * 4 * E);
  f32x4i* dh4 = dh ? reinterpret_cast<f32x4i*>(dh + b * L * E) : nullptr;
  f32x4i s = {0.f, 0.f, 0.f, 0.f};
  for (int t0 = 0; t0 < L; t0 += R) {
    const int t = t0 + r0;
    if (t < L) {
      const f32x4i g0 = g4[(t * 4 + 0) * C + c], g1 = g4[(t * 4 + 1) * C + c], g2 = g4[(t * 4 + 2) * C + c],
                   g3 = g4[(t * 4 + 3) * C + c];
      const f32x4i hv = h4[t * C + c];
      if (dh4) {
        f32x4i o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (g1[j] - g2[j]) + g3[j] * qv[j];
        if (acc_h) {
          const f32x4i old = dh4[t * C + c];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = old[j] + o[j];
        }
        dh4[t * C + c] = o;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] = s[j] + ((g0[j] + g2[j]) + g3[j] * hv[j]);
    }
  }
  if (dq) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = row_lanes_sum(s[j], C);
    if (r0 == 0) {
      f32x4i* o = reinterpret_cast<f32x4i*>(dq + b * E + c * 4);
      if (acc_q) {
        const f32x4i old = *o;
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = old[j] + s[j];
      }
      *o = s;
    }
  }
}

// one wave per example: masked softmax over L (any L, strided over lanes) then p @ hist
__global__ void __launch_bounds__(kBlock)
din_pool_fwd_kernel(const float* __restrict__ scores, const float* __restrict__ hist, const int32_t* __restrict__ len,
                    int B, int L, int E, float scale, float* __restrict__ probs, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (b >= B) return;
  const float kPad = -4294967295.0f;  // -2**32 + 1 (reference model/multi_tower_din.py:88)
  const int n = len[b];
  float mx = -INFINITY;
  for (int t = lane; t < L; t += 64) {
    const float s = (t < n) ? scores[b * L + t] * scale : kPad;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float den = 0.f;
  for (int t = lane; t < L; t += 64) {
    const float s = (t < n) ? scores[b * L + t] * scale : kPad;
    den = den + expf(s - mx);
  }
  den = wave_sum(den);
  for (int t = lane; t < L; t += 64) {
    const float s = (t < n) ? scores[b * L + t] * scale : kPad;
    probs[b * L + t] = expf(s - mx) / den;
  }
  // out[b, e] = sum_t p[t] * hist[b, t, e]; lanes over e.  p[t] is recomputed (wave-uniform
  // score load) instead of re-reading probs[] written by other lanes.
  for (int e = lane; e < E; e += 64) {
    float acc = 0.f;
    for (int t = 0; t < L; ++t) {
      const float s = (t < n) ? scores[b * L + t] * scale : kPad;
      acc = acc + (expf(s - mx) / den) * hist[(b * L + t) * E + e];
    }
    out[b * E + e] = acc;
  }
}

__global__ void __launch_bounds__(kBlock)
din_pool_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ hist, const int32_t* __restrict__ len,
                    const float* __restrict__ dout, int B, int L, int E, float scale, float* __restrict__ dscores,
                    float* __restrict__ dhist, int acc_h) {
  const int lane = threadIdx.x & 63;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int n = len[b];
  // dp[t] = sum_e dout[e]*hist[t,e];  ds = p*(dp - sum_t p*dp); masked positions get no gradient
  float dot_pd = 0.f;
  for (int t = lane; t < L; t += 64) {
    float dp = 0.f;
    for (int e = 0; e < E; ++e) dp = dp + dout[b * E + e] * hist[(b * L + t) * E + e];
    dot_pd = dot_pd + probs[b * L + t] * dp;
  }
  dot_pd = wave_sum(dot_pd);
  for (int t = lane; t < L; t += 64) {
    float dp = 0.f;
    for (int e = 0; e < E; ++e) dp = dp + dout[b * E + e] * hist[(b * L + t) * E + e];
    const float ds = probs[b * L + t] * (dp - dot_pd);
    dscores[b * L + t] = (t < n) ? ds * scale : 0.f;
  }
  for (int64_t i = lane; i < static_cast<int64_t>(L) * E; i += 64) {
    const int t = static_cast<int>(i / E), e = static_cast<int>(i % E);
    const float v = probs[b * L + t] * dout[b * E + e];
    float* o = dhist + (b * L + t) * E + e;
    *o = acc_h ? (*o + v) : v;
  }
}

// ------------------------------------------------------------------------------------------------
// MMoE mixing
// ------------------------------------------------------------------------------------------------
constexpr int kMaxExperts = 32;

__global__ void __launch_bounds__(kBlock)
mmoe_gate_softmax_kernel(const float* __restrict__ logits, int64_t rows, int E, float* __restrict__ gates) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (r >= rows) return;
  float mx = -INFINITY;
  for (int e = 0; e < E; ++e) mx = fmaxf(mx, logits[r * E + e]);
  float den = 0.f;
  for (int e = 0; e < E; ++e) den = den + expf(logits[r * E + e] - mx);
  for (int e = 0; e < E; ++e) gates[r * E + e] = expf(logits[r * E + e] - mx) / den;
}

__global__ void __launch_bounds__(kBlock)
mmoe_mix_fwd_kernel(const float* __restrict__ experts, const float* __restrict__ gates, int T, int E, int B, int H,
                    float* __restrict__ out) {
  const int64_t n = static_cast<int64_t>(T) * B * H;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int h = static_cast<int>(i % H);
  const int64_t tb = i / H;
  const int64_t b = tb % B;
  float acc = 0.f;
  for (int e = 0; e < E; ++e) acc = acc + experts[(static_cast<int64_t>(e) * B + b) * H + h] * gates[tb * E + e];
  out[i] = acc;
}

// dexperts[e,b,h] = sum_t gates[t,b,e]*dout[t,b,h]
__global__ void __launch_bounds__(kBlock)
mmoe_mix_bwd_experts_kernel(const float* __restrict__ gates, const float* __restrict__ dout, int T, int E, int B,
                            int H, float* __restrict__ dexperts) {
  const int64_t n = static_cast<int64_t>(E) * B * H;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int h = static_cast<int>(i % H);
  const int64_t eb = i / H;
  const int64_t b = eb % B;
  const int e = static_cast<int>(eb / B);
  float acc = 0.f;
  for (int t = 0; t < T; ++t)
    acc = acc + gates[(static_cast<int64_t>(t) * B + b) * E + e] * dout[(static_cast<int64_t>(t) * B + b) * H + h];
  dexperts[i] = acc;
}

// one wave per (t, b): dgate[e] = sum_h dout[h]*experts[e,b,h]; dlogit = g*(dgate - sum_e g*dgate)
__global__ void __launch_bounds__(kBlock)
mmoe_mix_bwd_gate_kernel(const float* __restrict__ experts, const float* __restrict__ gates,
                         const float* __restrict__ dout, int T, int E, int B, int H,
                         float* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63;
  const int64_t tb = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (tb >= static_cast<int64_t>(T) * B) return;
  const int64_t b = tb % B;
  float dg[kMaxExperts];
  float dotg = 0.f;
  for (int e = 0; e < E; ++e) {
    float part = 0.f;
    for (int h = lane; h < H; h += 64) part = part + dout[tb * H + h] * experts[(static_cast<int64_t>(e) * B + b) * H + h];
    dg[e] = wave_sum(part);
    dotg = dotg + gates[tb * E + e] * dg[e];
  }
  if (lane == 0) {
    for (int e = 0; e < E; ++e) dlogits[tb * E + e] = gates[tb * E + e] * (dg[e] - dotg);
  }
}

inline int blocks_for(int64_t n) { return static_cast<int>(ceil_div(n, kBlock)); }

// ------------------------------------------------------------------------------------------------
// K15 DLRM dot interaction: all pairwise dot products of the F feature vectors of an example.
// One workgroup per example: the [F][D] block is staged in LDS (row stride D + 1: lanes of a wave read
// different rows), thread t owns pairs t, t + 256, ... (forward) or elements (f, d) (backward).
// Pair order = the reference's: for i in 0..F-1: for j in i + offset .. F-1 (model/dlrm.py:51-57).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int dot_pair_index(int i, int j, int F, int offset) {
  // pairs before row i: sum_{r < i} (F - r - offset) = i * (F - offset) - i (i - 1) / 2
  return i * (F - offset) - (i * (i - 1)) / 2 + (j - i - offset);
}

__global__ void __launch_bounds__(kBlock)
dot_interaction_fwd_kernel(const float* __restrict__ x, int F, int D, int x_stride, int offset, int P,
                           float* __restrict__ out, int out_stride) {
  extern __shared__ float xs[];  // [F][D + 1]
  const int b = blockIdx.x;
  const float* xb = x + static_cast<int64_t>(b) * x_stride;
  const int SD = D + 1;
  for (int e = threadIdx.x; e < F * D; e += kBlock) xs[(e / D) * SD + e % D] = xb[e];
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += kBlock) {
    // invert the pair index: rows are short (F <= 128), walk them
    int i = 0, rem = p;
    while (rem >= F - i - offset) { rem -= F - i - offset; ++i; }
    const int j = i + offset + rem;
    const float* a = xs + i * SD;
    const float* c = xs + j * SD;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc = acc + a[d] * c[d];
    out[static_cast<int64_t>(b) * out_stride + p] = acc;
  }
}

__global__ void __launch_bounds__(kBlock)
dot_interaction_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, int F, int D, int x_stride,
                           int offset, int P, int g_stride, float* __restrict__ dx, int dx_stride, int accumulate) {
  extern __shared__ float sm[];  // xs [F][D + 1], then gs [P]
  const int b = blockIdx.x;
  const int SD = D + 1;
  float* xs = sm;
  float* gs = sm + F * SD;
  const float* xb = x + static_cast<int64_t>(b) * x_stride;
  for (int e = threadIdx.x; e < F * D; e += kBlock) xs[(e / D) * SD + e % D] = xb[e];
  for (int p = threadIdx.x; p < P; p += kBlock) gs[p] = g[static_cast<int64_t>(b) * g_stride + p];
  __syncthreads();
  for (int e = threadIdx.x; e < F * D; e += kBlock) {
    const int f = e / D, d = e % D;
    float acc = 0.f;
    // d out(i, j) / d x_f = x_j (i == f) + x_i (j == f): partners in ascending order
    for (int o = 0; o < F; ++o) {
      if (o == f) {
        if (offset == 0) acc = acc + 2.f * gs[dot_pair_index(f, f, F, 0)] * xs[f * SD + d];
        continue;
      }
      const int i = o < f ? o : f, j = o < f ? f : o;
      acc = acc + gs[dot_pair_index(i, j, F, offset)] * xs[o * SD + d];
    }
    float* p = dx + static_cast<int64_t>(b) * dx_stride + e;
    *p = accumulate ? *p + acc : acc;
  }
}

// ------------------------------------------------------------------------------------------------
// The gradient of an embedding group's output, finished in ONE launch for all groups of a model:
//   dout[b, c] = (has_base ? dout[b, c] : 0)            what the consumers' GEMMs deposited (er_gemm accumulate)
//              + deferred terms, in the order recorded    row-sum broadcast (the wide logit), FM (g * (S - x))
//              + lambda * out[b, c]                       d/d(out) of the embedding-output L2 (layers/input_layer.py:369-375)
// instead of er_rowsum_bwd + er_fm_bwd + one er_axpy2d per regularised group (+ a zero fill for groups nobody
// differentiated).  Elementwise, HBM-bound: B * W * (2 or 3) floats.
// ------------------------------------------------------------------------------------------------
struct GradFinishMulti {
  int n;
  int start[8 + 1];
  er_grad_group g[8];
};

// a group without deferred terms on 16-byte aligned rows (the general path's usual case: base + lambda * out): 4 columns per
// lane, a 32-bit division per lane instead of a 64-bit one per element (DIN / MMoE: 29 us per step for this launch at 3 TB/s)
__host__ __device__ inline bool grad_finish_vec(const er_grad_group& g) {
  return g.n_terms == 0 && g.width % 4 == 0 && g.ld % 4 == 0 &&
         ((reinterpret_cast<uintptr_t>(g.dout) | reinterpret_cast<uintptr_t>(g.out)) & 15) == 0 &&
         static_cast<int64_t>(g.batch) * (g.width / 4) < (1ll << 31);
}

__global__ void __launch_bounds__(kBlock)
group_grad_finish_kernel(GradFinishMulti ma) {
  int i = 0;
  while (i + 1 < ma.n && static_cast<int>(blockIdx.x) >= ma.start[i + 1]) ++i;
  const er_grad_group& g = ma.g[i];
  if (grad_finish_vec(g)) {  // (uniform over the workgroup)
    const uint32_t w4 = static_cast<uint32_t>(g.width / 4);
    const uint32_t id4 = (blockIdx.x - static_cast<uint32_t>(ma.start[i])) * kBlock + threadIdx.x;
    const uint32_t b = id4 / w4;
    if (b >= static_cast<uint32_t>(g.batch)) return;
    const int64_t at = static_cast<int64_t>(b) * g.ld + static_cast<int64_t>(id4 - b * w4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.has_base) v = *reinterpret_cast<const float4*>(g.dout + at);
    if (g.lambda != 0.f) {  // grad_finish_value's arithmetic per component
      const float4 o = *reinterpret_cast<const float4*>(g.out + at);
      v.x = v.x + g.lambda * o.x; v.y = v.y + g.lambda * o.y; v.z = v.z + g.lambda * o.z; v.w = v.w + g.lambda * o.w;
    }
    *reinterpret_cast<float4*>(g.dout + at) = v;
    return;
  }
  const int64_t idx = (static_cast<int64_t>(blockIdx.x) - ma.start[i]) * kBlock + threadIdx.x;
  const int64_t b = idx / g.width;
  if (b >= g.batch) return;
  const int c = static_cast<int>(idx - b * g.width);
  g.dout[b * g.ld + c] = grad_finish_value(g, b, c);
}

// Several device-to-device copies in ONE launch (the parts of a device-resident batch that go into the step's static
// input buffers: labels / ids / tag lists / sequences were 7 - 9 hipMemcpyAsync of a few microseconds each per step).
// Item i: workgroups [start[i], start[i + 1]), 16 bytes per lane when both ends and the size allow, else 4 / 1.
constexpr int kCopyMulti = 16;
struct CopyMultiArgs {
  int n;
  int start[kCopyMulti + 1];
  const unsigned char* src[kCopyMulti];
  unsigned char* dst[kCopyMulti];
  long long bytes[kCopyMulti];
};

__global__ void __launch_bounds__(kBlock)
copy_multi_kernel(CopyMultiArgs a) {
  int i = 0;
  while (i + 1 < a.n && static_cast<int>(blockIdx.x) >= a.start[i + 1]) ++i;
  const long long lane = static_cast<long long>(blockIdx.x - a.start[i]) * kBlock + threadIdx.x;
  const unsigned char* s = a.src[i];
  unsigned char* d = a.dst[i];
  const long long n = a.bytes[i];
  const unsigned long long align = reinterpret_cast<unsigned long long>(s) | reinterpret_cast<unsigned long long>(d) |
                                   static_cast<unsigned long long>(n);
  if ((align & 15) == 0) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if (lane * 16 < n) reinterpret_cast<u32x4*>(d)[lane] = reinterpret_cast<const u32x4*>(s)[lane];
  } else if ((align & 3) == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long k = lane * 4 + j;
      if (k * 4 < n) reinterpret_cast<unsigned int*>(d)[k] = reinterpret_cast<const unsigned int*>(s)[k];
    }
  } else {
    for (int j = 0; j < 16; ++j) {
      const long long k = lane * 16 + j;
      if (k < n) d[k] = s[k];
    }
  }
}

// out[b, col0_p + j] = part_p[b * ld_p + j]: tf.concat(axis=1) of up to 8 row-major blocks (model/deepfm.py:75-83) in one
// launch; pure copy, HBM-bound.
struct ConcatArgs {
  int n, batch, out_ld;
  float* out;
  const float* src[8];
  int ld[8], col0[8 + 1];
  uint16_t* out_bf16;  // a bf16 copy of out (row stride ld_bf16; the padding columns up to it are zeroed) or nullptr
  int ld_bf16;
};

__global__ void __launch_bounds__(kBlock)
concat_cols_kernel(ConcatArgs a) {
  const int width = a.col0[a.n];
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t b = idx / width;
  if (b >= a.batch) return;
  const int c = static_cast<int>(idx - b * width);
  int p = 0;
  while (p + 1 < a.n && c >= a.col0[p + 1]) ++p;
  const float v = a.src[p][b * a.ld[p] + (c - a.col0[p])];
  a.out[b * a.out_ld + c] = v;
  if (a.out_bf16) {
    a.out_bf16[b * a.ld_bf16 + c] = f32_to_bf16_bits(v);
    if (c == 0) for (int j = width; j < a.ld_bf16; ++j) a.out_bf16[b * a.ld_bf16 + j] = 0;
  }
}

// the same copy for 16-byte aligned blocks (every col0, ld, out_ld a multiple of 4 floats; the host checks): one wave per
// kConcatRows rows, a lane moves 16 bytes per row and column chunk with the rows' loads in flight together - no 64-bit
// division per element, and the block a column belongs to is found once per column chunk with unrolled selects (a per-lane
// index into the by-value argument arrays would copy them to scratch).  MMoE's [8192, 1040] concat: 29 us at 2.3 TB/s before.
constexpr int kConcatRows = 4;
__global__ void __launch_bounds__(kBlock)
concat_cols_vec_kernel(ConcatArgs a) {
  const int width = a.col0[a.n];
  const int lane = threadIdx.x & 63;
  const int64_t r0 = (static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6)) * kConcatRows;
  if (r0 >= a.batch) return;
  for (int c = lane * 4; c < width; c += 64 * 4) {
    const float* src = a.src[0];
    int ld = a.ld[0], c0 = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      if (q < a.n && c >= a.col0[q]) { src = a.src[q]; ld = a.ld[q]; c0 = a.col0[q]; }
    }
    float4 v[kConcatRows];
#pragma unroll
    for (int u = 0; u < kConcatRows; ++u) {
      const int64_t r = r0 + u < a.batch ? r0 + u : a.batch - 1;
      v[u] = *reinterpret_cast<const float4*>(src + r * ld + (c - c0));
    }
#pragma unroll
    for (int u = 0; u < kConcatRows; ++u) {
      const int64_t r = r0 + u;
      if (r < a.batch) {
        *reinterpret_cast<float4*>(a.out + r * a.out_ld + c) = v[u];
        if (a.out_bf16) {
          uint16_t* ob = a.out_bf16 + r * a.ld_bf16 + c;
          ob[0] = f32_to_bf16_bits(v[u].x); ob[1] = f32_to_bf16_bits(v[u].y);
          ob[2] = f32_to_bf16_bits(v[u].z); ob[3] = f32_to_bf16_bits(v[u].w);
        }
      }
    }
  }
  if (a.out_bf16 && lane == 0) {
#pragma unroll
    for (int u = 0; u < kConcatRows; ++u)
      if (r0 + u < a.batch)
        for (int j = width; j < a.ld_bf16; ++j) a.out_bf16[(r0 + u) * a.ld_bf16 + j] = 0;
  }
}

// The elementwise backward of the TOP cross layer of a stack (its dout comes from outside the stack) by 64 x 64 tiles, with
// what the fused chain needs of it: du = dout * x0 (fp32 for the weight gradient, bf16 for the bf16 contraction), dx0 (+)=
// dout * (u + bias + diag * x), per-tile column sums of du (the bias gradient: er_colsum_partials_multi).  dout's own
// contribution to the gradient of x is added by the layer's input-gradient contraction (ER_EPI_CROSS_BWD).
__global__ void __launch_bounds__(kBlock)
cross_v2_bwd_top_kernel(const float* __restrict__ x0, const float* __restrict__ x, const float* __restrict__ u,
                        const float* __restrict__ bias, float diag, const float* __restrict__ dout, int ldg, int B, int d,
                        float* __restrict__ dx0, int ld0, int acc0, float* __restrict__ du, uint16_t* __restrict__ dub,
                        int lddub, float* __restrict__ partial) {
  __shared__ float sm[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * 64;
  float cs = 0.f;
  if (c < d) {
    const float bv = bias ? bias[c] : 0.f;
    float gv[16], x0v[16], uv[16], xv[16], ov[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int r = r0 + rl + 4 * k;
      r = r < B ? r : B - 1;
      const int64_t i = static_cast<int64_t>(r) * d + c;
      gv[k] = dout[static_cast<int64_t>(r) * ldg + c];
      x0v[k] = x0[i];
      uv[k] = u[i];
      xv[k] = diag != 0.f ? x[i] : 0.f;
      ov[k] = acc0 ? dx0[static_cast<int64_t>(r) * ld0 + c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int r = r0 + rl + 4 * k;
      if (r < B) {
        float tt = uv[k] + bv;
        if (diag != 0.f) tt = tt + diag * xv[k];
        const float a = gv[k] * tt;
        const float w = gv[k] * x0v[k];
        dx0[static_cast<int64_t>(r) * ld0 + c] = acc0 ? ov[k] + a : a;
        du[static_cast<int64_t>(r) * d + c] = w;
        if (dub) dub[static_cast<int64_t>(r) * lddub + c] = f32_to_bf16_bits(w);
        cs = cs + w;
      }
    }
  }
  if (dub && c >= d && c < lddub) {  // (the k-tail of the bf16 copy reads zeros)
    for (int k = 0; k < 16; ++k) {
      const int r = r0 + rl + 4 * k;
      if (r < B) dub[static_cast<int64_t>(r) * lddub + c] = 0;
    }
  }
  sm[rl][cl] = cs;
  __syncthreads();
  if (rl == 0 && c < d && partial) partial[static_cast<int64_t>(blockIdx.y) * d + c] = (sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]);
}


// ------------------------------------------------------------------------------------------------
// K9b: CIN, the compressed interaction network of xDeepFM (reference layers/keras/interaction.py:370-409):
//   x_{k+1}[b, n, d] = relu( sum_{h, m} W_k[n, h, m] * x_k[b, h, d] * x_0[b, m, d] + bias_k[n] ),
//   output = concat_k sum_d x_{k+1}[b, :, d].
// The reference materialises [B, H_k+1, H_k, H_0, D] (tile + multiply + two reduce_sums).  Here the outer product
// z[(b, d), (h, m)] = x_k[b, h, d] * x_0[b, m, d] is written once, k-contiguous, and the contraction over (h, m) is an
// MFMA GEMM (er_gemm_f32 NT against W_k viewed as [H_k+1, H_k * H_0]) whose output rows are (b, d): the layers after
// the first keep x_k as [B, D, H_k] (h innermost), which is exactly that output - no transposes.  x_0 is [B, H_0, D].
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
cin_outer_fwd_kernel(const float* __restrict__ xi, int64_t xi_sb, int xi_sh, int xi_sd, int H,
                     const float* __restrict__ x0, int H0, int D, int64_t rows, float* __restrict__ z) {
  // workgroup (x, y): row x = b * D + d, columns y * 256 .. of z; col = h * H0 + m  (32-bit index arithmetic only)
  const int K = H * H0;
  const int64_t row = blockIdx.x;
  const int col = blockIdx.y * kBlock + threadIdx.x;
  if (col >= K) return;
  const int64_t b = row / D;
  const int d = static_cast<int>(row - b * D);
  const int h = col / H0, m = col - h * H0;
  z[row * K + col] = xi[b * xi_sb + static_cast<int64_t>(h) * xi_sh + static_cast<int64_t>(d) * xi_sd] *
                     x0[(b * H0 + m) * D + d];
}

// fm = relu(c + bias) in place ([B * D, N]); pooled[b, col0 + n] = sum_d fm[(b, d), n] (d ascending)
__global__ void __launch_bounds__(kBlock)
cin_act_pool_fwd_kernel(float* __restrict__ c, const float* __restrict__ bias, int64_t B, int D, int N,
                        float* __restrict__ pooled, int pooled_ld, int col0) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= B * N) return;
  const int64_t b = i / N;
  const int n = static_cast<int>(i - b * N);
  const float bv = bias[n];
  float s = 0.f;
  for (int d = 0; d < D; ++d) {
    float* p = c + (b * D + d) * N + n;
    float v = *p + bv;
    v = v > 0.f ? v : 0.f;
    *p = v;
    s = s + v;
  }
  pooled[b * pooled_ld + col0 + n] = s;
}

// dc[(b, d), n] = (dpooled[b, col0 + n] + dnext[(b, d), n]) * (fm > 0)      (dnext may be nullptr; dc may alias dnext)
__global__ void __launch_bounds__(kBlock)
cin_act_pool_bwd_kernel(const float* __restrict__ fm, const float* __restrict__ dpooled, int dpooled_ld, int col0,
                        const float* dnext, int64_t B, int D, int N, float* dc) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= B * D * N) return;
  const int64_t row = i / N;
  const int n = static_cast<int>(i - row * N);
  const int64_t b = row / D;
  float g = dpooled[b * dpooled_ld + col0 + n];
  if (dnext) g = g + dnext[i];
  dc[i] = fm[i] > 0.f ? g : 0.f;
}

// One workgroup per row (b, d) of dz [rows, H * H0]; the row is staged in LDS (coalesced 16-byte reads by all lanes),
// then reduced from there:
//   dxi[b, h, d]  = sum_m dz[row, h * H0 + m] * x0[b, m, d]       (m ascending; written, or added when add_xi)
//   dx0[b, m, d] += sum_h dz[row, h * H0 + m] * xi[b, h, d]       (h ascending)
// The first layer has xi == x0: both terms land in dx0 (dxi == dx0, add_xi set).
__global__ void __launch_bounds__(kBlock)
cin_outer_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ xi, int64_t xi_sb, int xi_sh, int xi_sd,
                     int H, const float* __restrict__ x0, int H0, int D, float* dxi, int add_xi, float* dx0) {
  extern __shared__ float sh[];  // [H] xi column, [H0] x0 column, [H * H0] the dz row
  float* s_xi = sh;
  float* s_x0 = sh + H;
  float* s_dz = sh + H + H0;
  const int64_t row = blockIdx.x;
  const int64_t b = row / D;
  const int d = static_cast<int>(row - b * D);
  const int K = H * H0;
  const float* r = dz + row * static_cast<int64_t>(K);
  for (int h = threadIdx.x; h < H; h += kBlock) s_xi[h] = xi[b * xi_sb + static_cast<int64_t>(h) * xi_sh + static_cast<int64_t>(d) * xi_sd];
  for (int m = threadIdx.x; m < H0; m += kBlock) s_x0[m] = x0[(b * H0 + m) * D + d];
  for (int k = threadIdx.x; k < K; k += kBlock) s_dz[k] = r[k];
  __syncthreads();
  // the x0 term first, then the xi term: when dxi aliases dx0 both updates of one element are made by different
  // threads - keep them in two phases separated by a barrier
  for (int m = threadIdx.x; m < H0; m += kBlock) {
    float s = 0.f;
    for (int h = 0; h < H; ++h) s = s + s_dz[h * H0 + m] * s_xi[h];
    float* o = dx0 + (b * H0 + m) * D + d;
    *o = *o + s;
  }
  __syncthreads();
  for (int h = threadIdx.x; h < H; h += kBlock) {
    float s = 0.f;
    for (int m = 0; m < H0; ++m) s = s + s_dz[h * H0 + m] * s_x0[m];
    float* o = dxi + b * xi_sb + static_cast<int64_t>(h) * xi_sh + static_cast<int64_t>(d) * xi_sd;
    *o = add_xi ? *o + s : s;
  }
}

template <int V>
__device__ __forceinline__ void ld_n(float (&r)[V], const float* p) {
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  } else {
    r[0] = p[0];
  }
}
template <int V>
__device__ __forceinline__ void st_n(float* p, const float (&r)[V]) {
  if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]); else p[0] = r[0];
}

// cross_v2_bwd with caller-provided destinations (round 4): dx0 and dx are ACCUMULATED into buffers the consumers of the
// same tensors share - an embedding group's gradient buffer (row stride ld), or the tensor another gradient of x_l already
// went to - so that autograd sums nothing (the 3-layer DCN-v2 backbone ran nine torch add kernels per step for these).
// dx == nullptr: x IS x0 (the first cross layer): its gradient joins dx0's.  16-byte lanes when d and the strides allow.
template <int V>
__global__ void __launch_bounds__(kBlock)
cross_v2_bwd_acc_kernel(const float* __restrict__ x0, const float* __restrict__ x, const float* __restrict__ u,
                        const float* __restrict__ bias, float diag, const float* __restrict__ dout, int ldg, int B, int d,
                        float* __restrict__ dx0, int ld0, int acc0, float* __restrict__ dx, int ldx, int accx,
                        float* __restrict__ du) {
  const int per_row = d / V;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= static_cast<int64_t>(B) * per_row) return;
  const int64_t r = t / per_row;
  const int c = static_cast<int>(t - r * per_row) * V;
  const int64_t i = r * d + c;
  float x0v[V], xv[V], uv[V], gv[V], bv[V], o0[V], ox[V];
  ld_n<V>(x0v, x0 + i);
  ld_n<V>(uv, u + i);
  ld_n<V>(gv, dout + r * ldg + c);
  if (diag != 0.f) ld_n<V>(xv, x + i);
  if (bias) ld_n<V>(bv, bias + c);
  if (acc0) ld_n<V>(o0, dx0 + r * ld0 + c);
  if (dx && accx) ld_n<V>(ox, dx + r * ldx + c);
  float r0[V], rx[V], ru[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    float tt = uv[j] + (bias ? bv[j] : 0.f);
    if (diag != 0.f) tt = tt + diag * xv[j];
    const float g = gv[j];
    const float a = g * tt;                                              // d out / d x0 (the Hadamard factor's side)
    const float b = g + (diag != 0.f ? g * x0v[j] * diag : 0.f);         // d out / d x (residual + diagonal term)
    ru[j] = g * x0v[j];
    float v0 = acc0 ? (o0[j] + a) : a;
    if (!dx) v0 = v0 + b;
    r0[j] = v0;
    rx[j] = (dx && accx) ? (ox[j] + b) : b;
  }
  st_n<V>(dx0 + r * ld0 + c, r0);
  if (dx) st_n<V>(dx + r * ldx + c, rx);
  st_n<V>(du + i, ru);
}

}  // namespace er

extern "C" {

int er_fm_fwd(const float* x, int32_t B, int32_t F, int32_t D, int32_t x_stride, float* fm_out, float* sum_out,
              er_stream_t stream) {
  ER_REQUIRE(x && fm_out && sum_out && B > 0 && F > 0 && D > 0, "er_fm_fwd: bad arguments");
  hipStream_t s = er::as_stream(stream);
  const bool vec = (D % 4 == 0) && (x_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (vec) {
    hipLaunchKernelGGL(er::fm_fwd_kernel<4>, dim3(er::blocks_for(static_cast<int64_t>(B) * (D / 4))), dim3(er::kBlock),
                       0, s, x, B, F, D, x_stride, fm_out, sum_out);
  } else {
    hipLaunchKernelGGL(er::fm_fwd_kernel<1>, dim3(er::blocks_for(static_cast<int64_t>(B) * D)), dim3(er::kBlock), 0, s,
                       x, B, F, D, x_stride, fm_out, sum_out);
  }
  ER_LAUNCH_CHECK();
  return 0;
}

int er_wide_fm_concat(const float* wide, int32_t n_w, int32_t ld_w, const float* fm_x, int32_t F, int32_t D, int32_t ld_x,
                      const float* deep, int32_t n_d, int32_t ld_d, int32_t B, float* out, int32_t ld_out, float* sum_out,
                      er_stream_t stream) {
  ER_REQUIRE(wide && fm_x && deep && out && sum_out && B > 0 && n_w > 0 && F > 0 && D > 0 && n_d > 0 &&
                 ld_out >= 1 + D + n_d && ld_w >= n_w && ld_x >= F * D && ld_d >= n_d,
             "er_wide_fm_concat: bad arguments");
  const bool vec = (D % 4 == 0) && (ld_x % 4 == 0) && ((reinterpret_cast<uintptr_t>(fm_x) & 15) == 0);
  const int fm_blocks = er::blocks_for(static_cast<int64_t>(B) * (vec ? D / 4 : D));
  const int rs_blocks = er::blocks_for(static_cast<int64_t>(B) * 4);
  const int cp_blocks = er::blocks_for(static_cast<int64_t>(B) * n_d);
  hipStream_t s = er::as_stream(stream);
  if (vec) {
    hipLaunchKernelGGL(er::wide_fm_concat_kernel<4>, dim3(fm_blocks + rs_blocks + cp_blocks), dim3(er::kBlock), 0, s, wide, n_w,
                       ld_w, fm_x, F, D, ld_x, deep, n_d, ld_d, B, out, ld_out, sum_out, fm_blocks, rs_blocks);
  } else {
    hipLaunchKernelGGL(er::wide_fm_concat_kernel<1>, dim3(fm_blocks + rs_blocks + cp_blocks), dim3(er::kBlock), 0, s, wide, n_w,
                       ld_w, fm_x, F, D, ld_x, deep, n_d, ld_d, B, out, ld_out, sum_out, fm_blocks, rs_blocks);
  }
  ER_LAUNCH_CHECK();
  return 0;
}

int er_fm_bwd(const float* x, const float* sum_saved, const float* g, int32_t B, int32_t F, int32_t D,
              int32_t x_stride, float* dx, int32_t dx_stride, int accumulate, er_stream_t stream) {
  ER_REQUIRE(x && sum_saved && g && dx && B > 0 && F > 0 && D > 0, "er_fm_bwd: bad arguments");
  hipLaunchKernelGGL(er::fm_bwd_kernel<1>, dim3(er::blocks_for(static_cast<int64_t>(B) * F * D)), dim3(er::kBlock), 0,
                     er::as_stream(stream), x, sum_saved, g, B, F, D, x_stride, dx, dx_stride, accumulate);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_rowsum_fwd(const float* x, int32_t B, int32_t n, int32_t x_stride, float* out, er_stream_t stream) {
  ER_REQUIRE(x && out && B > 0 && n > 0, "er_rowsum_fwd: bad arguments");
  hipLaunchKernelGGL(er::rowsum_fwd_kernel, dim3(er::blocks_for(static_cast<int64_t>(B) * 4)), dim3(er::kBlock), 0,
                     er::as_stream(stream), x, B, n, x_stride, out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_rowsum_bwd(const float* g, int32_t B, int32_t n, float* dx, int32_t dx_stride, int accumulate,
                  er_stream_t stream) {
  ER_REQUIRE(g && dx && B > 0 && n > 0, "er_rowsum_bwd: bad arguments");
  hipLaunchKernelGGL(er::rowsum_bwd_kernel, dim3(er::blocks_for(static_cast<int64_t>(B) * n)), dim3(er::kBlock), 0,
                     er::as_stream(stream), g, B, n, dx, dx_stride, accumulate);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_axpy2d(const float* x, int32_t x_stride, float alpha, float* y, int32_t y_stride, int32_t rows, int32_t cols,
              int accumulate, er_stream_t stream) {
  ER_REQUIRE(x && y && rows > 0 && cols > 0, "er_axpy2d: bad arguments");
  hipLaunchKernelGGL(er::axpy2d_kernel, dim3(er::blocks_for(static_cast<int64_t>(rows) * cols)), dim3(er::kBlock), 0,
                     er::as_stream(stream), x, x_stride, alpha, y, y_stride, rows, cols, accumulate);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_cross_v1_fwd(const float* x0, const float* w, const float* b, int32_t B, int32_t d, int32_t L, float* out,
                    float* xl_dots, er_stream_t stream) {
  ER_REQUIRE(x0 && w && b && out && xl_dots && B > 0 && L > 0, "er_cross_v1_fwd: bad arguments");
  ER_REQUIRE(d > 0 && d <= 64 * er::kCrossE, "er_cross_v1_fwd: d=%d exceeds %d", d, 64 * er::kCrossE);
  hipLaunchKernelGGL(er::cross_v1_fwd_kernel, dim3(static_cast<int>(er::ceil_div(B, er::kBlock / 64))),
                     dim3(er::kBlock), 0, er::as_stream(stream), x0, w, b, B, d, L, out, xl_dots);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_cross_v1_bwd_partials(int32_t B) { return B < 512 ? B : 512; }

int er_cross_v1_bwd(const float* x0, const float* w, const float* b, const float* xl_dots, const float* dout,
                    int32_t B, int32_t d, int32_t L, float* dx0, float* dw_partials, float* db_partials,
                    er_stream_t stream) {
  ER_REQUIRE(x0 && w && b && xl_dots && dout && dx0 && dw_partials && db_partials, "er_cross_v1_bwd: null argument");
  ER_REQUIRE(d > 0 && d <= 64 * er::kCrossE, "er_cross_v1_bwd: d=%d exceeds %d", d, 64 * er::kCrossE);
  const size_t lds = sizeof(float) * 2 * static_cast<size_t>(L) * d;
  ER_REQUIRE(lds <= 160 * 1024, "er_cross_v1_bwd: L*d too large for LDS accumulators");
  if (lds > 64 * 1024) {
    ER_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(er::cross_v1_bwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  }
  hipLaunchKernelGGL(er::cross_v1_bwd_kernel, dim3(er_cross_v1_bwd_partials(B)), dim3(64), lds, er::as_stream(stream),
                     x0, w, b, xl_dots, dout, B, d, L, dx0, dw_partials, db_partials);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_cross_v2_epilogue_fwd(const float* x0, const float* x, const float* u, const float* bias, float diag_scale,
                             int32_t B, int32_t d, float* out, er_stream_t stream) {
  ER_REQUIRE(x0 && x && u && out && B > 0 && d > 0, "er_cross_v2_epilogue_fwd: bad arguments");
  const int64_t n = static_cast<int64_t>(B) * d;
  hipLaunchKernelGGL(er::cross_v2_fwd_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, er::as_stream(stream), x0,
                     x, u, bias, diag_scale, n, d, out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_cross_v2_epilogue_bwd_acc(const float* x0, const float* x, const float* u, const float* bias, float diag_scale,
                                 const float* dout, int32_t ld_dout, int32_t B, int32_t d, float* dx0, int32_t ld_dx0, int accumulate_dx0,
                                 float* dx, int32_t ld_dx, int accumulate_dx, float* du, er_stream_t stream) {
  ER_REQUIRE(x0 && x && u && dout && dx0 && du && B > 0 && d > 0 && ld_dx0 >= d && (!dx || ld_dx >= d) && ld_dout >= d,
             "er_cross_v2_epilogue_bwd_acc: bad arguments");
  const uintptr_t al = reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(u) |
                       reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dx0) | reinterpret_cast<uintptr_t>(dx) |
                       reinterpret_cast<uintptr_t>(du) | reinterpret_cast<uintptr_t>(bias);
  const bool vec = d % 4 == 0 && ld_dx0 % 4 == 0 && (!dx || ld_dx % 4 == 0) && ld_dout % 4 == 0 && (al & 15) == 0;
  const int64_t n = static_cast<int64_t>(B) * (vec ? d / 4 : d);
  if (vec)
    hipLaunchKernelGGL(er::cross_v2_bwd_acc_kernel<4>, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, er::as_stream(stream), x0, x,
                       u, bias, diag_scale, dout, ld_dout, B, d, dx0, ld_dx0, accumulate_dx0, dx, ld_dx, accumulate_dx, du);
  else
    hipLaunchKernelGGL(er::cross_v2_bwd_acc_kernel<1>, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, er::as_stream(stream), x0, x,
                       u, bias, diag_scale, dout, ld_dout, B, d, dx0, ld_dx0, accumulate_dx0, dx, ld_dx, accumulate_dx, du);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_cross_v2_epilogue_bwd(const float* x0, const float* x, const float* u, const float* bias, float diag_scale,
                             const float* dout, int32_t B, int32_t d, float* dx0, int accumulate_dx0, float* dx,
                             float* du, er_stream_t stream) {
  ER_REQUIRE(x0 && x && u && dout && dx0 && dx && du && B > 0 && d > 0, "er_cross_v2_epilogue_bwd: bad arguments");
  const int64_t n = static_cast<int64_t>(B) * d;
  hipLaunchKernelGGL(er::cross_v2_bwd_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, er::as_stream(stream), x0,
                     x, u, bias, diag_scale, dout, n, d, dx0, accumulate_dx0, dx, du);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_din_concat_fwd(const float* query, const float* hist, int32_t B, int32_t L, int32_t E, float* out,
                      er_stream_t stream) {
  ER_REQUIRE(query && hist && out && B > 0 && L > 0 && E > 0, "er_din_concat_fwd: bad arguments");
  hipLaunchKernelGGL(er::din_concat_fwd_kernel, dim3(er::blocks_for(static_cast<int64_t>(B) * L * E)),
                     dim3(er::kBlock), 0, er::as_stream(stream), query, hist, B, L, E, out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_din_concat_bwd(const float* query, const float* hist, const float* dout, int32_t B, int32_t L, int32_t E,
                      float* dquery, int acc_q, float* dhist, int acc_h, er_stream_t stream) {
  ER_REQUIRE(query && hist && dout && B > 0 && L > 0 && E > 0, "er_din_concat_bwd: bad arguments");
  hipStream_t s = er::as_stream(stream);
  if (er::din_fast_ok(L, E, query, hist, dout) && er::din_fast_ok(L, E, dquery, dhist)) {
    hipLaunchKernelGGL(er::din_concat_bwd_fast_kernel, dim3(static_cast<int>(er::ceil_div(B, er::kBlock / 64))),
                       dim3(er::kBlock), 0, s, query, hist, dout, B, L, E, dquery, acc_q, dhist, acc_h);
    ER_LAUNCH_CHECK();
    return 0;
  }
  if (dhist) {
    hipLaunchKernelGGL(er::din_concat_bwd_hist_kernel, dim3(er::blocks_for(static_cast<int64_t>(B) * L * E)),
                       dim3(er::kBlock), 0, s, query, dout, B, L, E, dhist, acc_h);
    ER_LAUNCH_CHECK();
  }
  if (dquery) {
    hipLaunchKernelGGL(er::din_concat_bwd_query_kernel, dim3(er::blocks_for(static_cast<int64_t>(B) * E)),
                       dim3(er::kBlock), 0, s, hist, dout, B, L, E, dquery, acc_q);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

int er_din_pool_fwd(const float* scores, const float* hist, const int32_t* seq_len, int32_t B, int32_t L, int32_t E,
                    float scale, float* probs_out, float* out, er_stream_t stream) {
  ER_REQUIRE(scores && hist && seq_len && probs_out && out && B > 0 && L > 0 && E > 0, "er_din_pool_fwd: bad arguments");
  if (er::din_fast_ok(L, E, hist, out)) {
    hipLaunchKernelGGL(er::din_pool_fwd_fast_kernel, dim3(static_cast<int>(er::ceil_div(B, er::kBlock / 64))),
                       dim3(er::kBlock), 0, er::as_stream(stream), scores, hist, seq_len, B, L, E, scale, probs_out, out);
    ER_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(er::din_pool_fwd_kernel, dim3(static_cast<int>(er::ceil_div(B, er::kBlock / 64))),
                     dim3(er::kBlock), 0, er::as_stream(stream), scores, hist, seq_len, B, L, E, scale, probs_out, out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_din_pool_bwd(const float* probs, const float* hist, const int32_t* seq_len, const float* dout, int32_t B,
                    int32_t L, int32_t E, float scale, float* dscores, float* dhist, int acc_h, er_stream_t stream) {
  ER_REQUIRE(probs && hist && seq_len && dout && dscores && dhist, "er_din_pool_bwd: null argument");
  if (er::din_fast_ok(L, E, hist, dout, dhist)) {
    hipLaunchKernelGGL(er::din_pool_bwd_fast_kernel, dim3(static_cast<int>(er::ceil_div(B, er::kBlock / 64))),
                       dim3(er::kBlock), 0, er::as_stream(stream), probs, hist, seq_len, dout, B, L, E, scale, dscores,
                       dhist, acc_h);
    ER_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(er::din_pool_bwd_kernel, dim3(static_cast<int>(er::ceil_div(B, er::kBlock / 64))),
                     dim3(er::kBlock), 0, er::as_stream(stream), probs, hist, seq_len, dout, B, L, E, scale, dscores,
                     dhist, acc_h);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_mmoe_mix_fwd(const float* experts, const float* gate_logits, int32_t T, int32_t E, int32_t B, int32_t H,
                    float* gates_out, float* out, er_stream_t stream) {
  ER_REQUIRE(experts && gate_logits && gates_out && out && T > 0 && E > 0 && B > 0 && H > 0,
             "er_mmoe_mix_fwd: bad arguments");
  ER_REQUIRE(E <= er::kMaxExperts, "er_mmoe_mix_fwd: at most %d experts", er::kMaxExperts);
  hipStream_t s = er::as_stream(stream);
  hipLaunchKernelGGL(er::mmoe_gate_softmax_kernel, dim3(er::blocks_for(static_cast<int64_t>(T) * B)), dim3(er::kBlock),
                     0, s, gate_logits, static_cast<int64_t>(T) * B, E, gates_out);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::mmoe_mix_fwd_kernel, dim3(er::blocks_for(static_cast<int64_t>(T) * B * H)), dim3(er::kBlock),
                     0, s, experts, gates_out, T, E, B, H, out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_mmoe_mix_bwd(const float* experts, const float* gates, const float* dout, int32_t T, int32_t E, int32_t B,
                    int32_t H, float* dexperts, float* dgate_logits, er_stream_t stream) {
  ER_REQUIRE(experts && gates && dout && dexperts && dgate_logits, "er_mmoe_mix_bwd: null argument");
  ER_REQUIRE(E <= er::kMaxExperts, "er_mmoe_mix_bwd: at most %d experts", er::kMaxExperts);
  hipStream_t s = er::as_stream(stream);
  hipLaunchKernelGGL(er::mmoe_mix_bwd_experts_kernel, dim3(er::blocks_for(static_cast<int64_t>(E) * B * H)),
                     dim3(er::kBlock), 0, s, gates, dout, T, E, B, H, dexperts);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::mmoe_mix_bwd_gate_kernel,
                     dim3(static_cast<int>(er::ceil_div(static_cast<int64_t>(T) * B, er::kBlock / 64))),
                     dim3(er::kBlock), 0, s, experts, gates, dout, T, E, B, H, dgate_logits);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_dot_interaction_fwd(const float* x, int32_t B, int32_t F, int32_t D, int32_t x_stride, int self_interaction,
                           float* out, int32_t out_stride, er_stream_t stream) {
  ER_REQUIRE(x && out && B > 0 && F > 1 && D > 0 && x_stride >= F * D, "er_dot_interaction_fwd: bad arguments");
  const int offset = self_interaction ? 0 : 1;
  const int P = F * (F - 1) / 2 + (self_interaction ? F : 0);
  ER_REQUIRE(out_stride >= P, "er_dot_interaction_fwd: out_stride %d < %d pairs", out_stride, P);
  const size_t lds = sizeof(float) * static_cast<size_t>(F) * (D + 1);
  ER_REQUIRE(lds <= 60 * 1024, "er_dot_interaction_fwd: F * (D + 1) = %d floats exceed the LDS budget", F * (D + 1));
  hipLaunchKernelGGL(er::dot_interaction_fwd_kernel, dim3(B), dim3(er::kBlock), lds, er::as_stream(stream), x, F, D,
                     x_stride, offset, P, out, out_stride);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_dot_interaction_bwd(const float* x, const float* g, int32_t B, int32_t F, int32_t D, int32_t x_stride,
                           int self_interaction, int32_t g_stride, float* dx, int32_t dx_stride, int accumulate,
                           er_stream_t stream) {
  ER_REQUIRE(x && g && dx && B > 0 && F > 1 && D > 0 && x_stride >= F * D && dx_stride >= F * D,
             "er_dot_interaction_bwd: bad arguments");
  const int offset = self_interaction ? 0 : 1;
  const int P = F * (F - 1) / 2 + (self_interaction ? F : 0);
  ER_REQUIRE(g_stride >= P, "er_dot_interaction_bwd: g_stride %d < %d pairs", g_stride, P);
  const size_t lds = sizeof(float) * (static_cast<size_t>(F) * (D + 1) + P);
  ER_REQUIRE(lds <= 60 * 1024, "er_dot_interaction_bwd: %zu bytes exceed the LDS budget", lds);
  hipLaunchKernelGGL(er::dot_interaction_bwd_kernel, dim3(B), dim3(er::kBlock), lds, er::as_stream(stream), x, g, F, D,
                     x_stride, offset, P, g_stride, dx, dx_stride, accumulate);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_cin_outer_fwd(const float* xi, int64_t xi_stride_b, int32_t xi_stride_h, int32_t xi_stride_d, int32_t H,
                     const float* x0, int32_t H0, int32_t D, int64_t B, float* z, er_stream_t stream) {
  ER_REQUIRE(xi && x0 && z && B > 0 && H > 0 && H0 > 0 && D > 0, "er_cin_outer_fwd: bad arguments");
  ER_REQUIRE(B * D < 0x7FFFFFFFLL && static_cast<int64_t>(H) * H0 < (1 << 24), "er_cin_outer_fwd: too many rows / columns");
  dim3 grid(static_cast<unsigned>(B * D), static_cast<unsigned>(er::ceil_div(static_cast<int64_t>(H) * H0, er::kBlock)));
  hipLaunchKernelGGL(er::cin_outer_fwd_kernel, grid, dim3(er::kBlock), 0, er::as_stream(stream), xi, xi_stride_b,
                     xi_stride_h, xi_stride_d, H, x0, H0, D, B * D, z);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_cin_act_pool_fwd(float* c, const float* bias, int64_t B, int32_t D, int32_t N, float* pooled, int32_t pooled_ld,
                        int32_t col0, er_stream_t stream) {
  ER_REQUIRE(c && bias && pooled && B > 0 && D > 0 && N > 0 && pooled_ld >= col0 + N, "er_cin_act_pool_fwd: bad arguments");
  hipLaunchKernelGGL(er::cin_act_pool_fwd_kernel, dim3(er::blocks_for(B * N)), dim3(er::kBlock), 0, er::as_stream(stream),
                     c, bias, B, D, N, pooled, pooled_ld, col0);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_cin_act_pool_bwd(const float* fm, const float* dpooled, int32_t dpooled_ld, int32_t col0, const float* dnext,
                        int64_t B, int32_t D, int32_t N, float* dc, er_stream_t stream) {
  ER_REQUIRE(fm && dpooled && dc && B > 0 && D > 0 && N > 0 && dpooled_ld >= col0 + N, "er_cin_act_pool_bwd: bad arguments");
  hipLaunchKernelGGL(er::cin_act_pool_bwd_kernel, dim3(er::blocks_for(B * D * N)), dim3(er::kBlock), 0,
                     er::as_stream(stream), fm, dpooled, dpooled_ld, col0, dnext, B, D, N, dc);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_cin_outer_bwd(const float* dz, const float* xi, int64_t xi_stride_b, int32_t xi_stride_h, int32_t xi_stride_d,
                     int32_t H, const float* x0, int32_t H0, int32_t D, int64_t B, float* dxi, int add_xi, float* dx0,
                     er_stream_t stream) {
  ER_REQUIRE(dz && xi && x0 && dxi && dx0 && B > 0 && H > 0 && H0 > 0 && D > 0, "er_cin_outer_bwd: bad arguments");
  const size_t lds_bytes = (static_cast<size_t>(H) + H0 + static_cast<size_t>(H) * H0) * sizeof(float);
  ER_REQUIRE(B * D < 0x7FFFFFFFLL && lds_bytes <= 60 * 1024, "er_cin_outer_bwd: too many rows / features");
  ER_REQUIRE(dxi != dx0 || add_xi, "er_cin_outer_bwd: dxi aliasing dx0 must add");
  hipLaunchKernelGGL(er::cin_outer_bwd_kernel, dim3(static_cast<unsigned>(B * D)), dim3(er::kBlock), lds_bytes,
                     er::as_stream(stream), dz, xi, xi_stride_b, xi_stride_h,
                     xi_stride_d, H, x0, H0, D, dxi, add_xi, dx0);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_group_grad_finish(const er_grad_group* groups, int n, er_stream_t stream) {
  ER_REQUIRE(groups && n >= 1, "er_group_grad_finish: bad arguments");
  for (int i0 = 0; i0 < n; i0 += 8) {
    er::GradFinishMulti ma;
    ma.n = n - i0 < 8 ? n - i0 : 8;
    ma.start[0] = 0;
    for (int i = 0; i < ma.n; ++i) {
      const er_grad_group& g = groups[i0 + i];
      ER_REQUIRE(g.dout && g.out && g.batch > 0 && g.width > 0 && g.ld >= g.width && g.n_terms >= 0 && g.n_terms <= 4,
                 "er_group_grad_finish: group %d: bad descriptor", i0 + i);
      for (int t = 0; t < g.n_terms; ++t)
        ER_REQUIRE(g.terms[t].g && g.terms[t].width > 0 && g.terms[t].col0 >= 0 &&
                       g.terms[t].col0 + g.terms[t].width <= g.width &&
                       (g.terms[t].kind == ER_GRAD_TERM_ROWSUM ||
                        (g.terms[t].kind == ER_GRAD_TERM_FM && g.terms[t].saved && g.terms[t].dim > 0)),
                   "er_group_grad_finish: group %d term %d: bad descriptor", i0 + i, t);
      ma.g[i] = g;
      const int64_t lanes = er::grad_finish_vec(g) ? static_cast<int64_t>(g.batch) * (g.width / 4)
                                                   : static_cast<int64_t>(g.batch) * g.width;
      ma.start[i + 1] = ma.start[i] + static_cast<int>(er::ceil_div(lanes, er::kBlock));
    }
    hipLaunchKernelGGL(er::group_grad_finish_kernel, dim3(static_cast<unsigned>(ma.start[ma.n])), dim3(er::kBlock), 0,
                       er::as_stream(stream), ma);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

int er_copy_multi(const void* const* src, void* const* dst, const int64_t* bytes, int n, er_stream_t stream) {
  ER_REQUIRE(src && dst && bytes && n >= 1, "er_copy_multi: bad arguments");
  for (int base = 0; base < n; base += er::kCopyMulti) {
    er::CopyMultiArgs a;
    a.n = 0;
    a.start[0] = 0;
    for (int i = base; i < n && i < base + er::kCopyMulti; ++i) {
      ER_REQUIRE(bytes[i] >= 0 && (bytes[i] == 0 || (src[i] && dst[i])), "er_copy_multi: item %d: bad descriptor", i);
      if (bytes[i] == 0) continue;
      a.src[a.n] = static_cast<const unsigned char*>(src[i]);
      a.dst[a.n] = static_cast<unsigned char*>(dst[i]);
      a.bytes[a.n] = bytes[i];
      a.start[a.n + 1] = a.start[a.n] + static_cast<int>(er::ceil_div(er::ceil_div(bytes[i], 16), er::kBlock));  // 16 B per lane
      ++a.n;
    }
    if (a.n == 0) continue;
    hipLaunchKernelGGL(er::copy_multi_kernel, dim3(static_cast<unsigned>(a.start[a.n])), dim3(er::kBlock), 0,
                       er::as_stream(stream), a);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

int er_cross_v2_bwd_top(const float* x0, const float* x, const float* u, const float* bias, float diag_scale, const float* dout,
                        int32_t ld_dout, int32_t B, int32_t d, float* dx0, int32_t ld_dx0, int accumulate_dx0, float* du,
                        uint16_t* du_bf16, int32_t ld_du_bf16, float* partial, er_stream_t stream) {
  ER_REQUIRE(x0 && u && dout && dx0 && du && B > 0 && d > 0 && ld_dout >= d && ld_dx0 >= d && (diag_scale == 0.f || x) &&
                 (!du_bf16 || ld_du_bf16 >= d),
             "er_cross_v2_bwd_top: bad arguments");
  dim3 grid(static_cast<unsigned>(er::ceil_div(du_bf16 ? ld_du_bf16 : d, 64)), static_cast<unsigned>(er::ceil_div(B, 64)));
  hipLaunchKernelGGL(er::cross_v2_bwd_top_kernel, grid, dim3(er::kBlock), 0, er::as_stream(stream), x0, x, u, bias, diag_scale,
                     dout, ld_dout, B, d, dx0, ld_dx0, accumulate_dx0, du, du_bf16, ld_du_bf16, partial);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_concat_cols(const float* const* parts, const int32_t* widths, const int32_t* lds, int n, int32_t batch, float* out,
                   int32_t out_ld, er_stream_t stream) {
  return er_concat_cols_b16(parts, widths, lds, n, batch, out, out_ld, nullptr, 0, stream);
}

int er_concat_cols_b16(const float* const* parts, const int32_t* widths, const int32_t* lds, int n, int32_t batch, float* out,
                       int32_t out_ld, uint16_t* out_bf16, int32_t ld_bf16, er_stream_t stream) {
  ER_REQUIRE(parts && widths && lds && out && n >= 1 && n <= 8 && batch > 0, "er_concat_cols: bad arguments (n <= 8)");
  er::ConcatArgs a;
  a.n = n; a.batch = batch; a.out_ld = out_ld; a.out = out; a.out_bf16 = out_bf16; a.ld_bf16 = ld_bf16;
  a.col0[0] = 0;
  for (int i = 0; i < n; ++i) {
    ER_REQUIRE(parts[i] && widths[i] > 0 && lds[i] >= widths[i], "er_concat_cols: part %d: bad descriptor", i);
    a.src[i] = parts[i]; a.ld[i] = lds[i];
    a.col0[i + 1] = a.col0[i] + widths[i];
  }
  ER_REQUIRE(out_ld >= a.col0[n] && (!out_bf16 || ld_bf16 >= a.col0[n]), "er_concat_cols: out_ld too small");
  bool vec = out_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && a.col0[n] % 4 == 0;
  for (int i = 0; i < n; ++i)
    vec = vec && a.col0[i] % 4 == 0 && lds[i] % 4 == 0 && (reinterpret_cast<uintptr_t>(parts[i]) & 15) == 0;
  if (vec) {
    const int64_t waves = er::ceil_div(static_cast<int64_t>(batch), er::kConcatRows);
    hipLaunchKernelGGL(er::concat_cols_vec_kernel, dim3(static_cast<unsigned>(er::ceil_div(waves, er::kBlock / 64))),
                       dim3(er::kBlock), 0, er::as_stream(stream), a);
  } else {
    hipLaunchKernelGGL(er::concat_cols_kernel, dim3(er::blocks_for(static_cast<int64_t>(batch) * a.col0[n])), dim3(er::kBlock), 0,
                       er::as_stream(stream), a);
  }
  ER_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
