// FM second-order interaction and the wide row sum as device bodies (reference layers/fm.py:20-26, model/deepfm.py:62-63):
// shared by er_interaction.hip's kernels (fm_fwd_kernel, rowsum_fwd_kernel, wide_fm_concat_kernel) and by the launch that runs
// them beside a BatchNorm finalize + apply (er_dense.hip: bn_apply_wide_fm_kernel).
#pragma once
#include "er_common.h"

namespace er {

template <int V>
__device__ __forceinline__ void fm_fwd_body(int64_t idx, const float* __restrict__ x, int B, int F, int D, int x_stride,
                                            float* __restrict__ fm_out, int fm_stride, float* __restrict__ sum_out) {
  const int lanes = D / V;
  const int64_t b = idx / lanes;
  const int c = static_cast<int>(idx % lanes) * V;
  if (b >= B) return;
  const float* row = x + b * x_stride + c;
  float s[V], q[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { s[i] = 0.f; q[i] = 0.f; }
  // 16 fields per trip (a field past the end reads the last one and is skipped): the kernel has one wave per compute unit
  // at B = 4096, so what it costs is the number of dependent round trips - 3 for DeepFM's 39 fields instead of 10
  constexpr int kFields = 16;
  for (int f0 = 0; f0 < F; f0 += kFields) {
    float e[kFields][V];
#pragma unroll
    for (int u = 0; u < kFields; ++u) {
      const int f = f0 + u < F ? f0 + u : F - 1;
      if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(row + f * D);
        e[u][0] = t.x; e[u][1] = t.y; e[u][2] = t.z; e[u][3] = t.w;
      } else {
        e[u][0] = row[f * D];
      }
    }
#pragma unroll
    for (int u = 0; u < kFields; ++u) {
      if (f0 + u < F) {
#pragma unroll
        for (int i = 0; i < V; ++i) { s[i] = s[i] + e[u][i]; q[i] = q[i] + e[u][i] * e[u][i]; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    fm_out[b * fm_stride + c + i] = 0.5f * (s[i] * s[i] - q[i]);
    sum_out[b * D + c + i] = s[i];
  }
}

__device__ __forceinline__ void rowsum_fwd_body(int64_t idx, const float* __restrict__ x, int B, int n, int x_stride,
                                                float* __restrict__ out, int out_stride) {
  // 4 lanes per row, then a 4-lane tree: keeps loads semi-coalesced for n ~ 39
  const int64_t b = idx >> 2;
  const int sub = static_cast<int>(idx & 3);
  float s = 0.f;
  if (b < B) {
    // 8 of the lane's columns per trip (past the end: the last column, skipped), added in the same order
    for (int j0 = sub; j0 < n; j0 += 32) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = x[b * x_stride + (j0 + 4 * u < n ? j0 + 4 * u : n - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j0 + 4 * u < n) s = s + t[u];
    }
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if (b < B && sub == 0) out[b * out_stride] = s;
}

}  // namespace er
