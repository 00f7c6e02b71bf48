// One element of an embedding group's finished output gradient (er_group_grad_finish, include/easyrec_hip.h):
//   base (what the consumers' GEMMs deposited) + deferred terms (row-sum broadcast, FM) + lambda * out
// - d/d(out) of the wide logit's row sum (model/deepfm.py:62-63), of FM (layers/fm.py:20-26) and of the embedding-output
// L2 (layers/input_layer.py:369-375).  Shared by the elementwise kernel (er_interaction.hip) and by the fused embedding
// backward (er_embedding.hip), which evaluates it while gathering and never materialises the finished buffer.
#pragma once
#include "er_common.h"

namespace er {

__device__ __forceinline__ float grad_finish_value(const er_grad_group& g, int64_t b, int c) {
  float v = g.has_base ? g.dout[b * g.ld + c] : 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t >= g.n_terms) break;
    const er_grad_term& q = g.terms[t];
    if (c < q.col0 || c >= q.col0 + q.width) continue;
    if (q.kind == ER_GRAD_TERM_ROWSUM) {
      v = v + q.g[b * q.g_ld];
    } else {  // ER_GRAD_TERM_FM
      const int d = (c - q.col0) % q.dim;
      v = v + q.g[b * q.g_ld + d] * (q.saved[b * q.dim + d] - g.out[b * g.ld + c]);
    }
  }
  if (g.lambda != 0.f) v = v + g.lambda * g.out[b * g.ld + c];
  return v;
}

}  // namespace er
