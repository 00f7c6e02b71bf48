// Shared helpers for libeasyrec_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "easyrec_hip.h"

namespace er {

constexpr int kWave = 64;
constexpr int kBlock = 256;

void set_error(const char* fmt, ...);

#define ER_CHECK_HIP(expr)                                                                  \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      er::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return 1;                                                                             \
    }                                                                                       \
  } while (0)

#define ER_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      er::set_error(__VA_ARGS__);  \
      return 2;                    \
    }                              \
  } while (0)

#define ER_LAUNCH_CHECK() ER_CHECK_HIP(hipGetLastError())

inline hipStream_t as_stream(er_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// fp32 -> bf16 bits, round to nearest even (NaN stays NaN): what every bf16 copy in the library holds
__host__ __device__ inline uint16_t f32_to_bf16_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}

// Cross-lane moves inside a 16-lane row by DPP (a VALU modifier: a few cycles) instead of ds_bpermute (an LDS-unit
// instruction: ~100+ cycles of latency per dependent step - a 6-step __shfl_xor butterfly measured ~0.3 us per wave sum,
// tools/micro/lib_chain.cpp).  quad_perm [1,0,3,2] / [2,3,0,1] ARE the xor-1 / xor-2 exchanges; after them the four lanes
// of a quad hold bitwise-identical sums (fp addition commutes exactly), so row_half_mirror (i <-> 7 - i) and row_mirror
// (i <-> 15 - i) deliver the same operand values as the xor-4 / xor-8 exchanges would: the butterfly's bits, unchanged.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over aligned groups of G lanes (G = 1, 2, 4, 8, 16 inside a row; 32, 64 across rows); valid in every lane of the group
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  if (G >= 2) v = v + dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
  if (G >= 4) v = v + dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
  if (G >= 8) v = v + dpp_move<0x141>(v);   // row_half_mirror
  if (G >= 16) v = v + dpp_move<0x140>(v);  // row_mirror
  if (G >= 32) v = v + __shfl_xor(v, 16, 64);
  if (G >= 64) v = v + __shfl_xor(v, 32, 64);
  return v;
}
// the same for a power-of-two group size known only at run time (wave-uniform)
__device__ __forceinline__ float group_sum_rt(float v, int G) {
  switch (G) {
    case 64: return group_sum<64>(v);
    case 32: return group_sum<32>(v);
    case 16: return group_sum<16>(v);
    case 8: return group_sum<8>(v);
    case 4: return group_sum<4>(v);
    case 2: return group_sum<2>(v);
    default: return v;
  }
}
// sum over the 64 lanes of a wave; result valid in every lane.  Same combination tree as the xor butterfly 1, 2, 4, 8, 16,
// 32 - NOT the order 32, 16, .., 1 this function used before round 4: a sum may differ in its last bit from round 3's.
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// block-wide sum for kBlock threads (4 waves); result valid in thread 0 only.
// Fixed combination order -> deterministic.
__device__ __forceinline__ float block_sum_256(float v, float* smem4) {
  v = wave_sum(v);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) smem4[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = (smem4[0] + smem4[1]) + (smem4[2] + smem4[3]);
  __syncthreads();
  return r;
}

}  // namespace er
