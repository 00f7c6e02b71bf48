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

// sum over the 64 lanes of a wave; result valid in every lane
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
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
