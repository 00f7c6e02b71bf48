// Microbenchmarks behind DESIGN.md's GEMM analysis: what one wave per SIMD can get out of the f32 matrix core.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_chain(float* out, int iters, float a, float b) {
  f32x16 acc = {0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  float s = 0; for (int j = 0; j < 16; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_two(float* out, int iters, float a, float b) {
  f32x16 acc = {0}, acc2 = {0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc2, 0, 0, 0);
    }
  }
  float s = 0; for (int j = 0; j < 16; ++j) s += acc[j] + acc2[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// LDS fragments + MFMA + barrier per k-tile, no global traffic
__global__ void __launch_bounds__(256) k_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * 64 * 36];
  for (int i = threadIdx.x; i < 2 * 2 * 64 * 36; i += 256) lds[i] = 0.001f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fa = ((wave >> 1) * 32 + (lane & 31)) * 36 + (lane >> 5) * 16;
  const int fb = 64 * 36 + ((wave & 1) * 32 + (lane & 31)) * 36 + (lane >> 5) * 16;
  f32x16 acc = {0};
  for (int i = 0; i < iters; ++i) {
    const float* base = lds + (i & 1) * 2 * 64 * 36;
    f32x4v a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = *reinterpret_cast<const f32x4v*>(base + fa + 4 * q);
      b[q] = *reinterpret_cast<const f32x4v*>(base + fb + 4 * q);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][j], b[q][j], acc, 0, 0, 0);
    __syncthreads();
  }
  float s = 0; for (int j = 0; j < 16; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// streaming read: every block reads `bytes_per_block` with 16-byte loads, `inflight` independent loads per thread
template <int INFLIGHT>
__global__ void __launch_bounds__(256) k_stream(const f32x4v* __restrict__ in, float* out, int n_per_thread) {
  const f32x4v* p = in + static_cast<size_t>(blockIdx.x) * 256 * n_per_thread + threadIdx.x;
  f32x4v s = {0, 0, 0, 0};
  for (int i = 0; i < n_per_thread; i += INFLIGHT) {
    f32x4v v[INFLIGHT];
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) v[j] = p[(i + j) * 256];
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) s += v[j];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <typename F> float time_us(F f, int reps = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f / reps;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 256;  // x16 MFMAs
  for (int blocks : {256, 512, 1024}) {
    float t1 = time_us([&] { hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    float t2 = time_us([&] { hipLaunchKernelGGL(k_two, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    float t3 = time_us([&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, out, iters); });
    const double fl = 2.0 * 32 * 32 * 2 * 16.0 * iters * 4 * blocks;
    printf("blocks %4d: chain %7.1f us %6.1f TF/s (%.1f clk/MFMA @2.4GHz) | two-acc %7.1f us %6.1f TF/s | lds+mfma+barrier %7.1f us %6.1f TF/s\n",
           blocks, t1, fl / t1 / 1e6, t1 * 2400.0 / (16.0 * iters * ((blocks + 255) / 256)), t2, fl / t2 / 1e6, t3, fl / t3 / 1e6);
  }
  // short kernels: fixed cost
  for (int it : {1, 4, 8, 20}) {
    float t3 = time_us([&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 0, 0, out, it); }, 200);
    printf("k_lds iters %2d (256 blocks): %6.2f us per launch (back-to-back)\n", it, t3);
  }
  f32x4v* in; size_t bytes = size_t(256) << 20; hipMalloc(&in, bytes); hipMemset(in, 0, bytes);
  for (int mb : {10, 40, 256}) {
    const int blocks = 256 * 4;
    const int n_per_thread = int((size_t(mb) << 20) / 16 / 256 / blocks);
    float a = time_us([&] { hipLaunchKernelGGL(k_stream<2>, dim3(blocks), dim3(256), 0, 0, in, out, n_per_thread); });
    float b = time_us([&] { hipLaunchKernelGGL(k_stream<8>, dim3(blocks), dim3(256), 0, 0, in, out, n_per_thread); });
    const double by = double(n_per_thread) * 16 * 256 * blocks;
    printf("stream %3d MB: 2 in flight %7.1f us %6.2f TB/s | 8 in flight %7.1f us %6.2f TB/s\n", mb, a, by / a / 1e6, b, by / b / 1e6);
  }
  return 0;
}
