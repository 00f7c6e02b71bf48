// Micro-benchmark (standalone, no torch): what bounds a batch-long TN contraction dW[128 x 128] = A[K x 128]^T . B[K x 128]
// split over NB workgroups of 256 threads?  Variants of the natural-layout kernel (er_gemm.hip gemm_f32_tnn_block<2, 2>,
// interior tiles only):
//   0 stream      : the kernel's global loads only (16 rows x 512 B per operand and stage), summed in registers
//   1 stream+lds  : + the LDS stores and the barrier per stage
//   2 full        : + fragment reads and the 32 MFMAs per stage (two register sets: loads two stages ahead)
//   3 full/deep   : four register sets (loads four stages ahead)
//   4 half-rows   : loads only - a 64-column tile's pattern: 256 B of every 512-B row (what the 64 x 64 kernel reads)
//   5 full, 32-row stages (half as many barriers, 8 loads per thread in flight per set)
// usage: tn_stream <variant> <blocks> [K = 204800] [iters = 20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int LD = 160;   // LDS row stride (floats)
constexpr int W = 128;    // operand width

template <int ROWS>
__device__ __forceinline__ void fetch(const float* __restrict__ A, const float* __restrict__ B, int row0, int tid, f32x4v* ra,
                                      f32x4v* rb) {
  // ROWS x 128 floats per operand: ROWS * 32 units of 16 B; 256 threads -> ROWS / 8 units each
#pragma unroll
  for (int i = 0; i < ROWS / 8; ++i) {
    const int u = tid + i * 256;
    const int r = u >> 5, c4 = (u & 31) * 4;
    ra[i] = *reinterpret_cast<const f32x4v*>(A + static_cast<size_t>(row0 + r) * W + c4);
    rb[i] = *reinterpret_cast<const f32x4v*>(B + static_cast<size_t>(row0 + r) * W + c4);
  }
}

template <int ROWS>
__device__ __forceinline__ void stage(float* __restrict__ lds, int tid, const f32x4v* ra, const f32x4v* rb) {
#pragma unroll
  for (int i = 0; i < ROWS / 8; ++i) {
    const int u = tid + i * 256;
    const int r = u >> 5, c4 = (u & 31) * 4;
    *reinterpret_cast<f32x4v*>(&lds[r * LD + ((r >> 3) & 1) * 32 + c4]) = ra[i];
    *reinterpret_cast<f32x4v*>(&lds[ROWS * LD + 64 + r * LD + ((r >> 3) & 1) * 32 + c4]) = rb[i];
  }
}

template <int ROWS>
__device__ __forceinline__ void compute(const float* __restrict__ lds, int lane, int wm, int wn, f32x16 (&acc)[2][2]) {
  const int khalf = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int g = 0; g < ROWS / 16; ++g) {
    const float* As = lds + (g * 16 + khalf * 8) * LD + khalf * 32 + wm * 64 + l31;
    const float* Bs = lds + ROWS * LD + 64 + (g * 16 + khalf * 8) * LD + khalf * 32 + wn * 64 + l31;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a0 = As[j * LD], a1 = As[j * LD + 32], b0 = Bs[j * LD], b1 = Bs[j * LD + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
}

template <int ROWS, int SETS, bool LDS, bool MFMA>
__global__ void __launch_bounds__(256) tn_kernel(const float* __restrict__ A, const float* __restrict__ B, int rows_per_block,
                                                  float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (2 * ROWS * LD + 128)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int row_beg = blockIdx.x * rows_per_block;
  const int S = rows_per_block / ROWS;
  f32x4v ra[SETS][ROWS / 8], rb[SETS][ROWS / 8];
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  f32x4v sum = {0.f, 0.f, 0.f, 0.f};
  auto row_of = [&](int s) { return row_beg + (s < S ? s : S - 1) * ROWS; };
#pragma unroll
  for (int p = 0; p < SETS; ++p) fetch<ROWS>(A, B, row_of(p), tid, ra[p], rb[p]);
  if (LDS) {
    stage<ROWS>(lds, tid, ra[0], rb[0]);
    __syncthreads();
  }
  for (int s0 = 0; s0 < S; s0 += SETS) {
#pragma unroll
    for (int p = 0; p < SETS; ++p) {
      const int s = s0 + p;
      const int buf = s & 1;
      if (LDS) {
        // set p was staged for stage s; refill it with stage s + SETS, contract s, stage s + 1 from set (p + 1) % SETS
        fetch<ROWS>(A, B, row_of(s + SETS), tid, ra[p], rb[p]);
        __builtin_amdgcn_sched_barrier(0);
        if (MFMA) compute<ROWS>(lds + buf * (2 * ROWS * LD + 128), lane, wm, wn, acc);
        else sum[0] += lds[buf * (2 * ROWS * LD + 128) + tid];
        __builtin_amdgcn_sched_barrier(0);
        stage<ROWS>(lds + (buf ^ 1) * (2 * ROWS * LD + 128), tid, ra[(p + 1) % SETS], rb[(p + 1) % SETS]);
        __syncthreads();
      } else {
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) sum += ra[p][i] + rb[p][i];
        fetch<ROWS>(A, B, row_of(s + SETS), tid, ra[p], rb[p]);
      }
    }
  }
  float r = sum[0] + sum[1] + sum[2] + sum[3];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) r += acc[a][b][i];
  out[blockIdx.x * 256 + tid] = r;
}

// a 64-column tile's loads: 256 B of each 512-B row, 32 rows per step, two blocks (column halves) per row range
__global__ void __launch_bounds__(256) half_rows_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         int rows_per_block, float* __restrict__ out) {
  const int tid = threadIdx.x;
  const int half = blockIdx.x & 1, row_beg = (blockIdx.x >> 1) * rows_per_block;
  f32x4v sum = {0.f, 0.f, 0.f, 0.f};
  f32x4v ra[2][2], rb[2][2];
  auto fetch2 = [&](int s, f32x4v* a, f32x4v* b) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (tid & 15) + 16 * i, c4 = (tid >> 4) * 4 + half * 64;
      a[i] = *reinterpret_cast<const f32x4v*>(A + static_cast<size_t>(row_beg + s * 32 + r) * W + c4);
      b[i] = *reinterpret_cast<const f32x4v*>(B + static_cast<size_t>(row_beg + s * 32 + r) * W + c4);
    }
  };
  const int S = rows_per_block / 32;
  fetch2(0, ra[0], rb[0]);
  fetch2(S > 1 ? 1 : 0, ra[1], rb[1]);
  for (int s = 0; s < S; s += 2) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int i = 0; i < 2; ++i) sum += ra[p][i] + rb[p][i];
      const int n = s + p + 2;
      fetch2(n < S ? n : S - 1, ra[p], rb[p]);
    }
  }
  out[blockIdx.x * 256 + tid] = sum[0] + sum[1] + sum[2] + sum[3];
}

int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 2;
  const int blocks = argc > 2 ? atoi(argv[2]) : 100;
  const int K = argc > 3 ? atoi(argv[3]) : 204800;
  const int iters = argc > 4 ? atoi(argv[4]) : 20;
  float *A, *B, *out;
  const size_t n = static_cast<size_t>(K) * W;
  CK(hipMalloc(&A, n * 4));
  CK(hipMalloc(&B, n * 4));
  CK(hipMalloc(&out, 4096 * 256 * 4));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = static_cast<float>((i * 2654435761u >> 20) & 255) / 256.f - 0.5f;
  CK(hipMemcpy(A, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), n * 4, hipMemcpyHostToDevice));
  int rpb = (K / blocks) / 64 * 64;
  if (rpb < 64) rpb = 64;
  const int nb = K / rpb;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto launch = [&]() {
    switch (variant) {
      case 0: hipLaunchKernelGGL((tn_kernel<16, 2, false, false>), dim3(nb), dim3(256), 0, 0, A, B, rpb, out); break;
      case 1: hipLaunchKernelGGL((tn_kernel<16, 2, true, false>), dim3(nb), dim3(256), 0, 0, A, B, rpb, out); break;
      case 2: hipLaunchKernelGGL((tn_kernel<16, 2, true, true>), dim3(nb), dim3(256), 0, 0, A, B, rpb, out); break;
      case 3: hipLaunchKernelGGL((tn_kernel<16, 4, true, true>), dim3(nb), dim3(256), 0, 0, A, B, rpb, out); break;
      case 4: hipLaunchKernelGGL(half_rows_kernel, dim3(2 * nb), dim3(256), 0, 0, A, B, rpb, out); break;
      case 5: hipLaunchKernelGGL((tn_kernel<32, 2, true, true>), dim3(nb), dim3(256), 0, 0, A, B, rpb, out); break;
      case 6: hipLaunchKernelGGL((tn_kernel<16, 4, false, false>), dim3(nb), dim3(256), 0, 0, A, B, rpb, out); break;
      case 7: hipLaunchKernelGGL((tn_kernel<32, 4, false, false>), dim3(nb), dim3(256), 0, 0, A, B, rpb, out); break;
      default: printf("unknown variant\n"); exit(1);
    }
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  const double bytes = 2.0 * nb * rpb * W * 4 * (variant == 4 ? 1 : 1);
  printf("variant %d blocks %4d rows/block %5d : %8.1f us  %5.2f TB/s  (%.1f TF/s if contracted)\n", variant, variant == 4 ? 2 * nb : nb,
         rpb, us, bytes / us / 1e6, 2.0 * nb * rpb * W * W / us / 1e6);
  return 0;
}
