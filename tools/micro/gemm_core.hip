// Micro-benchmark (standalone): the core of the f32 MFMA GEMM on K-contiguous operands, C[M x N] = A[M x K] . B[N x K]^T
// (the NT input-gradient layout; NN's A and TN differ only in how a tile reaches LDS), interior tiles only, exact grids.
// What does the block tile / k-tile / prefetch depth / instruction scheduling buy on MI355X?
//   usage: gemm_core <variant> M N K [iters]
//   0: 64 x 64 tile, BK 32, two register sets, sched barriers  (= gemm_f32_block)
//   1: 64 x 64, no sched barriers      2: 128 x 128, BK 32        3: 128 x 64, BK 32       4: 64 x 64, BK 64
//   5: 128 x 128, BK 16                6: 128 x 128, BK 32, sched barriers                  7: 128 x 64 sched barriers
//   8: 1 + XCD-aware tile order        9: 0 with every fragment read up front (library)     10: 1 + C (+)=    11: 9 + 8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int TM, int TN, int BK, bool SCHED, bool REMAP = false, bool FRONT = false, bool ACC = false>
__global__ void __launch_bounds__(256) gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                    int M, int N, int K) {
  constexpr int SK = BK + 4;
  constexpr int WM = TM / 64, WN = TN / 64;          // 32 x 32 accumulators per wave
  constexpr int UA = TM * BK / 4 / 256, UB = TN * BK / 4 / 256;  // 16-byte units per thread and stage
  constexpr int kStage = (TM + TN) * SK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int gx = N / TN;
  int tile = blockIdx.x;
  if (REMAP) {  // the library's XCD-aware order: XCD c = workgroup % 8 owns tiles [c * per, (c + 1) * per)
    const int nt = gx * (M / TM), per = nt / 8;
    if (tile < 8 * per) tile = (tile % 8) * per + tile / 8;
  }
  const int tx = tile % gx, ty = tile / gx;
  const int m0 = ty * TM, n0 = tx * TN;
  const int T = K / BK;
  f32x16 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  f32x4v ra[2][UA], rb[2][UB];
  auto fetch = [&](int set, int t) {
    t = t < T ? t : T - 1;
#pragma unroll
    for (int i = 0; i < UA; ++i) {
      const int u = tid + i * 256;
      ra[set][i] = *reinterpret_cast<const f32x4v*>(A + static_cast<size_t>(m0 + u / (BK / 4)) * K + t * BK + (u % (BK / 4)) * 4);
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int u = tid + i * 256;
      rb[set][i] = *reinterpret_cast<const f32x4v*>(B + static_cast<size_t>(n0 + u / (BK / 4)) * K + t * BK + (u % (BK / 4)) * 4);
    }
  };
  auto stage = [&](int buf, int set) {
    float* As = lds + buf * kStage;
    float* Bs = As + TM * SK;
#pragma unroll
    for (int i = 0; i < UA; ++i) {
      const int u = tid + i * 256;
      *reinterpret_cast<f32x4v*>(&As[(u / (BK / 4)) * SK + (u % (BK / 4)) * 4]) = ra[set][i];
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int u = tid + i * 256;
      *reinterpret_cast<f32x4v*>(&Bs[(u / (BK / 4)) * SK + (u % (BK / 4)) * 4]) = rb[set][i];
    }
  };
  const int khalf = lane >> 5, l31 = lane & 31;
  auto compute = [&](int buf, int q0, int q1) {
    const float* As = lds + buf * kStage + (wm * 32 * WM + l31) * SK + khalf * (BK / 2);
    const float* Bs = lds + buf * kStage + TM * SK + (wn * 32 * WN + l31) * SK + khalf * (BK / 2);
#pragma unroll
    for (int q = q0; q < q1; ++q) {
      f32x4v a[WM], b[WN];
#pragma unroll
      for (int h = 0; h < WM; ++h) a[h] = *reinterpret_cast<const f32x4v*>(As + h * 32 * SK + 4 * q);
#pragma unroll
      for (int c = 0; c < WN; ++c) b[c] = *reinterpret_cast<const f32x4v*>(Bs + c * 32 * SK + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < WM; ++h)
#pragma unroll
          for (int c = 0; c < WN; ++c) acc[h][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[h][i], b[c][i], acc[h][c], 0, 0, 0);
    }
  };
  constexpr int Q = BK / 8;
  auto compute_front = [&](int buf, f32x4v (&a)[WM][BK / 8], f32x4v (&b)[WN][BK / 8], int q0, int q1) {
#pragma unroll
    for (int q = q0; q < q1; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < WM; ++h)
#pragma unroll
          for (int c = 0; c < WN; ++c) acc[h][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[h][q][i], b[c][q][i], acc[h][c], 0, 0, 0);
  };
  fetch(0, 0);
  fetch(1, 1);
  stage(0, 0);
  __syncthreads();
  for (int t = 0; t < T; t += 2) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      // set p was staged for k-tile t + p: refill it with k-tile t + p + 2, contract t + p, stage t + p + 1 from the other set
      if (FRONT) {  // the library's order: every fragment of the k-tile is read before its global loads are issued
        f32x4v fa[WM][BK / 8], fb[WN][BK / 8];
        const float* As = lds + p * kStage + (wm * 32 * WM + l31) * SK + khalf * (BK / 2);
        const float* Bs = lds + p * kStage + TM * SK + (wn * 32 * WN + l31) * SK + khalf * (BK / 2);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
#pragma unroll
          for (int h = 0; h < WM; ++h) fa[h][q] = *reinterpret_cast<const f32x4v*>(As + h * 32 * SK + 4 * q);
#pragma unroll
          for (int c = 0; c < WN; ++c) fb[c][q] = *reinterpret_cast<const f32x4v*>(Bs + c * 32 * SK + 4 * q);
        }
        fetch(p, t + p + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute_front(p, fa, fb, 0, Q - 1);
        __builtin_amdgcn_sched_barrier(0);
        stage(p ^ 1, p ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        compute_front(p, fa, fb, Q - 1, Q);
        __syncthreads();
        continue;
      }
      fetch(p, t + p + 2);
      if (SCHED) __builtin_amdgcn_sched_barrier(0);
      compute(p, 0, Q - 1 > 0 ? Q - 1 : Q);
      if (SCHED) __builtin_amdgcn_sched_barrier(0);
      stage(p ^ 1, p ^ 1);
      if (SCHED) __builtin_amdgcn_sched_barrier(0);
      if (Q - 1 > 0) compute(p, Q - 1, Q);
      __syncthreads();
    }
  }
#pragma unroll
  for (int h = 0; h < WM; ++h)
#pragma unroll
    for (int c = 0; c < WN; ++c) {
      const int col = n0 + wn * 32 * WN + c * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 * WM + h * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        float* pc = C + static_cast<size_t>(row) * N + col;
        *pc = ACC ? *pc + acc[h][c][r] : acc[h][c][r];
      }
    }
}

template <int TM, int TN, int BK, bool SCHED, bool REMAP = false, bool FRONT = false, bool ACC = false>
void run(const float* A, const float* B, float* C, int M, int N, int K) {
  const int lds_bytes = 2 * (TM + TN) * (BK + 4) * 4;
  static bool once = false;
  if (!once) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<TM, TN, BK, SCHED, REMAP, FRONT, ACC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           lds_bytes));
    once = true;
  }
  hipLaunchKernelGGL((gemm_kernel<TM, TN, BK, SCHED, REMAP, FRONT, ACC>), dim3((M / TM) * (N / TN)), dim3(256), lds_bytes, 0, A, B, C, M, N, K);
}

int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0;
  const int M = argc > 2 ? atoi(argv[2]) : 8192, N = argc > 3 ? atoi(argv[3]) : 1152, K = argc > 4 ? atoi(argv[4]) : 256;
  const int iters = argc > 5 ? atoi(argv[5]) : 30;
  if (M % 128 || N % 128 || K % 128) { printf("M, N, K must be multiples of 128\n"); return 1; }
  float *A, *B, *C;
  CK(hipMalloc(&A, static_cast<size_t>(M) * K * 4));
  CK(hipMalloc(&B, static_cast<size_t>(N) * K * 4));
  CK(hipMalloc(&C, static_cast<size_t>(M) * N * 4));
  std::vector<float> ha(static_cast<size_t>(M) * K), hb(static_cast<size_t>(N) * K);
  for (size_t i = 0; i < ha.size(); ++i) ha[i] = static_cast<float>((i * 2654435761u >> 18) & 255) / 256.f - 0.5f;
  for (size_t i = 0; i < hb.size(); ++i) hb[i] = static_cast<float>((i * 40503u >> 7) & 255) / 256.f - 0.5f;
  CK(hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  auto launch = [&]() {
    switch (variant) {
      case 0: run<64, 64, 32, true>(A, B, C, M, N, K); break;
      case 1: run<64, 64, 32, false>(A, B, C, M, N, K); break;
      case 2: run<128, 128, 32, false>(A, B, C, M, N, K); break;
      case 3: run<128, 64, 32, false>(A, B, C, M, N, K); break;
      case 4: run<64, 64, 64, false>(A, B, C, M, N, K); break;
      case 5: run<128, 128, 16, false>(A, B, C, M, N, K); break;
      case 6: run<128, 128, 32, true>(A, B, C, M, N, K); break;
      case 7: run<128, 64, 32, true>(A, B, C, M, N, K); break;
      case 8: run<64, 64, 32, false, true>(A, B, C, M, N, K); break;               // + XCD-aware tile order
      case 9: run<64, 64, 32, true, false, true>(A, B, C, M, N, K); break;         // fragments up front (library order)
      case 10: run<64, 64, 32, false, false, false, true>(A, B, C, M, N, K); break;  // + C (+)=
      case 11: run<64, 64, 32, true, true, true>(A, B, C, M, N, K); break;         // library order + XCD-aware tiles
      default: printf("unknown variant\n"); exit(1);
    }
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  // spot check of 64 entries against the host
  std::vector<float> hc(static_cast<size_t>(M) * N);
  CK(hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int s = 0; s < 64; ++s) {
    const int r = (s * 977 + 13) % M, c = (s * 331 + 7) % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += static_cast<double>(ha[static_cast<size_t>(r) * K + k]) * hb[static_cast<size_t>(c) * K + k];
    const double e = ref - hc[static_cast<size_t>(r) * N + c];
    if ((e < 0 ? -e : e) > maxerr) maxerr = e < 0 ? -e : e;
  }
  printf("variant %d  M %6d N %5d K %5d : %8.1f us  %6.1f TF/s  (max err %.2e)\n", variant, M, N, K, us, 2.0 * M * N * K / us / 1e6, maxerr);
  return 0;
}
