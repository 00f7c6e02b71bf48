// Times the LIBRARY's er_gemm_f32 (through the C ABI, no torch) on one shape: the companion of gemm_core.hip.
//   usage: lib_gemm <layout 0 NN | 1 NT | 2 TN> M N K [accumulate] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "easyrec_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int layout = argc > 1 ? atoi(argv[1]) : 1;
  const int M = argc > 2 ? atoi(argv[2]) : 8192, N = argc > 3 ? atoi(argv[3]) : 1152, K = argc > 4 ? atoi(argv[4]) : 256;
  const int acc = argc > 5 ? atoi(argv[5]) : 0, iters = argc > 6 ? atoi(argv[6]) : 30;
  // NN: A[M,K] B[K,N]; NT: A[M,K] B[N,K]; TN: A[K,M] B[K,N]
  const size_t na = static_cast<size_t>(M) * K, nb = static_cast<size_t>(N) * K, nc = static_cast<size_t>(M) * N;
  const int lda = layout == 2 ? M : K, ldb = layout == 1 ? K : N;
  float *A, *B, *C;
  CK(hipMalloc(&A, na * 4)); CK(hipMalloc(&B, nb * 4)); CK(hipMalloc(&C, nc * 4));
  std::vector<float> h(na > nb ? na : nb);
  for (size_t i = 0; i < h.size(); ++i) h[i] = static_cast<float>((i * 2654435761u >> 18) & 255) / 256.f - 0.5f;
  CK(hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), nb * 4, hipMemcpyHostToDevice));
  CK(hipMemset(C, 0, nc * 4));
  if (er_gemm_reserve(1 << 24)) { printf("reserve failed\n"); return 1; }
  auto launch = [&]() {
    if (er_gemm_f32(layout, M, N, K, A, lda, B, ldb, C, N, nullptr, acc, nullptr, nullptr)) { printf("er_gemm_f32: %s\n", er_last_error()); exit(1); }
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  printf("library layout %d M %6d N %5d K %5d acc %d : %8.1f us  %6.1f TF/s\n", layout, M, N, K, acc, us, 2.0 * M * N * K / us / 1e6);
  return 0;
}
