// Micro-benchmark (standalone, no torch): one library entry point called N times in a captured hipGraph chain, replayed R
// times: us per launch with the kernel's code and operands warm - what a kernel costs by itself, to set against what the
// step's profile shows for it (cold operands, cold instructions, neighbours).  Links libeasyrec_hip.so through its C ABI.
// usage: lib_chain <op> [chain = 32] [replays = 100]      op: head | tail | ce | bn_apply | bn_bwd | finish
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "easyrec_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define ER(x) do { if ((x) != 0) { printf("library error: %s (line %d)\n", er_last_error(), __LINE__); exit(1); } } while (0)

static float* dev_rand(size_t n, float scale = 1.f, float shift = 0.f) {
  std::vector<float> h(n);
  unsigned s = 12345u + static_cast<unsigned>(n);
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = shift + scale * (static_cast<float>(s >> 8) / 8388608.f - 1.f); }
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const std::string op = argc > 1 ? argv[1] : "head";
  const int chain = argc > 2 ? atoi(argv[2]) : 32;
  const int replays = argc > 3 ? atoi(argv[3]) : 100;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int B = 4096, K = 64, T = (B + 63) / 64;
  float *x = dev_rand(size_t(B) * K), *z = dev_rand(size_t(B) * K), *w = dev_rand(K, 0.3f), *b = dev_rand(1), *y = dev_rand(B, 0.5f, 0.5f);
  float *mean = dev_rand(K, 0.1f), *invstd = dev_rand(K, 0.25f, 1.f);
  float *logits = dev_rand(B), *probs = dev_rand(B), *dz = dev_rand(B), *dx = dev_rand(size_t(B) * K), *lp = dev_rand(T), *wb = dev_rand(size_t(T) * (K + 1));
  float *bnp = dev_rand(size_t(T) * K * 2), *loss = dev_rand(1), *reg = dev_rand(1), *total = dev_rand(1), *wg = dev_rand(K + 1), *rep = dev_rand(1);
  float *embp = dev_rand(2496), *densep = dev_rand(1100);
  std::function<void()> body;
  if (op == "head") {
    body = [&] { ER(er_head_sigmoid_ce(x, K, w, b, y, B, K, 1.f, logits, probs, dz, dx, lp, wb, z, K, mean, invstd, ER_ACT_RELU, bnp, st)); };
  } else if (op == "head_nosrc") {
    body = [&] { ER(er_head_sigmoid_ce(x, K, w, b, y, B, K, 1.f, logits, probs, dz, dx, lp, wb, nullptr, 0, nullptr, nullptr, 0, nullptr, st)); };
  } else if (op == "tail") {
    const float* losses[1] = {lp};
    float* report[1] = {rep};
    float* values[1] = {loss};
    int32_t parts[1] = {T};
    float scales[1] = {1.f}, divs[1] = {float(B)};
    er_tail_job jobs[2] = {{wb, wg, T, K, K + 1}, {wb + K, wg + K, T, 1, K + 1}};
    body = [&, losses, report, values, parts, scales, divs, jobs] {
      ER(er_loss_tail(embp, 2496, 5e-6f, densep, 1100, losses, report, parts, scales, divs, values, 1, jobs, 2, reg, total, st)); };
  } else if (op == "ce") {
    body = [&] { ER(er_sigmoid_ce_fwd_bwd(logits, y, nullptr, B, 1.f, loss, dz, probs, st)); };
  } else {
    printf("unknown op %s\n", op.c_str());
    return 1;
  }
  for (int i = 0; i < 3; ++i) body();
  CK(hipStreamSynchronize(st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < chain; ++i) body();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const char* dbg = getenv("ER_HEAD_DEBUG");
  printf("%-12s ER_HEAD_DEBUG=%s  %7.3f us/launch (chain %d x %d replays)\n", op.c_str(), dbg ? dbg : "-", ms * 1e3 / (chain * replays), chain, replays);
  return 0;
}
