// Micro-benchmark (standalone, no torch): what does ONE small dependent kernel cost inside a replayed hipGraph, and which
// ingredient of the step's small kernels (VERDICT r3 "What's weak": sigmoid_ce 8.2 us, hyper_select 5.4 us, ... against
// the guide's 1.45 us boundary) pays for it?  A chain of N launches of one variant on one stream is captured, replayed
// R times and timed with HIP events: us per launch = total / (N * R).
//   0 empty, 1 x 64          1 empty, 256 x 256            2 empty, 1024 x 256
//   3 1 KB by-value struct argument, one uniform field read, 256 x 256
//   4 3.5 KB by-value struct argument (RunMulti-sized), one uniform field read, 832 x 256
//   5 two dependent loads (descriptor table -> pointer -> value), 416 x 256
//   6 four dependent loads (sorted key -> permutation -> pointer -> row), 416 x 256  (the tile kernel's gather chain)
//   7 one-workgroup reduction over 4096 floats, two dependent scalar loads in front (sigmoid_ce-like)
//   8 the same reduction by 16 workgroups + last-arriver combine through one atomic counter
//   9 streaming 4 MB read + 4 MB write, 1024 x 256, 16-byte lanes (bn_apply-sized)
//  10 streaming 4 MB read + 4 MB write behind a 64-partial merge per workgroup (bn_finalize_apply-like: 1024 workgroups
//     each re-reading 48 KB of L2-resident partials first)
//  11 20 MB read-modify-write, 2048 x 256 (group_grad_finish-sized)
//  12 empty kernel launched with a 64 KB dynamic LDS request, 832 x 256
// usage: launch_floor [chain = 48] [replays = 200]      (run once with HIP_FORCE_DEV_KERNARG=0 and once with =1)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <utility>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct Big1K { float* out; int sel; int pad[253]; };
struct Big3K { float* out; int sel; int pad[893]; };

__global__ void k_empty(float* out) { if (out == nullptr) out[0] = 1.f; }
__global__ void k_big1(Big1K a) { if (threadIdx.x == 0 && a.pad[a.sel] == 12345) a.out[blockIdx.x] = 1.f; }
__global__ void k_big3(Big3K a) { if (threadIdx.x == 0 && a.pad[a.sel] == 12345) a.out[blockIdx.x] = 1.f; }

struct Desc { const float* base; int stride; int pad; };
__global__ void k_dep2(const Desc* __restrict__ descs, int n_desc, float* __restrict__ out) {
  const Desc d = descs[blockIdx.x % n_desc];
  out[blockIdx.x * 256 + threadIdx.x] = d.base[(blockIdx.x * 256 + threadIdx.x) % d.stride];
}
__global__ void k_dep4(const unsigned* __restrict__ skeys, const unsigned* __restrict__ perm, const float* const* __restrict__ gptr,
                       float* __restrict__ out, int n) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const unsigned key = skeys[p];
  if (key == 0xFFFFFFFFu) return;
  const unsigned j = perm[p];
  const float* g = gptr[j];
  out[p] = g[key & 15];
}
__global__ void k_reduce1(const float* __restrict__ x, const int* __restrict__ n_ptr, const float* __restrict__ scale_ptr,
                          float* __restrict__ out) {
  __shared__ float red[4];
  const int n = *n_ptr;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * *scale_ptr;
}
__global__ void k_reduce16(const float* __restrict__ x, const int* __restrict__ n_ptr, float* __restrict__ partial,
                           unsigned* __restrict__ counter, float* __restrict__ out) {
  __shared__ float red[4];
  __shared__ bool last;
  const int n = *n_ptr;
  float s = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += 256 * gridDim.x) s += x[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partial[blockIdx.x], (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (t % gridDim.x) == gridDim.x - 1;
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    float v = threadIdx.x < gridDim.x ? __hip_atomic_load(&partial[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (threadIdx.x == 0) out[0] = v;
  }
}
__global__ void k_stream(const f32x4v* __restrict__ x, f32x4v* __restrict__ y, int n4) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
    f32x4v v = x[i];
    v = v * 1.0001f + 0.5f;
    y[i] = v;
  }
}
__global__ void k_merge_stream(const float* __restrict__ partial, int chunks, int N, const float* __restrict__ x,
                               float* __restrict__ y, int B) {
  // grid (N / 64, B / 16): merge `chunks` partials of 64 columns (3 floats each), then transform a 16-row tile
  __shared__ float sm[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f;
  for (int k = rl; k < chunks; k += 4) {
    const float* p = partial + (static_cast<size_t>(k) * N + c) * 3;
    a += p[0] + p[1] * 0.5f + p[2] * 0.25f;
  }
  sm[rl][cl] = a;
  __syncthreads();
  const float m = (sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]);
  for (int k = 0; k < 4; ++k) {
    const int r = blockIdx.y * 16 + rl + 4 * k;
    if (r < B) y[static_cast<size_t>(r) * N + c] = x[static_cast<size_t>(r) * N + c] * 0.999f + m;
  }
}
__global__ void k_rmw(f32x4v* __restrict__ d, const f32x4v* __restrict__ o, int n4) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) d[i] = d[i] + o[i] * 1e-5f;
}
__global__ void k_lds(float* out) {
  extern __shared__ float sm[];
  if (out == nullptr) out[0] = sm[threadIdx.x];
}

// ---- shader clock: clock64() (s_memtime: shader cycles) against wall_clock64() (100 MHz) over a dependent ALU chain
__global__ void k_clock(unsigned long long* out, int iters) {
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  float acc = static_cast<float>(threadIdx.x);
  for (int i = 0; i < iters; ++i) acc = acc * 1.0001f + 0.5f;
  const unsigned long long w1 = wall_clock64(), c1 = clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = w1 - w0; out[blockIdx.x * 4 + 1] = c1 - c0; out[blockIdx.x * 4 + 2] = static_cast<unsigned long long>(acc); }
}

// ---- loss-kernel shapes: which ingredient of a one-workgroup loss kernel (sigmoid_ce: 6-8 us) costs what
template <int MATH, int REDUCE>
__global__ void k_ce(const float* __restrict__ z, const float* __restrict__ y, int B, float* __restrict__ dz, float* __restrict__ probs,
                     float* __restrict__ loss_part) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
    const float zi = z[i], yi = y[i];
    float ce, p;
    if (MATH) { ce = fmaxf(zi, 0.f) - zi * yi + log1pf(expf(-fabsf(zi))); p = 1.f / (1.f + expf(-zi)); }
    else { ce = zi * yi; p = zi * 0.5f; }
    acc += ce;
    probs[i] = p;
    dz[i] = (p - yi) * 0.001f;
  }
  if (REDUCE) {
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float r = 0.f;
      for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) r += red[w];
      loss_part[blockIdx.x] = r;
    }
  } else if (acc == 12345.f) {
    loss_part[blockIdx.x] = acc;
  }
}

// ---- cold instruction fetch: 48 DISTINCT kernels of ~4 KB / ~16 KB of straight-line code against ONE of them 48 times
template <int ID, int N>
__global__ void k_code(float* out, int n) {
  float acc = static_cast<float>(threadIdx.x);
#pragma unroll
  for (int i = 0; i < N; ++i) acc = acc * 1.0001f + static_cast<float>((i * 7 + ID) & 1023);
  out[(blockIdx.x * 256 + threadIdx.x) * n] = acc;  // (always stored: the chain cannot be sunk into a branch)
}
template <int ID, int N>
void launch_code(hipStream_t st, float* buf) { hipLaunchKernelGGL((k_code<ID, N>), dim3(256), dim3(256), 0, st, buf, 1); }
inline void launch_rmw(hipStream_t st, f32x4v* y, const f32x4v* x, int n4) { hipLaunchKernelGGL(k_rmw, dim3(2048), dim3(256), 0, st, y, x, n4); }
template <int N, int... IDS>
void launch_code_chain(hipStream_t st, float* buf, bool distinct, std::integer_sequence<int, IDS...>) {
  if (distinct) { (launch_code<IDS, N>(st, buf), ...); }
  else { for (size_t i = 0; i < sizeof...(IDS); ++i) launch_code<0, N>(st, buf); }
}
template <int N, int... IDS>
void launch_code_chain_thrash(hipStream_t st, float* buf, f32x4v* y, const f32x4v* x, int n4, bool distinct, std::integer_sequence<int, IDS...>) {
  if (distinct) { ((launch_rmw(st, y, x, n4), launch_code<IDS, N>(st, buf)), ...); }
  else { for (size_t i = 0; i < sizeof...(IDS); ++i) { launch_rmw(st, y, x, n4); launch_code<0, N>(st, buf); } }
}

template <typename F>
float time_graph(hipStream_t st, int replays, F&& body) {
  hipGraph_t g; hipGraphExec_t ge;
  body();
  CK(hipStreamSynchronize(st));
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  body();
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
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return ms * 1e3f / replays;
}

int main(int argc, char** argv) {
  const int chain = argc > 1 ? atoi(argv[1]) : 48;
  const int replays = argc > 2 ? atoi(argv[2]) : 200;
  const char* kn = getenv("HIP_FORCE_DEV_KERNARG");
  printf("launch_floor: chain %d, replays %d, HIP_FORCE_DEV_KERNARG=%s\n", chain, replays, kn ? kn : "(unset)");
  float *buf, *x, *y, *partial;
  const int n = 4096 * 256;  // 4 MB
  CK(hipMalloc(&buf, 1 << 26));
  CK(hipMalloc(&x, 5 * 4096 * 256 * 4));
  CK(hipMalloc(&y, 5 * 4096 * 256 * 4));
  CK(hipMalloc(&partial, 64 * 256 * 3 * 4));
  CK(hipMemset(buf, 0, 1 << 26));
  CK(hipMemset(x, 0, 5 * n * 4));
  CK(hipMemset(y, 0, 5 * n * 4));
  CK(hipMemset(partial, 0, 64 * 256 * 3 * 4));
  // dependent-load inputs
  const int n_ent = 416 * 256;
  std::vector<unsigned> hk(n_ent), hp(n_ent);
  std::vector<const float*> hg(n_ent);
  unsigned s = 12345;
  for (int i = 0; i < n_ent; ++i) { hk[i] = i; s = s * 1664525u + 1013904223u; hp[i] = s % n_ent; hg[i] = x + ((s >> 8) % (n - 16)); }
  unsigned *dk, *dp; const float** dg; Desc* dd; int* dn; float* dscale; unsigned* counter;
  CK(hipMalloc(&dk, n_ent * 4)); CK(hipMalloc(&dp, n_ent * 4)); CK(hipMalloc(&dg, n_ent * 8));
  CK(hipMalloc(&dd, 39 * sizeof(Desc))); CK(hipMalloc(&dn, 4)); CK(hipMalloc(&dscale, 4)); CK(hipMalloc(&counter, 4));
  CK(hipMemcpy(dk, hk.data(), n_ent * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, hp.data(), n_ent * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, hg.data(), n_ent * 8, hipMemcpyHostToDevice));
  std::vector<Desc> hd(39);
  for (int i = 0; i < 39; ++i) hd[i] = Desc{x + i * 1024, 4096, 0};
  CK(hipMemcpy(dd, hd.data(), 39 * sizeof(Desc), hipMemcpyHostToDevice));
  const int hn = 4096; const float hs = 0.5f;
  CK(hipMemcpy(dn, &hn, 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dscale, &hs, 4, hipMemcpyHostToDevice));
  CK(hipMemset(counter, 0, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  Big1K b1; memset(&b1, 0, sizeof(b1)); b1.out = buf; b1.sel = 7;
  Big3K b3; memset(&b3, 0, sizeof(b3)); b3.out = buf; b3.sel = 7;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const char* names[] = {"empty 1x64", "empty 256x256", "empty 1024x256", "1 KB by-value arg 256x256", "3.5 KB by-value arg 832x256",
                         "2 dependent loads 416x256", "4 dependent loads 416x256", "1-workgroup reduce of 4096 (+2 scalar loads)",
                         "16-workgroup reduce + last-arriver combine", "stream 4 MB r + 4 MB w, 1024x256",
                         "64-partial merge + 4 MB r/w tile, 4x256 workgroups", "20 MB read-modify-write 2048x256",
                         "empty + 64 KB dynamic LDS 832x256"};
  for (int v = 0; v <= 12; ++v) {
    auto launch = [&]() {
      switch (v) {
        case 0: hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, buf); break;
        case 1: hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, buf); break;
        case 2: hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st, buf); break;
        case 3: hipLaunchKernelGGL(k_big1, dim3(256), dim3(256), 0, st, b1); break;
        case 4: hipLaunchKernelGGL(k_big3, dim3(832), dim3(256), 0, st, b3); break;
        case 5: hipLaunchKernelGGL(k_dep2, dim3(416), dim3(256), 0, st, dd, 39, buf); break;
        case 6: hipLaunchKernelGGL(k_dep4, dim3(416), dim3(256), 0, st, dk, dp, dg, buf, n_ent); break;
        case 7: hipLaunchKernelGGL(k_reduce1, dim3(1), dim3(256), 0, st, x, dn, dscale, buf); break;
        case 8: hipLaunchKernelGGL(k_reduce16, dim3(16), dim3(256), 0, st, x, dn, buf + 64, counter, buf); break;
        case 9: hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, st, reinterpret_cast<const f32x4v*>(x), reinterpret_cast<f32x4v*>(y), n / 4); break;
        case 10: hipLaunchKernelGGL(k_merge_stream, dim3(4, 256), dim3(256), 0, st, partial, 64, 256, x, y, 4096); break;
        case 11: hipLaunchKernelGGL(k_rmw, dim3(2048), dim3(256), 0, st, reinterpret_cast<f32x4v*>(y), reinterpret_cast<const f32x4v*>(x), 5 * n / 4); break;
        case 12: hipLaunchKernelGGL(k_lds, dim3(832), dim3(256), 65536, st, buf); break;
      }
    };
    for (int i = 0; i < 4; ++i) launch();
    CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < chain; ++i) launch();
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
    // eager for comparison (host-bound below ~3 us per launch)
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < chain * 20; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms2 = 0.f;
    CK(hipEventElapsedTime(&ms2, e0, e1));
    printf("%2d %-52s graph %7.3f us/launch   eager %7.3f us/launch\n", v, names[v], ms * 1e3 / (chain * replays), ms2 * 1e3 / (chain * 20));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  // shader clock under three kinds of load
  {
    unsigned long long* ck;
    CK(hipMalloc(&ck, 256 * 4 * 8));
    auto report = [&](const char* what) {
      hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, st, ck, 20000);
      CK(hipStreamSynchronize(st));
      unsigned long long h[4];
      CK(hipMemcpy(h, ck, sizeof(h), hipMemcpyDeviceToHost));
      printf("clock %-58s wall %6.1f us, clock64 %9llu ticks -> %7.1f MHz; %5.2f shader-clock-ticks per dependent mul+add\n", what, h[0] / 100.0, h[1],
             h[1] / (h[0] / 100.0), static_cast<double>(h[1]) / 20000);
    };
    CK(hipStreamSynchronize(st));
    report("(idle GPU)");
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, buf);
    report("(after 2000 empty kernels, ~3.4 ms)");
    for (int i = 0; i < 3000; ++i) hipLaunchKernelGGL(k_rmw, dim3(2048), dim3(256), 0, st, reinterpret_cast<f32x4v*>(y), reinterpret_cast<const f32x4v*>(x), 5 * n / 4);
    report("(after 3000 x 40 MB-traffic kernels, ~25 ms)");
    for (int i = 0; i < 20000; ++i) hipLaunchKernelGGL(k_reduce1, dim3(1), dim3(256), 0, st, x, dn, dscale, buf);
    report("(after 20000 one-workgroup kernels, ~60 ms)");
  }
  // loss-kernel shapes
  {
    struct V { const char* name; int grid, block, math, reduce; };
    const V vs[] = {{"ce 1 x 1024, no math, no reduce", 1, 1024, 0, 0}, {"ce 1 x 1024, no math, reduce", 1, 1024, 0, 1},
                    {"ce 1 x 1024, math, reduce", 1, 1024, 1, 1}, {"ce 1 x 256, math, reduce", 1, 256, 1, 1},
                    {"ce 16 x 256, math, reduce (partials)", 16, 256, 1, 1}, {"ce 64 x 64, math, reduce (partials)", 64, 64, 1, 1},
                    {"ce 16 x 256, no math, no reduce", 16, 256, 0, 0}};
    for (const V& v : vs) {
      auto one = [&] {
        if (v.math && v.reduce) hipLaunchKernelGGL((k_ce<1, 1>), dim3(v.grid), dim3(v.block), 0, st, x, x + 4096, 4096, y, y + 4096, buf);
        else if (v.reduce) hipLaunchKernelGGL((k_ce<0, 1>), dim3(v.grid), dim3(v.block), 0, st, x, x + 4096, 4096, y, y + 4096, buf);
        else hipLaunchKernelGGL((k_ce<0, 0>), dim3(v.grid), dim3(v.block), 0, st, x, x + 4096, 4096, y, y + 4096, buf);
      };
      const float us = time_graph(st, replays, [&] { for (int i = 0; i < chain; ++i) one(); });
      printf("%-44s graph %7.3f us/launch\n", v.name, us / chain);
    }
  }
  // cold instruction fetch
  {
    auto seq = std::make_integer_sequence<int, 48>{};
    f32x4v* y4 = reinterpret_cast<f32x4v*>(y); const f32x4v* x4 = reinterpret_cast<const f32x4v*>(x);
    const float same_s = time_graph(st, replays, [&] { launch_code_chain<256>(st, buf, false, seq); });
    const float dist_s = time_graph(st, replays, [&] { launch_code_chain<256>(st, buf, true, seq); });
    const float same_l = time_graph(st, replays, [&] { launch_code_chain<1024>(st, buf, false, seq); });
    const float dist_l = time_graph(st, replays, [&] { launch_code_chain<1024>(st, buf, true, seq); });
    const float same_t = time_graph(st, replays / 4, [&] { launch_code_chain_thrash<256>(st, buf, y4, x4, 5 * n / 4, false, seq); });
    const float dist_t = time_graph(st, replays / 4, [&] { launch_code_chain_thrash<256>(st, buf, y4, x4, 5 * n / 4, true, seq); });
    const float same_tl = time_graph(st, replays / 4, [&] { launch_code_chain_thrash<1024>(st, buf, y4, x4, 5 * n / 4, false, seq); });
    const float dist_tl = time_graph(st, replays / 4, [&] { launch_code_chain_thrash<1024>(st, buf, y4, x4, 5 * n / 4, true, seq); });
    printf("code  ~4 KB straight-line kernel: ONE kernel x48 %7.3f us/launch | 48 DISTINCT kernels %7.3f us/launch\n", same_s / 48, dist_s / 48);
    printf("code ~16 KB straight-line kernel: ONE kernel x48 %7.3f us/launch | 48 DISTINCT kernels %7.3f us/launch\n", same_l / 48, dist_l / 48);
    printf("code  ~4 KB, a 40 MB-traffic kernel between launches (pair): ONE %7.3f us/pair | DISTINCT %7.3f us/pair\n", same_t / 48, dist_t / 48);
    printf("code ~16 KB, a 40 MB-traffic kernel between launches (pair): ONE %7.3f us/pair | DISTINCT %7.3f us/pair\n", same_tl / 48, dist_tl / 48);
  }
  return 0;
}
