// K13b: the bf16 contraction with bf16 operands IN HBM (BASELINE config 3: "bf16 dense + fp32 emb").
//
// Replaces the MatMul of the DCN-v2 cross layer W.x_l (reference layers/keras/interaction.py:249-286) and of
// tf.layers.dense (layers/dnn.py:57-62) and its input gradient when the dense part runs in bf16.  er_gemm_bf16
// (er_gemm.hip) rounds fp32 operands while staging them: it moves fp32 bytes, spends VALU on the conversion and
// stages through registers - measured 57-86 TFLOP/s, no faster than the fp32 MFMA path.  Here the operands are bf16
// already (activations cast once by their producer, weights kept as bf16 shadows of the fp32 masters by the optimizer
// step), both k-contiguous ("NT": A[M,K], Bt[N,K]), and go HBM -> LDS without touching a register.
//
// MI355X design
//   * 128 x 128 output tile per workgroup of 4 waves (2 x 2, 64 x 64 per wave = 2 x 2 accumulators of
//     v_mfma_f32_32x32x16_bf16, 64 accumulator VGPRs), k-tile 64.
//   * global_load_lds_dwordx4: a wave instruction lands 64 x 16 B contiguously in LDS (8 rows x 128 B of the tile).
//     The tile image is XOR-swizzled - 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7) - by swizzling the
//     per-lane SOURCE address (the LDS side of the DMA is lane-linear): the fragment reads (32 rows, same logical
//     chunk, ds_read_b128) then spread over all 16 chunk slots of the 256-byte bank row instead of 2.
//   * out-of-range rows / k (M, N not multiples of 128, K not a multiple of 64) read a 16-byte zero buffer: no
//     branches around the DMA, zero contribution to the product.
//   * a ring of kStages LDS tile pairs; the loads of tile t + kStages - 1 are issued before the MFMAs of tile t and
//     stay in flight ACROSS the workgroup barrier (raw s_barrier + counted s_waitcnt vmcnt: __syncthreads() would
//     drain the DMA queue).  The contraction is short here (K = 624: 10 k-tiles), so what decides the launch's
//     duration is how early the HBM/L2 latency of the first tiles is paid and how many tiles are in flight.
//   * epilogue from the accumulators: + bias[col], optional C +=, fp32 C and/or a bf16 copy of C (the next
//     layer's operand) in one pass.
#include "er_common.h"

namespace er {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kNtM = 128, kNtN = 128, kNtK = 64;
constexpr int kNtOperandBytes = kNtM * kNtK * 2;       // 16 KB
constexpr int kNtStageBytes = 2 * kNtOperandBytes;     // A tile + B tile
constexpr int kNtThreads = 256;
// (8 global_load_lds per thread and k-tile: 4 for A, 4 for Bt - the counted waits below assume it)

struct NtArgs {
  const uint16_t* A;    // [M][lda] bf16
  const uint16_t* Bt;   // [N][ldb] bf16
  const uint16_t* zero; // >= 16 bytes of zeros
  float* C;             // [M][ldc] fp32 or nullptr
  uint16_t* Cb;         // [M][ldcb] bf16 or nullptr
  const float* bias;    // [N] or nullptr
  int M, N, K;
  int lda, ldb, ldc, ldcb;
  int accumulate;       // C += (fp32 output only)
  int debug;            // (micro-benchmarks only) 1: no epilogue stores, 2: no k-loop, 4: XCD-grouped tile order
  er_gemm_epilogue e;   // what the epilogue does besides + bias / C += (EPI template parameter = e.kind)
};

__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
  return static_cast<uint16_t>(u >> 16);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int STAGES, int EPI>
__global__ void __launch_bounds__(kNtThreads)
gemm_bf16_nt_kernel(NtArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [STAGES][A tile | Bt tile]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.debug & 4) {
    // consecutive workgroup ids go round-robin over the 8 XCDs: give each XCD a contiguous run of tiles (row-major:
    // the column tiles of one row block share the A rows in that XCD's L2)
    const int ncol = gridDim.x, total = gridDim.x * gridDim.y;
    const int id = blockIdx.y * gridDim.x + blockIdx.x;
    const int chunk = (total + 7) / 8;
    const int t = (id & 7) * chunk + (id >> 3);
    if (t >= total) return;  // (only when total is not a multiple of 8: those ids are remapped below)
    bx = t % ncol;
    by = t / ncol;
  }
  const int m0 = by * kNtM, n0 = bx * kNtN;
  const int T = (g.debug & 2) ? 0 : (g.K + kNtK - 1) / kNtK;

  // the 4 + 4 chunks this thread moves per k-tile: LDS chunk p = (j * 4 + w) * 64 + lane -> row p >> 3, slot p & 7,
  // holding logical chunk (slot ^ swizzle(row)) of that row.  Rows past M / N are clamped to the last row: what they
  // contribute lands in output rows / columns that are never stored.  Only k past K must read zeros (the zero chunk).
  uint32_t off_a[4], off_b[4];  // byte offsets from A / Bt (the operands are < 4 GB)
  int kc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = (j * 4 + w) * 64 + lane;
    const int row = p >> 3;
    kc[j] = ((p & 7) ^ ((row >> 1) & 7)) * 8;  // k offset (elements) inside the k-tile
    const int ra = m0 + row < g.M ? m0 + row : g.M - 1;
    const int rb = n0 + row < g.N ? n0 + row : g.N - 1;
    off_a[j] = (static_cast<uint32_t>(ra) * static_cast<uint32_t>(g.lda) + kc[j]) * 2u;
    off_b[j] = (static_cast<uint32_t>(rb) * static_cast<uint32_t>(g.ldb) + kc[j]) * 2u;
  }
  const char* base_a = reinterpret_cast<const char*>(g.A);
  const char* base_b = reinterpret_cast<const char*>(g.Bt);
  const int T_full = g.K / kNtK;  // k-tiles that lie completely inside K

  auto issue = [&](int kt) {
    unsigned char* base = lds + (kt % STAGES) * kNtStageBytes;
    const int k0 = kt * kNtK;
    if (kt < T_full && !(g.debug & 2)) {  // (wave-uniform) the common case: no per-chunk conditions
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned char* da = base + (j * 4 + w) * 1024;  // wave-uniform: the DMA adds lane * 16
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base_a + (off_a[j] + static_cast<uint32_t>(k0) * 2u)), (lds_ptr_t)da, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base_b + (off_b[j] + static_cast<uint32_t>(k0) * 2u)),
                                         (lds_ptr_t)(da + kNtOperandBytes), 16, 0, 0);
      }
    } else {  // the k-tail tile, and the tiles past the end (issued to keep the counted waits uniform)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool k_ok = kt < T && k0 + kc[j] < g.K;
        const void* pa = k_ok ? static_cast<const void*>(base_a + (off_a[j] + static_cast<uint32_t>(k0) * 2u)) : g.zero;
        const void* pb = k_ok ? static_cast<const void*>(base_b + (off_b[j] + static_cast<uint32_t>(k0) * 2u)) : g.zero;
        unsigned char* da = base + (j * 4 + w) * 1024;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)pa, (lds_ptr_t)da, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)pb, (lds_ptr_t)(da + kNtOperandBytes), 16, 0, 0);
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses inside a stage (bytes): row * 128 + ((chunk ^ swizzle(row)) * 16), chunk = 2 * kk + (lane >> 5)
  int a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wr * 64 + i * 32 + (lane & 31);
    const int rb = wc * 64 + i * 32 + (lane & 31);
    a_off[i] = ra * 128;
    b_off[i] = kNtOperandBytes + rb * 128;
    a_sw[i] = (ra >> 1) & 7;
    b_sw[i] = (rb >> 1) & 7;
  }

  static_assert(STAGES == 4, "the counted waits below assume a ring of 4 tile pairs");
  // Fragment reads as inline asm: hipcc then tracks no LDS operation of its own in the loop and inserts no
  // `s_waitcnt lgkmcnt(0)` (it cannot count across the loop's back edge, so its wait before a k-step's MFMAs also waited
  // for the reads of the NEXT k-step issued just before - no overlap); the counted waits are placed by hand: LDS
  // operations return in order, 4 reads per fragment set.
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_ptr_t)lds));
  auto read_frags = [&](uint32_t base, int kk, bf16x8* fa, bf16x8* fb) {
    const int chunk = 2 * kk + (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(fa[i]) : "v"(base + a_off[i] + ((chunk ^ a_sw[i]) << 4)));
      asm volatile("ds_read_b128 %0, %1" : "=v"(fb[i]) : "v"(base + b_off[i] + ((chunk ^ b_sw[i]) << 4)));
    }
  };
  // the set read before the 4 most recent reads has arrived (ties the wait to the registers the MFMAs consume)
  auto wait_set = [&](bf16x8* fa, bf16x8* fb) {
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]));
  };
  auto mfma_group = [&](const bf16x8* fa, const bf16x8* fb) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  };

  // Software pipeline over the k-tiles (one wave per SIMD: nothing but the wave's own instruction order hides
  // latencies).  Per iteration t (tile t = 4 k-steps of 16):
  //     reads(t,1)  mfma(t,0)  reads(t,2)  mfma(t,1)
  //     wait: tile t+1 landed (this thread's DMA: at most tile t+2 still in flight)  |  workgroup barrier
  //     dma(t+3)    reads(t,3)  mfma(t,2)  reads(t+1,0)  mfma(t,3)
  // The barrier sits in the MIDDLE of the tile: behind it every wave's DMA of tile t+1 is visible (so its first
  // fragments can be read before tile t is finished: no read latency at the tile boundary) and every wave has issued
  // all its reads of tile t-1, whose buffer dma(t+3) overwrites.  Fragment registers: two sets, alternating.
  bf16x8 a[2][2], b[2][2];
  // Cross epilogues: what the epilogue reads at this lane's 16 x 4 output positions is requested BEFORE the k loop (the
  // wave has registers to spare at one wave per SIMD) - with 160 workgroups in lock step an epilogue that starts its loads
  // after the last MFMA pays the HBM round trips on top of the contraction (22.5 us against 11.5 for the plain launch).
  constexpr bool kPre = (EPI == ER_EPI_CROSS_FWD || EPI == ER_EPI_CROSS_BWD);
  f32x4 pre_a[kPre ? 16 : 1], pre_b[kPre ? 16 : 1];
  if (kPre) {
    int pc = n0 + wc * 64 + (lane & 15) * 4;
    pc = pc + 4 <= g.N ? pc : (g.N >= 4 ? g.N - 4 : 0);
    const float* pa = EPI == ER_EPI_CROSS_FWD ? g.e.x0 : g.e.dout;
    const int lda_ = EPI == ER_EPI_CROSS_FWD ? g.e.ld_x0 : g.e.ld_dout;
    const float* pb = EPI == ER_EPI_CROSS_FWD ? g.e.xl : g.e.x0;  // (backward: x0 only with a lower cross layer)
    const int ldb_ = EPI == ER_EPI_CROSS_FWD ? g.e.ld_xl : g.e.ld_x0;
    const bool have_b = EPI == ER_EPI_CROSS_FWD || g.e.prev_u != nullptr;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      int row = m0 + wr * 64 + (lane >> 4) + it * 4;
      row = row < g.M ? row : g.M - 1;
      pre_a[it] = *reinterpret_cast<const f32x4*>(pa + static_cast<int64_t>(row) * lda_ + pc);
      if (have_b) pre_b[it] = *reinterpret_cast<const f32x4*>(pb + static_cast<int64_t>(row) * ldb_ + pc);
    }
  }
  issue(0);
  issue(1);
  issue(2);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(lds_base, 0, a[0], b[0]);
  for (int t = 0; t < T; ++t) {
    const uint32_t base = lds_base + (t % STAGES) * kNtStageBytes;
    const uint32_t next = lds_base + ((t + 1) % STAGES) * kNtStageBytes;
    read_frags(base, 1, a[1], b[1]);
    wait_set(a[0], b[0]);
    mfma_group(a[0], b[0]);
    read_frags(base, 2, a[0], b[0]);
    wait_set(a[1], b[1]);
    mfma_group(a[1], b[1]);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(t + 3);
    read_frags(base, 3, a[1], b[1]);
    wait_set(a[0], b[0]);
    mfma_group(a[0], b[0]);
    read_frags(next, 0, a[0], b[0]);
    wait_set(a[1], b[1]);
    mfma_group(a[1], b[1]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the fragments read ahead for a tile that does not exist)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the (zero-source) tiles issued past the end

  if (g.debug & 1) return;
  // Epilogue through LDS: the accumulator layout (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5): one
  // dword per lane and store) would take 64 dword stores per lane; each wave transposes its 64 x 64 block through a
  // private LDS region instead and stores rows as 16-byte pieces (16 lanes = 256 contiguous bytes of a row).
  __builtin_amdgcn_s_barrier();  // (no DMA in flight: every wave is done with the operand tiles)
  asm volatile("" ::: "memory");
  constexpr int kLd = 68;  // floats per staged row (64 + 4: the 16-byte reads of 4 rows spread over the banks)
  float* stage = reinterpret_cast<float*>(lds) + w * 64 * kLd;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * kLd + j * 32 + (lane & 31)] = acc[i][j][r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (same wave reads what it wrote: no workgroup barrier needed)
  const int c4 = (lane & 15) * 4;
  const int col = n0 + wc * 64 + c4;
  const int rsub = lane >> 4;
  const int row0 = m0 + wr * 64;
  const int row_base = row0 + rsub;
  const bool vec = (g.N % 4 == 0) && (!g.C || ((g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0))) &&
                   (!g.Cb || ((g.ldcb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.Cb) & 7) == 0)));
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (col + q < g.N) bv[q] = g.bias[col + q];
  }
  if (col >= g.N) return;  // (lanes l, l ^ 16, l ^ 32 share their columns: the exchanges below stay among live lanes)
  auto staged = [&](int it) { return *reinterpret_cast<const f32x4*>(&stage[(it * 4 + rsub) * kLd + c4]); };
  auto store_out = [&](int row, f32x4 v) {
    if (g.C) {
      f32x4* c = reinterpret_cast<f32x4*>(g.C + static_cast<int64_t>(row) * g.ldc + col);
      if (g.accumulate) {
        const f32x4 old = *c;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += old[q];
      }
      *c = v;
    }
    if (g.Cb) {
      ushort4 h;
      h.x = f32_to_bf16(v[0]); h.y = f32_to_bf16(v[1]); h.z = f32_to_bf16(v[2]); h.w = f32_to_bf16(v[3]);
      *reinterpret_cast<ushort4*>(g.Cb + static_cast<int64_t>(row) * g.ldcb + col) = h;
    }
  };
  // sum over the four lanes (rsub = 0 .. 3) that hold the other rows of this lane's columns; the same bits in all four
  auto rows_sum = [&](float v) {
    v = v + __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
  };
  const er_gemm_epilogue& e = g.e;
  const int tile64 = row0 / 64;             // the 64-row tile (er_gemm_row_tiles) this wave's block is
  const bool tile_live = row0 < g.M;
  if (EPI == ER_EPI_PLAIN) {
    if (vec) {  // (col + 3 < N follows from N % 4 == 0)
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int row = row_base + it * 4;
        if (row >= g.M) break;
        f32x4 v = staged(it);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += bv[q];
        store_out(row, v);
      }
    } else {
      for (int it = 0; it < 16; ++it) {
        const int row = row_base + it * 4;
        if (row >= g.M) break;
        for (int q = 0; q < 4 && col + q < g.N; ++q) {
          float v = stage[(it * 4 + rsub) * kLd + c4 + q] + bv[q];
          if (g.C) {
            float* c = g.C + static_cast<int64_t>(row) * g.ldc + col + q;
            if (g.accumulate) v += *c;
            *c = v;
          }
          if (g.Cb) g.Cb[static_cast<int64_t>(row) * g.ldcb + col + q] = f32_to_bf16(v);
        }
      }
    }
  } else if (EPI == ER_EPI_STATS) {
    // Welford triple of the wave's 64 rows per column: two passes over the staged block (mean, then M2 about it)
    const int nvalid = g.M - row0 < 64 ? (g.M - row0 > 0 ? g.M - row0 : 0) : 64;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int row = row_base + it * 4;
      if (row < g.M) {
        f32x4 v = staged(it);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] += bv[q]; s[q] += v[q]; }
        store_out(row, v);
      }
    }
    float mean[4], m2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) mean[q] = nvalid > 0 ? rows_sum(s[q]) / static_cast<float>(nvalid) : 0.f;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int row = row_base + it * 4;
      if (row < g.M) {
        const f32x4 v = staged(it);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float d = (v[q] + bv[q]) - mean[q]; m2[q] += d * d; }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) m2[q] = rows_sum(m2[q]);
    if (rsub == 0 && tile_live) {
      float* o = e.col_stats + (static_cast<int64_t>(tile64) * g.N + col) * 3;
#pragma unroll
      for (int q = 0; q < 4; ++q) { o[q * 3 + 0] = static_cast<float>(nvalid); o[q * 3 + 1] = mean[q]; o[q * 3 + 2] = m2[q]; }
    }
  } else if (EPI == ER_EPI_BN_BWD) {
    // (sum g, sum g * xhat) of the wave's 64 rows per source column; z / y of 8 rows in flight per lane
    const int cs = col - e.bn_col0;
    const bool in_block = cs >= 0 && cs < e.bn_n_src;
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
    float zb[4] = {0.f, 0.f, 0.f, 0.f}, mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f};
    if (in_block) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (e.bn_zbias) zb[q] = e.bn_zbias[cs + q];
        if (e.bn_use_bn) { mu[q] = e.bn_mean[cs + q]; is[q] = e.bn_invstd[cs + q]; }
      }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 yv[8], zv[8];
      if (in_block) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          int row = row_base + (half * 8 + k) * 4;
          row = row < g.M ? row : g.M - 1;
          const int64_t i = static_cast<int64_t>(row) * e.bn_ld + cs;
          yv[k] = *reinterpret_cast<const f32x4*>(e.bn_y + i);
          zv[k] = *reinterpret_cast<const f32x4*>(e.bn_z + i);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int it = half * 8 + k;
        const int row = row_base + it * 4;
        if (row < g.M) {
          f32x4 v = staged(it);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += bv[q];
          store_out(row, v);
          if (in_block) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float gq = v[q];
              if (e.bn_act == ER_ACT_RELU && !(yv[k][q] > 0.f)) gq = 0.f;
              sg[q] = sg[q] + gq;
              if (e.bn_use_bn) sgx[q] = sgx[q] + gq * ((zv[k][q] + zb[q] - mu[q]) * is[q]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { sg[q] = rows_sum(sg[q]); sgx[q] = rows_sum(sgx[q]); }
    if (rsub == 0 && in_block && tile_live) {
      float* o = e.bn_partial + (static_cast<int64_t>(tile64) * e.bn_n_src + cs) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) { o[q * 2] = sg[q]; o[q * 2 + 1] = sgx[q]; }
    }
  } else if (EPI == ER_EPI_CROSS_FWD) {
    // x_{l+1} = x0 * (acc + b + diag * x_l) + x_l (reference layers/keras/interaction.py:276-286); u keeps acc
    const bool with_diag = e.diag != 0.f;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = row_base + it * 4;
      if (row < g.M) {
        const f32x4 a = staged(it);
        if (e.u) *reinterpret_cast<f32x4*>(e.u + static_cast<int64_t>(row) * e.ld_u + col) = a;
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float t = a[q] + bv[q];
          if (with_diag) t = t + e.diag * pre_b[it][q];
          o[q] = pre_a[it][q] * t + pre_b[it][q];
        }
        store_out(row, o);
      }
    }
  } else if (EPI == ER_EPI_CROSS_BWD) {
    // v = acc + dout + diag * du_in: the whole gradient of x_{l-1}; the lower cross layer's elementwise backward on it
    const bool with_diag = e.diag != 0.f;
    const bool prev = e.prev_u != nullptr;
    float pb[4] = {0.f, 0.f, 0.f, 0.f}, cs4[4] = {0.f, 0.f, 0.f, 0.f};
    if (prev && e.prev_bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) pb[q] = e.prev_bias[col + q];
    }
#pragma unroll
    for (int part = 0; part < 4; ++part) {
      f32x4 gv[4], dv[4], x0v[4], uv[4], xlv[4], o0[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int row = row_base + (part * 4 + k) * 4;
        row = row < g.M ? row : g.M - 1;
        const int64_t r = row;
        gv[k] = pre_a[part * 4 + k];
        if (with_diag) dv[k] = *reinterpret_cast<const f32x4*>(e.du_in + r * e.ld_du_in + col);
        if (prev) {
          x0v[k] = pre_b[part * 4 + k];
          uv[k] = *reinterpret_cast<const f32x4*>(e.prev_u + r * e.ld_prev_u + col);
          if (with_diag) xlv[k] = *reinterpret_cast<const f32x4*>(e.xl + r * e.ld_xl + col);
          if (e.accumulate_dx0) o0[k] = *reinterpret_cast<const f32x4*>(e.dx0 + r * e.ld_dx0 + col);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int it = part * 4 + k;
        const int row = row_base + it * 4;
        if (row < g.M) {
          f32x4 v = staged(it);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[q] = v[q] + bv[q] + gv[k][q];
            if (with_diag) v[q] = v[q] + e.diag * dv[k][q];
          }
          store_out(row, v);
          if (prev) {
            f32x4 du, d0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              du[q] = v[q] * x0v[k][q];
              float t = uv[k][q] + pb[q];
              if (with_diag) t = t + e.diag * xlv[k][q];
              const float a = v[q] * t;
              d0[q] = e.accumulate_dx0 ? o0[k][q] + a : a;
              cs4[q] = cs4[q] + du[q];
            }
            const int64_t r = row;
            *reinterpret_cast<f32x4*>(e.dx0 + r * e.ld_dx0 + col) = d0;
            if (e.du_out) *reinterpret_cast<f32x4*>(e.du_out + r * e.ld_du_out + col) = du;
            if (e.du_out_bf16) {
              ushort4 h;
              h.x = f32_to_bf16(du[0]); h.y = f32_to_bf16(du[1]); h.z = f32_to_bf16(du[2]); h.w = f32_to_bf16(du[3]);
              *reinterpret_cast<ushort4*>(e.du_out_bf16 + r * e.ld_du_out_bf16 + col) = h;
            }
          }
        }
      }
    }
    if (prev && e.partial) {
#pragma unroll
      for (int q = 0; q < 4; ++q) cs4[q] = rows_sum(cs4[q]);
      if (rsub == 0 && tile_live) {
        float* o = e.partial + static_cast<int64_t>(tile64) * g.N + col;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = cs4[q];
      }
    }
  }
}

// fp32 -> bf16 (RNE), optionally transposed: dst[c][r] = src[r][c].  One descriptor per matrix, any number of
// matrices in one launch (the weights' shadows after an optimizer step; an activation matrix for a GEMM).
constexpr int kMaxCast = 16;
struct CastMulti {
  int n;
  int start[kMaxCast + 1];
  er_cast_desc d[kMaxCast];
};

__global__ void __launch_bounds__(256)
cast_bf16_kernel(CastMulti cm) {
  int i = 0;
  while (i + 1 < cm.n && static_cast<int>(blockIdx.x) >= cm.start[i + 1]) ++i;
  const er_cast_desc& d = cm.d[i];
  const int bid = blockIdx.x - cm.start[i];
  if (!d.transpose) {
    // 4 elements per thread along the row
    const int64_t q = static_cast<int64_t>(bid) * 256 + threadIdx.x;
    const int per_row = (d.cols + 3) / 4;
    const int64_t r = q / per_row;
    const int c = static_cast<int>(q % per_row) * 4;
    if (r >= d.rows) return;
    const float* s = d.src + r * d.ld_src + c;
    uint16_t* o = d.dst + r * d.ld_dst + c;
    if (c + 3 < d.cols && ((reinterpret_cast<uintptr_t>(s) & 15) == 0) && ((reinterpret_cast<uintptr_t>(o) & 7) == 0)) {
      const float4 v = *reinterpret_cast<const float4*>(s);
      ushort4 h;
      h.x = f32_to_bf16(v.x); h.y = f32_to_bf16(v.y); h.z = f32_to_bf16(v.z); h.w = f32_to_bf16(v.w);
      *reinterpret_cast<ushort4*>(o) = h;
    } else {
      for (int j = 0; j < 4 && c + j < d.cols; ++j) o[j] = f32_to_bf16(s[j]);
    }
    // zero the padding columns [cols, ld_dst) of the destination once per row (they are read as k-tail chunks)
    if (c == 0) for (int j = d.cols; j < d.ld_dst; ++j) d.dst[r * d.ld_dst + j] = 0;
  } else {
    // 32 x 32 tiles through LDS: coalesced reads along src rows, coalesced writes along dst rows
    __shared__ float tile[32][33];
    const int tiles_c = (d.cols + 31) / 32;
    const int tr = bid / tiles_c, tc = bid % tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
      const int64_t r = static_cast<int64_t>(tr) * 32 + k;
      const int c = tc * 32 + tx;
      tile[k][tx] = (r < d.rows && c < d.cols) ? d.src[r * d.ld_src + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      const int c = tc * 32 + k;                                 // dst row
      const int64_t r = static_cast<int64_t>(tr) * 32 + tx;      // dst col
      if (c < d.cols && r < d.ld_dst) d.dst[static_cast<int64_t>(c) * d.ld_dst + r] = r < d.rows ? f32_to_bf16(tile[tx][k]) : 0;
    }
  }
}

}  // namespace er

namespace {

uint16_t* g_zero16 = nullptr;
int ensure_zero() {
  if (!g_zero16) {
    ER_CHECK_HIP(hipMalloc(&g_zero16, 256));
    ER_CHECK_HIP(hipMemset(g_zero16, 0, 256));
  }
  return 0;
}

constexpr int kNtStages = 4;
bool g_nt_attr_set = false;

}  // namespace

extern "C" {

int er_gemm_bf16_nt_prepare(void) {
  if (int rc = ensure_zero()) return rc;
  if (!g_nt_attr_set) {
#define ER_NT_ATTR(EPI)                                                                                         \
  ER_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&er::gemm_bf16_nt_kernel<kNtStages, EPI>),     \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kNtStages * er::kNtStageBytes))
    ER_NT_ATTR(ER_EPI_PLAIN);
    ER_NT_ATTR(ER_EPI_STATS);
    ER_NT_ATTR(ER_EPI_BN_BWD);
    ER_NT_ATTR(ER_EPI_CROSS_FWD);
    ER_NT_ATTR(ER_EPI_CROSS_BWD);
#undef ER_NT_ATTR
    g_nt_attr_set = true;
  }
  return 0;
}

int er_gemm_bf16_nt_epi(int32_t M, int32_t N, int32_t K, const uint16_t* A, int32_t lda, const uint16_t* Bt, int32_t ldb,
                        float* C, int32_t ldc, uint16_t* C_bf16, int32_t ldc_bf16, const float* bias, int accumulate,
                        const er_gemm_epilogue* epi, er_stream_t stream) {
  ER_REQUIRE(A && Bt && (C || C_bf16) && M > 0 && N > 0 && K > 0, "er_gemm_bf16_nt: bad arguments");
  ER_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K,
             "er_gemm_bf16_nt: K and the leading dimensions must be multiples of 8 bf16 (16-byte chunks)");
  ER_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bt)) & 15) == 0,
             "er_gemm_bf16_nt: operands must be 16-byte aligned");
  ER_REQUIRE((!C || ldc >= N) && (!C_bf16 || ldc_bf16 >= N), "er_gemm_bf16_nt: leading dimension of C too small");
  ER_REQUIRE(!accumulate || C, "er_gemm_bf16_nt: accumulate needs the fp32 output");
  ER_REQUIRE(static_cast<int64_t>(M) * lda < (1LL << 31) && static_cast<int64_t>(N) * ldb < (1LL << 31),
             "er_gemm_bf16_nt: an operand larger than 4 GB");
  ER_REQUIRE(g_zero16 && g_nt_attr_set, "er_gemm_bf16_nt: call er_gemm_bf16_nt_prepare() first (not capturable)");
  er::NtArgs a;
  a.A = A; a.Bt = Bt; a.zero = g_zero16; a.C = C; a.Cb = C_bf16; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldcb = ldc_bf16; a.accumulate = accumulate;
  memset(&a.e, 0, sizeof(a.e));
  const int kind = epi ? epi->kind : ER_EPI_PLAIN;
  if (kind != ER_EPI_PLAIN) {
    a.e = *epi;
    er_gemm_epilogue& e = a.e;
    auto al16 = [](const void* p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 4 == 0; };
    ER_REQUIRE(N % 4 == 0 && (!C || al16(C, ldc)) && (!C_bf16 || ((reinterpret_cast<uintptr_t>(C_bf16) & 7) == 0 && ldc_bf16 % 4 == 0)),
               "er_gemm_bf16_nt_epi: an epilogue needs N %% 4 == 0 and aligned output rows");
    ER_REQUIRE(!bias || (reinterpret_cast<uintptr_t>(bias) & 3) == 0, "er_gemm_bf16_nt_epi: bias");
    switch (kind) {
      case ER_EPI_STATS:
        ER_REQUIRE(e.col_stats && !accumulate, "er_gemm_bf16_nt_epi: STATS needs col_stats and a plain output");
        break;
      case ER_EPI_BN_BWD:
        if (e.bn_n_src == 0) { e.bn_col0 = 0; e.bn_n_src = N; }
        ER_REQUIRE(e.bn_z && e.bn_y && e.bn_partial && e.bn_col0 >= 0 && e.bn_col0 % 4 == 0 && e.bn_n_src % 4 == 0 &&
                       e.bn_col0 + e.bn_n_src <= N && e.bn_ld >= e.bn_n_src && al16(e.bn_z, e.bn_ld) && al16(e.bn_y, e.bn_ld) &&
                       (!e.bn_use_bn || (e.bn_mean && e.bn_invstd)),
                   "er_gemm_bf16_nt_epi: bad BatchNorm-backward epilogue arguments");
        break;
      case ER_EPI_CROSS_FWD:
        ER_REQUIRE(e.x0 && e.xl && e.ld_x0 >= N && e.ld_xl >= N && al16(e.x0, e.ld_x0) && al16(e.xl, e.ld_xl) &&
                       (!e.u || (e.ld_u >= N && al16(e.u, e.ld_u))) && !accumulate,
                   "er_gemm_bf16_nt_epi: bad cross-forward epilogue arguments");
        break;
      case ER_EPI_CROSS_BWD:
        ER_REQUIRE(e.dout && e.ld_dout >= N && al16(e.dout, e.ld_dout) &&
                       (e.diag == 0.f || (e.du_in && e.ld_du_in >= N && al16(e.du_in, e.ld_du_in))),
                   "er_gemm_bf16_nt_epi: bad cross-backward epilogue arguments (dout / du_in)");
        if (e.prev_u) {
          ER_REQUIRE(e.x0 && e.dx0 && e.ld_x0 >= N && e.ld_prev_u >= N && e.ld_dx0 >= N && al16(e.x0, e.ld_x0) &&
                         al16(e.prev_u, e.ld_prev_u) && al16(e.dx0, e.ld_dx0) &&
                         (!e.du_out || (e.ld_du_out >= N && al16(e.du_out, e.ld_du_out))) &&
                         (!e.du_out_bf16 || (e.ld_du_out_bf16 >= N && e.ld_du_out_bf16 % 4 == 0 &&
                                             (reinterpret_cast<uintptr_t>(e.du_out_bf16) & 7) == 0)) &&
                         (e.diag == 0.f || (e.xl && e.ld_xl >= N && al16(e.xl, e.ld_xl))) &&
                         (!e.prev_bias || (reinterpret_cast<uintptr_t>(e.prev_bias) & 3) == 0),
                     "er_gemm_bf16_nt_epi: bad cross-backward epilogue arguments (the lower layer's part)");
        }
        break;
      default:
        ER_REQUIRE(false, "er_gemm_bf16_nt_epi: unknown epilogue kind %d", kind);
    }
  }
  static const int debug_flags = [] {  // (micro-benchmarks: tools/gemm_bf16_bench.py)
    const char* dbg = getenv("ER_NT_DEBUG");
    return dbg ? atoi(dbg) : 0;
  }();
  a.debug = debug_flags;
  dim3 grid(static_cast<unsigned>(er::ceil_div(N, er::kNtN)), static_cast<unsigned>(er::ceil_div(M, er::kNtM)));
  const size_t lds = kNtStages * er::kNtStageBytes;
  hipStream_t s = er::as_stream(stream);
  switch (kind) {
    case ER_EPI_STATS: hipLaunchKernelGGL((er::gemm_bf16_nt_kernel<kNtStages, ER_EPI_STATS>), grid, dim3(er::kNtThreads), lds, s, a); break;
    case ER_EPI_BN_BWD: hipLaunchKernelGGL((er::gemm_bf16_nt_kernel<kNtStages, ER_EPI_BN_BWD>), grid, dim3(er::kNtThreads), lds, s, a); break;
    case ER_EPI_CROSS_FWD: hipLaunchKernelGGL((er::gemm_bf16_nt_kernel<kNtStages, ER_EPI_CROSS_FWD>), grid, dim3(er::kNtThreads), lds, s, a); break;
    case ER_EPI_CROSS_BWD: hipLaunchKernelGGL((er::gemm_bf16_nt_kernel<kNtStages, ER_EPI_CROSS_BWD>), grid, dim3(er::kNtThreads), lds, s, a); break;
    default: hipLaunchKernelGGL((er::gemm_bf16_nt_kernel<kNtStages, ER_EPI_PLAIN>), grid, dim3(er::kNtThreads), lds, s, a); break;
  }
  ER_LAUNCH_CHECK();
  return 0;
}

int er_gemm_bf16_nt(int32_t M, int32_t N, int32_t K, const uint16_t* A, int32_t lda, const uint16_t* Bt, int32_t ldb,
                    float* C, int32_t ldc, uint16_t* C_bf16, int32_t ldc_bf16, const float* bias, int accumulate,
                    er_stream_t stream) {
  return er_gemm_bf16_nt_epi(M, N, K, A, lda, Bt, ldb, C, ldc, C_bf16, ldc_bf16, bias, accumulate, nullptr, stream);
}

int er_cast_bf16(const er_cast_desc* descs_host, int n, er_stream_t stream) {
  ER_REQUIRE(descs_host && n >= 1, "er_cast_bf16: bad arguments");
  for (int base = 0; base < n; base += er::kMaxCast) {
    er::CastMulti cm;
    cm.n = 0;
    cm.start[0] = 0;
    for (int i = base; i < n && i < base + er::kMaxCast; ++i) {
      const er_cast_desc& d = descs_host[i];
      ER_REQUIRE(d.src && d.dst && d.rows > 0 && d.cols > 0 && d.ld_src >= d.cols, "er_cast_bf16: bad descriptor %d", i);
      ER_REQUIRE(d.ld_dst >= (d.transpose ? d.rows : d.cols), "er_cast_bf16: descriptor %d: ld_dst too small", i);
      const int64_t blocks = d.transpose ? er::ceil_div(d.rows, 32) * er::ceil_div(d.cols, 32)
                                         : er::ceil_div(d.rows * er::ceil_div(d.cols, 4), 256);
      ER_REQUIRE(cm.start[cm.n] + blocks < 0x7FFFFFFFLL, "er_cast_bf16: too large for one launch");
      cm.d[cm.n] = d;
      cm.start[cm.n + 1] = cm.start[cm.n] + static_cast<int>(blocks);
      ++cm.n;
    }
    hipLaunchKernelGGL(er::cast_bf16_kernel, dim3(cm.start[cm.n]), dim3(256), 0, er::as_stream(stream), cm);
    ER_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
