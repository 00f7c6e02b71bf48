// K1: string_to_hash_bucket_fast  (FarmHash Fingerprint64 mod num_buckets) on host and device.
// Replaces the StringToHashBucketFast ops issued at
//   reference easy_rec/python/compat/feature_column/feature_column_v2.py:3915-3921.
// HBM-bound integer/byte work: one lane per string, ids are the only output (8 B per string).
#include "er_common.h"

namespace er {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace fh {

constexpr uint64_t K0 = 0xc3a5c85c97cb3127ULL;
constexpr uint64_t K1 = 0xb492b66fbe98f273ULL;
constexpr uint64_t K2 = 0x9ae16a3b2f90404fULL;

// byte-wise little-endian loads: strings are unaligned inside the packed buffer
__host__ __device__ __forceinline__ uint64_t ld64(const uint8_t* p) {
  uint64_t v = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) v |= static_cast<uint64_t>(p[i]) << (8 * i);
  return v;
}
__host__ __device__ __forceinline__ uint64_t ld32(const uint8_t* p) {
  uint64_t v = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) v |= static_cast<uint64_t>(p[i]) << (8 * i);
  return v;
}
__host__ __device__ __forceinline__ uint64_t rotr(uint64_t v, int s) {
  return (v >> s) | (v << (64 - s));
}
__host__ __device__ __forceinline__ uint64_t smix(uint64_t v) { return v ^ (v >> 47); }
__host__ __device__ __forceinline__ uint64_t len16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul;
  a ^= (a >> 47);
  uint64_t b = (v ^ a) * mul;
  b ^= (b >> 47);
  return b * mul;
}

struct Pair {
  uint64_t a, b;
};
__host__ __device__ __forceinline__ Pair weak32(const uint8_t* p, uint64_t a, uint64_t b) {
  const uint64_t w = ld64(p), x = ld64(p + 8), y = ld64(p + 16), z = ld64(p + 24);
  a += w;
  b = rotr(b + a + z, 21);
  const uint64_t c = a;
  a += x;
  a += y;
  b += rotr(a, 44);
  return Pair{a + z, b + c};
}

__host__ __device__ inline uint64_t fingerprint64(const uint8_t* s, uint64_t len) {
  if (len <= 16) {
    if (len >= 8) {
      const uint64_t mul = K2 + len * 2;
      const uint64_t a = ld64(s) + K2;
      const uint64_t b = ld64(s + len - 8);
      const uint64_t c = rotr(b, 37) * mul + a;
      const uint64_t d = (rotr(a, 25) + b) * mul;
      return len16(c, d, mul);
    }
    if (len >= 4) {
      const uint64_t mul = K2 + len * 2;
      const uint64_t a = ld32(s);
      return len16(len + (a << 3), ld32(s + len - 4), mul);
    }
    if (len > 0) {
      const uint32_t a = s[0], b = s[len >> 1], c = s[len - 1];
      const uint32_t y = a + (b << 8);
      const uint32_t z = static_cast<uint32_t>(len) + (c << 2);
      return smix(y * K2 ^ z * K0) * K2;
    }
    return K2;
  }
  if (len <= 32) {
    const uint64_t mul = K2 + len * 2;
    const uint64_t a = ld64(s) * K1;
    const uint64_t b = ld64(s + 8);
    const uint64_t c = ld64(s + len - 8) * mul;
    const uint64_t d = ld64(s + len - 16) * K2;
    return len16(rotr(a + b, 43) + rotr(c, 30) + d, a + rotr(b + K2, 18) + c, mul);
  }
  if (len <= 64) {
    const uint64_t mul = K2 + len * 2;
    const uint64_t a = ld64(s) * K2;
    const uint64_t b = ld64(s + 8);
    const uint64_t c = ld64(s + len - 8) * mul;
    const uint64_t d = ld64(s + len - 16) * K2;
    const uint64_t y = rotr(a + b, 43) + rotr(c, 30) + d;
    const uint64_t z = len16(y, a + rotr(b + K2, 18) + c, mul);
    const uint64_t e = ld64(s + 16) * mul;
    const uint64_t f = ld64(s + 24);
    const uint64_t g = (y + ld64(s + len - 32)) * mul;
    const uint64_t h = (z + ld64(s + len - 24)) * mul;
    return len16(rotr(e + f, 43) + rotr(g, 30) + h, e + rotr(f + a, 18) + g, mul);
  }
  uint64_t x = 81;
  uint64_t y = 81 * K1 + 113;
  uint64_t z = smix(y * K2 + 113) * K2;
  Pair v{0, 0}, w{0, 0};
  x = x * K2 + ld64(s);
  const uint8_t* end = s + ((len - 1) / 64) * 64;
  const uint8_t* last64 = end + ((len - 1) & 63) - 63;
  do {
    x = rotr(x + y + v.a + ld64(s + 8), 37) * K1;
    y = rotr(y + v.b + ld64(s + 48), 42) * K1;
    x ^= w.b;
    y += v.a + ld64(s + 40);
    z = rotr(z + w.a, 33) * K1;
    v = weak32(s, v.b * K1, x + w.a);
    w = weak32(s + 32, z + w.b, y + ld64(s + 16));
    const uint64_t t = z;
    z = x;
    x = t;
    s += 64;
  } while (s != end);
  const uint64_t mul = K1 + ((z & 0xff) << 1);
  s = last64;
  w.a += ((len - 1) & 63);
  v.a += w.a;
  w.a += v.a;
  x = rotr(x + y + v.a + ld64(s + 8), 37) * mul;
  y = rotr(y + v.b + ld64(s + 48), 42) * mul;
  x ^= w.b * 9;
  y += v.a * 9 + ld64(s + 40);
  z = rotr(z + w.a, 33) * mul;
  v = weak32(s, v.b * mul, x + w.a);
  w = weak32(s + 32, z + w.b, y + ld64(s + 16));
  const uint64_t t = z;
  z = x;
  x = t;
  return len16(len16(v.a, w.a, mul) + smix(y) * K0 + z, len16(v.b, w.b, mul) + x, mul);
}

}  // namespace fh

__global__ void __launch_bounds__(kBlock)
hash_bucket_kernel(const uint8_t* __restrict__ bytes, const int64_t* __restrict__ offsets, int64_t n,
                   int64_t n_per_col, const uint64_t* __restrict__ num_buckets, int drop_empty,
                   int64_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t b = offsets[i], e = offsets[i + 1];
    const uint64_t len = static_cast<uint64_t>(e - b);
    int64_t r;
    if (len == 0 && drop_empty) {
      r = -1;
    } else {
      const uint64_t nb = num_buckets[i / n_per_col];
      r = static_cast<int64_t>(fh::fingerprint64(bytes + b, len) % nb);
    }
    out[i] = r;
  }
}

// AsString(int64) + hash, without materialising the decimal text in memory.
__global__ void __launch_bounds__(kBlock)
hash_bucket_int64_kernel(const int64_t* __restrict__ values, int64_t n, int64_t n_per_col,
                         const uint64_t* __restrict__ num_buckets, int64_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t val = values[i];
    uint8_t buf[24];
    int pos = 24;
    uint64_t mag = val < 0 ? (~static_cast<uint64_t>(val) + 1ULL) : static_cast<uint64_t>(val);
    do {
      buf[--pos] = static_cast<uint8_t>('0' + (mag % 10));
      mag /= 10;
    } while (mag);
    if (val < 0) buf[--pos] = '-';
    const uint64_t nb = num_buckets[i / n_per_col];
    out[i] = static_cast<int64_t>(fh::fingerprint64(buf + pos, static_cast<uint64_t>(24 - pos)) % nb);
  }
}

}  // namespace er

extern "C" {

int er_abi_version(void) { return ER_ABI_VERSION; }
const char* er_last_error(void) { return er::g_err; }

int er_device_info(int* cu_count, int* wave_size, char* arch_name, int arch_name_len) {
  int dev = 0;
  ER_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  ER_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (wave_size) *wave_size = prop.warpSize;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, prop.gcnArchName, static_cast<size_t>(arch_name_len) - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return 0;
}

int er_hash_bucket_fast_host(const uint8_t* bytes, const int64_t* offsets, int64_t n, int64_t n_per_col,
                             const uint64_t* num_buckets, int drop_empty, int64_t* out) {
  ER_REQUIRE(n >= 0 && n_per_col > 0, "er_hash_bucket_fast_host: bad sizes n=%lld n_per_col=%lld",
             (long long)n, (long long)n_per_col);
  for (int64_t i = 0; i < n; ++i) {
    const int64_t b = offsets[i], e = offsets[i + 1];
    const uint64_t nb = num_buckets[i / n_per_col];
    ER_REQUIRE(nb > 0, "er_hash_bucket_fast_host: num_buckets must be > 0");
    if (e == b && drop_empty) {
      out[i] = -1;
    } else {
      out[i] = static_cast<int64_t>(er::fh::fingerprint64(bytes + b, static_cast<uint64_t>(e - b)) % nb);
    }
  }
  return 0;
}

// TensorFlow's FingerprintCat64 (tensorflow/core/platform/fingerprint.h): the order-dependent mix SparseCross folds
// the columns' fingerprints with.
static inline uint64_t fingerprint_cat64(uint64_t fp1, uint64_t fp2) {
  const uint64_t k = 0xc6a4a7935bd1e995ULL;
  uint64_t r = fp1 ^ k;
  r ^= er::fh::smix(fp2 * k) * k;
  r *= k;
  r = er::fh::smix(r) * k;
  return er::fh::smix(r);
}

int er_sparse_cross_hashed_host(const uint8_t* bytes, const int64_t* offsets, int64_t n_rows, int32_t n_cols,
                                uint64_t num_buckets, uint64_t hash_key, int64_t* out) {
  ER_REQUIRE(bytes && offsets && out && n_rows >= 0 && n_cols >= 1 && num_buckets > 0,
             "er_sparse_cross_hashed_host: bad arguments");
  for (int64_t r = 0; r < n_rows; ++r) {
    uint64_t h = hash_key;
    for (int32_t c = 0; c < n_cols; ++c) {
      // ('' is a value like any other here: CrossedColumn hands the dense string tensors to the op as they are)
      const int64_t i = static_cast<int64_t>(c) * n_rows + r;
      const int64_t b = offsets[i], e = offsets[i + 1];
      h = fingerprint_cat64(h, er::fh::fingerprint64(bytes + b, static_cast<uint64_t>(e - b)));
    }
    out[r] = static_cast<int64_t>(h % num_buckets);
  }
  return 0;
}

int er_hash_bucket_fast(const uint8_t* bytes, const int64_t* offsets, int64_t n, int64_t n_per_col,
                        const uint64_t* num_buckets, int drop_empty, int64_t* out, er_stream_t stream) {
  ER_REQUIRE(n >= 0 && n_per_col > 0, "er_hash_bucket_fast: bad sizes");
  if (n == 0) return 0;
  const int blocks = static_cast<int>(er::ceil_div(n, er::kBlock) < 4096 ? er::ceil_div(n, er::kBlock) : 4096);
  hipLaunchKernelGGL(er::hash_bucket_kernel, dim3(blocks), dim3(er::kBlock), 0, er::as_stream(stream), bytes,
                     offsets, n, n_per_col, num_buckets, drop_empty, out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_hash_bucket_fast_int64(const int64_t* values, int64_t n, int64_t n_per_col, const uint64_t* num_buckets,
                              int64_t* out, er_stream_t stream) {
  ER_REQUIRE(n >= 0 && n_per_col > 0, "er_hash_bucket_fast_int64: bad sizes");
  if (n == 0) return 0;
  const int blocks = static_cast<int>(er::ceil_div(n, er::kBlock) < 4096 ? er::ceil_div(n, er::kBlock) : 4096);
  hipLaunchKernelGGL(er::hash_bucket_int64_kernel, dim3(blocks), dim3(er::kBlock), 0, er::as_stream(stream),
                     values, n, n_per_col, num_buckets, out);
  ER_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
