// FarmHash Fingerprint64 (the hash behind TensorFlow's string_to_hash_bucket_fast; reference call site
// easy_rec/python/compat/feature_column/feature_column_v2.py:3915-3921), host and device: shared by er_hash.hip (the
// standalone launch / host loop) and the step prologue in er_dense.hip, which hashes the batch's id strings in the same
// launch that selects the step's optimizer scalars.
#pragma once
#include "er_common.h"

namespace er {
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

// one id string -> bucket (drop_empty: '' -> -1, the "missing id" of the lookups)
__host__ __device__ __forceinline__ int64_t hash_bucket_one(const uint8_t* __restrict__ bytes, const int64_t* __restrict__ offsets,
                                                            int64_t i, int64_t n_per_col, const uint64_t* __restrict__ num_buckets,
                                                            int drop_empty) {
  const int64_t b = offsets[i], e = offsets[i + 1];
  const uint64_t len = static_cast<uint64_t>(e - b);
  if (len == 0 && drop_empty) return -1;
  const uint64_t nb = num_buckets[i / n_per_col];
  return static_cast<int64_t>(fh::fingerprint64(bytes + b, len) % nb);
}

}  // namespace er
