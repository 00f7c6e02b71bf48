/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Nothing under easyrec_amd/ may import, link or call this.
 *
 * CPU restatement of the id-hashing step of EasyRec's embedding hot path:
 *   HashedCategoricalColumn._transform_input_tensor
 *     reference easy_rec/python/compat/feature_column/feature_column_v2.py:3903-3926
 *     -> string_ops.string_to_hash_bucket_fast(input, hash_bucket_size)         (:3920-3921)
 * whose arithmetic lives in a third-party dependency that is NOT vendored in the reference
 * tree: TensorFlow (unpinned; docker images use 1.12-2.12) kernel StringToHashBucketFast =
 *   farmhash::Fingerprint64(bytes) % num_buckets        (Fingerprint64 == farmhashna::Hash64,
 *   google/farmhash, the version bundled by TF: commit 816a4ae622e964763ca0862d9dbd19324a1eaf45).
 * The algorithm below is restated from FarmHash's published description.
 *
 * Pinning (tests/test_oracle_hash.py): TF's documented example
 *   tf.strings.to_hash_bucket_fast(["Hello","TensorFlow","2.x"], 3) == [0, 2, 2]
 * and the four Fingerprint64 values quoted in TF's own string_to_hash_bucket_op_test.py
 *   'a' -> 12917804110809363939, 'b' -> 11795596070477164822,
 *   'c' -> 11430444447143000872, 'd' -> 4470636696479570465     (buckets mod 10: 9,2,2,5).
 * Branches for len > 16 have no external pin in this environment ("parity unpinned" for them);
 * Criteo / Taobao ids and decimal integers are all <= 16 bytes.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static const uint64_t K0 = 0xc3a5c85c97cb3127ULL;
static const uint64_t K1 = 0xb492b66fbe98f273ULL;
static const uint64_t K2 = 0x9ae16a3b2f90404fULL;

static uint64_t fetch64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v; /* little-endian host assumed (x86_64) */
}
static uint64_t fetch32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
static uint64_t rot(uint64_t v, int s) { return s == 0 ? v : (v >> s) | (v << (64 - s)); }
static uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

static uint64_t hash_len16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul;
  a ^= (a >> 47);
  uint64_t b = (v ^ a) * mul;
  b ^= (b >> 47);
  b *= mul;
  return b;
}

static uint64_t hash_0to16(const uint8_t* s, size_t len) {
  if (len >= 8) {
    uint64_t mul = K2 + len * 2;
    uint64_t a = fetch64(s) + K2;
    uint64_t b = fetch64(s + len - 8);
    uint64_t c = rot(b, 37) * mul + a;
    uint64_t d = (rot(a, 25) + b) * mul;
    return hash_len16(c, d, mul);
  }
  if (len >= 4) {
    uint64_t mul = K2 + len * 2;
    uint64_t a = fetch32(s);
    return hash_len16(len + (a << 3), fetch32(s + len - 4), mul);
  }
  if (len > 0) {
    uint8_t a = s[0];
    uint8_t b = s[len >> 1];
    uint8_t c = s[len - 1];
    uint32_t y = (uint32_t)a + ((uint32_t)b << 8);
    uint32_t z = (uint32_t)len + ((uint32_t)c << 2);
    return shift_mix(y * K2 ^ z * K0) * K2;
  }
  return K2;
}

static uint64_t hash_17to32(const uint8_t* s, size_t len) {
  uint64_t mul = K2 + len * 2;
  uint64_t a = fetch64(s) * K1;
  uint64_t b = fetch64(s + 8);
  uint64_t c = fetch64(s + len - 8) * mul;
  uint64_t d = fetch64(s + len - 16) * K2;
  return hash_len16(rot(a + b, 43) + rot(c, 30) + d, a + rot(b + K2, 18) + c, mul);
}

static uint64_t hash_33to64(const uint8_t* s, size_t len) {
  uint64_t mul = K2 + len * 2;
  uint64_t a = fetch64(s) * K2;
  uint64_t b = fetch64(s + 8);
  uint64_t c = fetch64(s + len - 8) * mul;
  uint64_t d = fetch64(s + len - 16) * K2;
  uint64_t y = rot(a + b, 43) + rot(c, 30) + d;
  uint64_t z = hash_len16(y, a + rot(b + K2, 18) + c, mul);
  uint64_t e = fetch64(s + 16) * mul;
  uint64_t f = fetch64(s + 24);
  uint64_t g = (y + fetch64(s + len - 32)) * mul;
  uint64_t h = (z + fetch64(s + len - 24)) * mul;
  return hash_len16(rot(e + f, 43) + rot(g, 30) + h, e + rot(f + a, 18) + g, mul);
}

typedef struct {
  uint64_t first, second;
} u128;

static u128 weak32(const uint8_t* p, uint64_t a, uint64_t b) {
  uint64_t w = fetch64(p), x = fetch64(p + 8), y = fetch64(p + 16), z = fetch64(p + 24);
  a += w;
  b = rot(b + a + z, 21);
  uint64_t c = a;
  a += x;
  a += y;
  b += rot(a, 44);
  u128 r = {a + z, b + c};
  return r;
}

uint64_t er_oracle_fingerprint64(const uint8_t* s, size_t len) {
  if (len <= 16) return hash_0to16(s, len);
  if (len <= 32) return hash_17to32(s, len);
  if (len <= 64) return hash_33to64(s, len);
  const uint64_t seed = 81;
  uint64_t x = seed;
  uint64_t y = seed * K1 + 113;
  uint64_t z = shift_mix(y * K2 + 113) * K2;
  u128 v = {0, 0}, w = {0, 0};
  x = x * K2 + fetch64(s);
  const uint8_t* end = s + ((len - 1) / 64) * 64;
  const uint8_t* last64 = end + ((len - 1) & 63) - 63;
  do {
    x = rot(x + y + v.first + fetch64(s + 8), 37) * K1;
    y = rot(y + v.second + fetch64(s + 48), 42) * K1;
    x ^= w.second;
    y += v.first + fetch64(s + 40);
    z = rot(z + w.first, 33) * K1;
    v = weak32(s, v.second * K1, x + w.first);
    w = weak32(s + 32, z + w.second, y + fetch64(s + 16));
    uint64_t t = z;
    z = x;
    x = t;
    s += 64;
  } while (s != end);
  uint64_t mul = K1 + ((z & 0xff) << 1);
  s = last64;
  w.first += ((len - 1) & 63);
  v.first += w.first;
  w.first += v.first;
  x = rot(x + y + v.first + fetch64(s + 8), 37) * mul;
  y = rot(y + v.second + fetch64(s + 48), 42) * mul;
  x ^= w.second * 9;
  y += v.first * 9 + fetch64(s + 40);
  z = rot(z + w.first, 33) * mul;
  v = weak32(s, v.second * mul, x + w.first);
  w = weak32(s + 32, z + w.second, y + fetch64(s + 16));
  uint64_t t = z;
  z = x;
  x = t;
  return hash_len16(hash_len16(v.first, w.first, mul) + shift_mix(y) * K0 + z,
                    hash_len16(v.second, w.second, mul) + x, mul);
}

/* string_to_hash_bucket_fast over a packed byte buffer: string i = bytes[offsets[i] .. offsets[i+1]).
 * Empty strings ARE hashed here (TF hashes ''); dropping '' is the caller's job
 * (reference compat/feature_column/feature_column.py:2599-2643). */
void er_oracle_hash_bucket_fast(const uint8_t* bytes, const int64_t* offsets, int64_t n,
                                uint64_t num_buckets, int64_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    uint64_t h = er_oracle_fingerprint64(bytes + offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
    out[i] = (int64_t)(h % num_buckets);
  }
}
