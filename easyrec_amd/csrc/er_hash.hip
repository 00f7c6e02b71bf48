// K1: string_to_hash_bucket_fast  (FarmHash Fingerprint64 mod num_buckets) on host and device.
// Replaces the StringToHashBucketFast ops issued at
//   reference easy_rec/python/compat/feature_column/feature_column_v2.py:3915-3921.
// HBM-bound integer/byte work: one lane per string, ids are the only output (8 B per string).
#include "er_common.h"
#include "er_farmhash.h"

namespace er {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}


__global__ void __launch_bounds__(kBlock)
hash_bucket_kernel(const uint8_t* __restrict__ bytes, const int64_t* __restrict__ offsets, int64_t n,
                   int64_t n_per_col, const uint64_t* __restrict__ num_buckets, int drop_empty,
                   int64_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    out[i] = hash_bucket_one(bytes, offsets, i, n_per_col, num_buckets, drop_empty);
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
