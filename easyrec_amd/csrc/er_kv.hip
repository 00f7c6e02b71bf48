// K1b: hash-table ("KV") embedding tables: rows exist only for the ids that have been looked up in training.
//
// Replaces what `ev_params` turns an embedding column into in the reference: PAI-TF's `get_embedding_variable`
// (compat/feature_column/feature_column_v2.py:3478-3513) / SOK's `DynamicVariable` under embedding parallelism
// (compat/feature_column/feature_column.py:470-503, compat/dynamic_variable.py) - a key -> row map that creates a row,
// drawn from the column's initializer, the first time a key is looked up in training and serves zeros for unseen keys
// at evaluation (feature_column_v2.py:3487-3493).  Both are closed dependencies that are absent from /root/reference;
// what is restated here is the behaviour the reference's call sites rely on.
//
// MI355X design: the table's rows are an ARENA inside the ordinary table-group storage ([capacity, dim] var / m / v), so
// everything after the id translation - sort, catch-up, lookup, segmented reduction, row optimizer, lazy decay,
// checkpoints - is the dense-table path unchanged.  The translation id -> arena row is an open-addressing hash map in
// HBM (linear probing, 64-bit keys, load factor <= 0.5) in TWO launches per lookup array:
//   insert: every id claims its key with one 64-bit CAS; the winner of a new key takes the next arena row (one atomic
//           counter), initialises the row and publishes the row index;
//   find  : every id probes to its key and reads the row index (all published: the launch boundary orders it).
// Two launches instead of a spin on the winner: lanes of one wavefront may hold the same new key, and a lane spinning
// for another lane of its own wave never sees it progress.  Which key gets which arena row depends on the atomics'
// order and differs from run to run; nothing computed from the rows does (a row's value is a pure function of its
// key: rows are initialised from a counter-based generator keyed by (seed, key, column)).
//
// ev_params.filter_freq / steps_to_live (feature_config.proto:27-29, handed to PAI-TF / DeepRec as
// CounterFilter(filter_freq) and get_embedding_variable(steps_to_live=...), feature_column_v2.py:3497-3512).  The
// filters themselves are DeepRec's (closed here); what is restated is their documented behaviour:
//   CounterFilter  : the map counts a key's occurrences in training lookups (`freq`, one per occurrence); the key gets its
//                    row in the launch in which the count reaches filter_freq, until then it reads the no-permission
//                    default (zeros) and takes no update.  A key without a row still occupies a map slot, so the map of
//                    a filtered table is sized for the ids SEEN (`n_keys` counts them, the overflow flag guards it).
//   GlobalStepEvict: every training lookup stamps the key with the global step (`version`); when a checkpoint is
//                    written, keys with global_step - version > steps_to_live are dropped and the arena is compacted
//                    (host-driven: er_kv_export_all -> keep list -> er_kv_rebuild; layers/input_layer.py evict_stale).
#include "er_common.h"

namespace er {

constexpr int64_t kKvEmpty = -1;
inline int blocks_for(int64_t n) { return static_cast<int>(ceil_div(n, kBlock)); }

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// value of column c of the row of `key`: mean + stddev * z, z = ((u0 + u1) + (u2 + u3) - 2) * sqrt(3) with four
// 24-bit uniforms (sum of four U(0,1): variance 1/3 -> unit variance, support +-3.46 sigma; fp32, this operation order:
// oracle/kernel_ref.py kv_init_value restates it bit for bit)
__device__ __forceinline__ float kv_init_value(uint64_t seed, int64_t key, int c, float mean, float stddev) {
  const uint64_t base = mix64(seed ^ mix64(static_cast<uint64_t>(key))) + static_cast<uint64_t>(c) * 0x9E3779B97F4A7C15ull;
  float u[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    u[k] = static_cast<float>(mix64(base + static_cast<uint64_t>(k) * 0xD1B54A32D192ED03ull) >> 40) * 5.9604644775390625e-08f;
  const float z = (((u[0] + u[1]) + (u[2] + u[3])) - 2.0f) * 1.7320508075688772f;
  return mean + stddev * z;
}

__device__ __forceinline__ uint64_t kv_home(int64_t key, uint64_t mask) { return mix64(static_cast<uint64_t>(key)) & mask; }

// the optional per-slot state of a filtered / evicting table (all NULL / 0: the plain table)
struct KvFilter {
  int32_t* freq;        // occurrences counted so far (stops counting once it has reached filter_freq)
  int32_t* version;     // global step of the last training lookup
  int32_t* n_keys;      // occupied slots (a filtered table tracks keys that have no row yet)
  const int64_t* step;  // device step counter (read once per lane)
  int32_t filter_freq;
};

__device__ __forceinline__ void kv_create_row(int64_t key, uint64_t pos, int32_t* rows, int32_t* next_row, int32_t capacity,
                                              float* __restrict__ var, int dim, int ld, uint64_t seed, float mean, float stddev,
                                              int32_t* overflow) {
  const int32_t r = atomicAdd(next_row, 1);
  if (r >= capacity) {
    atomicExch(overflow, 1);  // the key stays without a row (find returns -1 for it: a zero embedding)
    return;
  }
  float* dst = var + static_cast<int64_t>(r) * ld;  // (ld: floats between rows - er_kv_job.table_ld)
  for (int c = 0; c < dim; ++c) dst[c] = kv_init_value(seed, key, c, mean, stddev);
  __hip_atomic_store(rows + pos, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// CounterFilter / version stamp: every occurrence of a key in a training lookup lands here once
__device__ __forceinline__ void kv_insert_filtered(int64_t key, uint64_t pos, uint64_t mask, int64_t* keys, int32_t* rows,
                                                   int32_t* next_row, int32_t capacity, float* __restrict__ var, int dim,
                                                   int ld, uint64_t seed, float mean, float stddev, int32_t* overflow,
                                                   const KvFilter& f) {
  const bool counted = f.filter_freq > 1;
  // a full map (filtered tables: slots in use, else rows in use) claims no further slot
  const bool full = counted ? __hip_atomic_load(f.n_keys, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= static_cast<int32_t>((mask + 1) >> 1)
                            : __hip_atomic_load(next_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= capacity;
  const int32_t now = f.step ? static_cast<int32_t>(*f.step) : 0;
  for (uint64_t probes = 0; probes <= mask; ++probes, pos = (pos + 1) & mask) {
    int64_t prev;
    if (full) {
      prev = __hip_atomic_load(keys + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == kKvEmpty) break;
    } else {
      prev = static_cast<int64_t>(atomicCAS(reinterpret_cast<unsigned long long*>(keys + pos),
                                            static_cast<unsigned long long>(kKvEmpty), static_cast<unsigned long long>(key)));
    }
    if (prev != key && prev != kKvEmpty) continue;
    const bool claimed = prev == kKvEmpty;
    if (claimed && counted && atomicAdd(f.n_keys, 1) >= static_cast<int32_t>((mask + 1) >> 1)) atomicExch(overflow, 1);
    if (f.version) __hip_atomic_store(f.version + pos, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool create = claimed;
    if (counted) {
      create = false;
      if (__hip_atomic_load(f.freq + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < f.filter_freq)
        create = atomicAdd(f.freq + pos, 1) + 1 == f.filter_freq;  // exactly one occurrence brings the count to the threshold
    }
    if (create) kv_create_row(key, pos, rows, next_row, capacity, var, dim, ld, seed, mean, stddev, overflow);
    return;
  }
  atomicExch(overflow, 1);
}

__device__ __forceinline__ void kv_insert_one(int64_t i, const int64_t* __restrict__ ids, int64_t n, int64_t* keys,
                                              int32_t* rows, uint64_t mask, int32_t* next_row, int32_t capacity,
                                              float* __restrict__ var, int dim, int ld, uint64_t seed, float mean,
                                              float stddev, int32_t* overflow, const KvFilter& flt) {
  if (i >= n) return;
  const int64_t key = ids[i];
  if (key < 0) return;  // ('' / padding: no row)
  uint64_t pos = kv_home(key, mask);
  if (flt.filter_freq > 1 || flt.version) {
    kv_insert_filtered(key, pos, mask, keys, rows, next_row, capacity, var, dim, ld, seed, mean, stddev, overflow, flt);
    return;
  }
  // A full arena claims no further key slots: ids that have a row are found, new ones only raise the (sticky) overflow
  // flag.  Otherwise every unseen id of every later step would occupy a slot for good (with row -1), the map's load
  // factor would pass 0.5 and probe sequences would grow towards the size of the map.  (The arena filling up between
  // this read and the CAS below leaves at most one launch's worth of such slots.)
  if (__hip_atomic_load(next_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= capacity) {
    for (uint64_t probes = 0; probes <= mask; ++probes, pos = (pos + 1) & mask) {
      const int64_t k = __hip_atomic_load(keys + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (k == key) return;
      if (k == kKvEmpty) break;
    }
    atomicExch(overflow, 1);
    return;
  }
  for (uint64_t probes = 0; probes <= mask; ++probes, pos = (pos + 1) & mask) {
    const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(keys + pos),
                                              static_cast<unsigned long long>(kKvEmpty), static_cast<unsigned long long>(key));
    if (static_cast<int64_t>(prev) == key) return;  // present (or being created by its winner)
    if (static_cast<int64_t>(prev) == kKvEmpty) {    // this lane created the key
      kv_create_row(key, pos, rows, next_row, capacity, var, dim, ld, seed, mean, stddev, overflow);
      return;
    }
  }
  atomicExch(overflow, 1);
}

__global__ void __launch_bounds__(kBlock)
kv_insert_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t* keys, int32_t* rows, uint64_t mask,
                 int32_t* next_row, int32_t capacity, float* __restrict__ var, int dim, int ld, uint64_t seed, float mean,
                 float stddev, int32_t* overflow, KvFilter flt) {
  kv_insert_one(static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x, ids, n, keys, rows, mask, next_row, capacity, var,
                dim, ld, seed, mean, stddev, overflow, flt);
}

__device__ __forceinline__ void kv_find_one(int64_t i, const int64_t* __restrict__ ids, int64_t n,
                                            const int64_t* __restrict__ keys, const int32_t* __restrict__ rows,
                                            uint64_t mask, int64_t* __restrict__ out) {
  if (i >= n) return;
  const int64_t key = ids[i];
  int64_t r = -1;
  if (key >= 0) {
    uint64_t pos = kv_home(key, mask);
    for (uint64_t probes = 0; probes <= mask; ++probes, pos = (pos + 1) & mask) {
      const int64_t k = keys[pos];
      if (k == key) {
        r = rows[pos];  // (-1: created past the capacity)
        break;
      }
      if (k == kKvEmpty) break;  // never seen: zero embedding (evaluation, or an id first met outside training)
    }
  }
  out[i] = r;
}

__global__ void __launch_bounds__(kBlock)
kv_find_kernel(const int64_t* __restrict__ ids, int64_t n, const int64_t* __restrict__ keys,
               const int32_t* __restrict__ rows, uint64_t mask, int64_t* __restrict__ out) {
  kv_find_one(static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x, ids, n, keys, rows, mask, out);
}

// All hash-table lookups of a model in one grid (a model with 26 hash-table features would otherwise pay 52 launches
// per step): workgroup b serves job j with blk_start[j] <= b < blk_start[j + 1] (descriptors in device memory).
__device__ __forceinline__ int kv_job_of(const int32_t* __restrict__ blk_start, int n_jobs, int b) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (blk_start[mid] <= b) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ void __launch_bounds__(kBlock)
kv_insert_multi_kernel(const er_kv_job* __restrict__ jobs, const int32_t* __restrict__ blk_start, int n_jobs) {
  const int j = kv_job_of(blk_start, n_jobs, blockIdx.x);
  const er_kv_job q = jobs[j];
  const int64_t n = q.n_limit ? (static_cast<int64_t>(*q.n_limit) < q.n ? static_cast<int64_t>(*q.n_limit) : q.n) : q.n;
  kv_insert_one(static_cast<int64_t>(blockIdx.x - blk_start[j]) * kBlock + threadIdx.x, q.ids, n, q.map_keys, q.map_rows,
                static_cast<uint64_t>(q.map_slots - 1), q.next_row, q.capacity, q.var, q.dim, q.table_ld ? q.table_ld : q.dim, q.seed, q.init_mean,
                q.init_stddev, q.overflow, KvFilter{q.freq, q.version, q.n_keys, q.step, q.filter_freq});
}

__global__ void __launch_bounds__(kBlock)
kv_find_multi_kernel(const er_kv_job* __restrict__ jobs, const int32_t* __restrict__ blk_start, int n_jobs) {
  const int j = kv_job_of(blk_start, n_jobs, blockIdx.x);
  const er_kv_job q = jobs[j];
  const int64_t n = q.n_limit ? (static_cast<int64_t>(*q.n_limit) < q.n ? static_cast<int64_t>(*q.n_limit) : q.n) : q.n;
  kv_find_one(static_cast<int64_t>(blockIdx.x - blk_start[j]) * kBlock + threadIdx.x, q.ids, n, q.map_keys, q.map_rows,
              static_cast<uint64_t>(q.map_slots - 1), q.rows_out);
}

// export: (key, row) of every occupied slot, compacted in slot order (the host sorts by key)
__global__ void __launch_bounds__(kBlock)
kv_export_kernel(const int64_t* __restrict__ keys, const int32_t* __restrict__ rows, int64_t slots,
                 int64_t* __restrict__ out_keys, int32_t* __restrict__ out_rows, int32_t* count) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= slots) return;
  if (keys[i] != kKvEmpty && rows[i] >= 0) {
    const int32_t p = atomicAdd(count, 1);
    out_keys[p] = keys[i];
    out_rows[p] = rows[i];
  }
}

// every occupied slot with its state (keys that have no row yet included: rows -1)
__global__ void __launch_bounds__(kBlock)
kv_export_all_kernel(const int64_t* __restrict__ keys, const int32_t* __restrict__ rows, const int32_t* __restrict__ freq,
                     const int32_t* __restrict__ version, int64_t slots, int64_t* __restrict__ out_keys,
                     int32_t* __restrict__ out_rows, int32_t* __restrict__ out_freq, int32_t* __restrict__ out_version,
                     int32_t* count) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= slots) return;
  if (keys[i] != kKvEmpty) {
    const int32_t p = atomicAdd(count, 1);
    out_keys[p] = keys[i];
    out_rows[p] = rows[i];
    out_freq[p] = freq ? freq[i] : 0;
    out_version[p] = version ? version[i] : 0;
  }
}

// (key, row, freq, version) records into a CLEARED map (distinct keys)
__global__ void __launch_bounds__(kBlock)
kv_rebuild_kernel(const int64_t* __restrict__ in_keys, const int32_t* __restrict__ in_rows,
                  const int32_t* __restrict__ in_freq, const int32_t* __restrict__ in_version, int64_t n, int64_t* keys,
                  int32_t* rows, int32_t* freq, int32_t* version, uint64_t mask, int32_t* overflow) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t key = in_keys[i];
  uint64_t pos = kv_home(key, mask);
  for (uint64_t probes = 0; probes <= mask; ++probes, pos = (pos + 1) & mask) {
    const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(keys + pos),
                                              static_cast<unsigned long long>(kKvEmpty), static_cast<unsigned long long>(key));
    if (static_cast<int64_t>(prev) == kKvEmpty) {
      rows[pos] = in_rows[i];
      if (freq) freq[pos] = in_freq ? in_freq[i] : 0;
      if (version) version[pos] = in_version ? in_version[i] : 0;
      return;
    }
    if (static_cast<int64_t>(prev) == key) break;  // (a repeated key: refused)
  }
  atomicExch(overflow, 1);
}

// ---- embedding-parallel hash tables: the ids travel to their owners before the route ---------------------------------
// Under the reference's embedding parallelism a hash-table column is an SOK DynamicVariable sharded by id % world
// (compat/feature_column/feature_column.py:470-503, compat/dynamic_variable.py).  Here every rank owns the map and the
// arena of the ids with id % world == rank; a step's ids go to their owners (fixed-capacity all-to-all), the owners
// translate them (the insert / find launches above, counting filter and stamps included) and send the arena rows back,
// and the requester turns (owner, arena row) into the VIRTUAL dense id arena_row * world + owner - which the ordinary
// embedding-parallel route sends to the same owner (id % world) and local row (id / world).  Everything after this
// pre-pass is the dense sharded path unchanged.
//   bucket  : send[owner][job region] <- the job's ids of that owner in arrival order (slot[i] remembers where)
//   unbucket: rows_out[i] = back[slot[i]] * world + owner, or -1
__global__ void __launch_bounds__(kBlock)
kv_bucket_clear_kernel(int64_t* __restrict__ send, int64_t send_n, int32_t* __restrict__ counts, int64_t counts_n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < send_n; i += stride) send[i] = -1;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < counts_n; i += stride) counts[i] = 0;
}

__global__ void __launch_bounds__(kBlock)
kv_bucket_kernel(const er_kv_route_job* __restrict__ jobs, const int32_t* __restrict__ blk_start, int n_jobs, int world,
                 int64_t block_stride, int64_t* __restrict__ send, int32_t* __restrict__ counts) {
  const int j = kv_job_of(blk_start, n_jobs, blockIdx.x);
  const er_kv_route_job q = jobs[j];
  const int64_t i = static_cast<int64_t>(blockIdx.x - blk_start[j]) * kBlock + threadIdx.x;
  if (i >= q.n) return;
  const int64_t n = q.n_limit ? (static_cast<int64_t>(*q.n_limit) < q.n ? static_cast<int64_t>(*q.n_limit) : q.n) : q.n;
  const int64_t key = i < n ? q.ids[i] : -1;
  if (key < 0) {
    q.slot[i] = -1;
    return;
  }
  const int owner = static_cast<int>(static_cast<uint64_t>(key) % static_cast<uint64_t>(world));
  const int32_t s = atomicAdd(counts + static_cast<int64_t>(j) * world + owner, 1);  // (< q.n: the region holds the whole job)
  const int64_t at = static_cast<int64_t>(owner) * block_stride + q.send_off + s;
  send[at] = key;
  q.slot[i] = static_cast<int32_t>(at);
}

__global__ void __launch_bounds__(kBlock)
kv_unbucket_kernel(const er_kv_route_job* __restrict__ jobs, const int32_t* __restrict__ blk_start, int n_jobs, int world,
                   int64_t block_stride, const int64_t* __restrict__ back) {
  const int j = kv_job_of(blk_start, n_jobs, blockIdx.x);
  const er_kv_route_job q = jobs[j];
  const int64_t i = static_cast<int64_t>(blockIdx.x - blk_start[j]) * kBlock + threadIdx.x;
  if (i >= q.n) return;
  const int32_t at = q.slot[i];
  int64_t v = -1;
  if (at >= 0) {
    const int64_t r = back[at];
    if (r >= 0) v = r * world + at / block_stride;
  }
  q.rows_out[i] = v;
}

}  // namespace er

extern "C" {

int er_kv_translate(const int64_t* ids, int64_t n, int64_t* map_keys, int32_t* map_rows, int64_t map_slots,
                    int32_t* next_row, int32_t capacity, float* var, int32_t dim, uint64_t seed, float init_mean,
                    float init_stddev, int insert, int64_t* rows_out, int32_t* overflow, er_stream_t stream) {
  ER_REQUIRE(ids && map_keys && map_rows && next_row && rows_out && overflow && n >= 0, "er_kv_translate: null argument");
  ER_REQUIRE(map_slots >= 2 && (map_slots & (map_slots - 1)) == 0, "er_kv_translate: map_slots must be a power of two");
  ER_REQUIRE(capacity > 0 && static_cast<int64_t>(capacity) * 2 <= map_slots && dim > 0 && (var || !insert),
             "er_kv_translate: capacity must be <= map_slots / 2");
  if (n == 0) return 0;
  hipStream_t s = er::as_stream(stream);
  const uint64_t mask = static_cast<uint64_t>(map_slots - 1);
  if (insert) {
    hipLaunchKernelGGL(er::kv_insert_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, s, ids, n, map_keys, map_rows,
                       mask, next_row, capacity, var, dim, dim, seed, init_mean, init_stddev, overflow,
                       er::KvFilter{nullptr, nullptr, nullptr, nullptr, 0});
    ER_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(er::kv_find_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, s, ids, n, map_keys, map_rows, mask,
                     rows_out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_kv_translate_multi(const er_kv_job* jobs_dev, const int32_t* blk_start_dev, int n_jobs, int total_blocks, int insert,
                          er_stream_t stream) {
  ER_REQUIRE(jobs_dev && blk_start_dev && n_jobs >= 1 && total_blocks >= 1, "er_kv_translate_multi: bad arguments");
  hipStream_t s = er::as_stream(stream);
  if (insert) {
    hipLaunchKernelGGL(er::kv_insert_multi_kernel, dim3(total_blocks), dim3(er::kBlock), 0, s, jobs_dev, blk_start_dev, n_jobs);
    ER_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(er::kv_find_multi_kernel, dim3(total_blocks), dim3(er::kBlock), 0, s, jobs_dev, blk_start_dev, n_jobs);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_kv_translate_job(const er_kv_job* job, int insert, er_stream_t stream) {
  ER_REQUIRE(job && job->ids && job->map_keys && job->map_rows && job->next_row && job->rows_out && job->overflow && job->n >= 0,
             "er_kv_translate_job: null argument");
  ER_REQUIRE(job->map_slots >= 2 && (job->map_slots & (job->map_slots - 1)) == 0 && job->capacity > 0 &&
                 static_cast<int64_t>(job->capacity) * 2 <= job->map_slots && job->dim > 0 && (job->var || !insert),
             "er_kv_translate_job: map_slots must be a power of two >= 2 * capacity");
  ER_REQUIRE(job->filter_freq <= 1 || (job->freq && job->n_keys), "er_kv_translate_job: filter_freq needs freq and n_keys");
  ER_REQUIRE(!job->n_limit, "er_kv_translate_job: n_limit is for the device-resident job table");
  if (job->n == 0) return 0;
  hipStream_t s = er::as_stream(stream);
  const uint64_t mask = static_cast<uint64_t>(job->map_slots - 1);
  if (insert) {
    hipLaunchKernelGGL(er::kv_insert_kernel, dim3(er::blocks_for(job->n)), dim3(er::kBlock), 0, s, job->ids, job->n,
                       job->map_keys, job->map_rows, mask, job->next_row, job->capacity, job->var, job->dim, job->table_ld ? job->table_ld : job->dim,
                       job->seed,
                       job->init_mean, job->init_stddev, job->overflow,
                       er::KvFilter{job->freq, job->version, job->n_keys, job->step, job->filter_freq});
    ER_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(er::kv_find_kernel, dim3(er::blocks_for(job->n)), dim3(er::kBlock), 0, s, job->ids, job->n,
                     job->map_keys, job->map_rows, mask, job->rows_out);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_kv_export_all(const int64_t* map_keys, const int32_t* map_rows, const int32_t* map_freq, const int32_t* map_version,
                     int64_t map_slots, int64_t* out_keys, int32_t* out_rows, int32_t* out_freq, int32_t* out_version,
                     int32_t* count, er_stream_t stream) {
  ER_REQUIRE(map_keys && map_rows && out_keys && out_rows && out_freq && out_version && count && map_slots > 0,
             "er_kv_export_all: null argument");
  hipLaunchKernelGGL(er::kv_export_all_kernel, dim3(er::blocks_for(map_slots)), dim3(er::kBlock), 0, er::as_stream(stream),
                     map_keys, map_rows, map_freq, map_version, map_slots, out_keys, out_rows, out_freq, out_version, count);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_kv_rebuild(const int64_t* keys, const int32_t* rows, const int32_t* freq, const int32_t* version, int64_t n,
                  int64_t* map_keys, int32_t* map_rows, int32_t* map_freq, int32_t* map_version, int64_t map_slots,
                  int32_t* overflow, er_stream_t stream) {
  ER_REQUIRE(map_keys && map_rows && overflow && n >= 0 && (n == 0 || (keys && rows)), "er_kv_rebuild: null argument");
  ER_REQUIRE(map_slots >= 2 && (map_slots & (map_slots - 1)) == 0 && n * 2 <= map_slots,
             "er_kv_rebuild: map_slots must be a power of two >= 2 * n");
  if (n == 0) return 0;
  hipLaunchKernelGGL(er::kv_rebuild_kernel, dim3(er::blocks_for(n)), dim3(er::kBlock), 0, er::as_stream(stream), keys, rows,
                     freq, version, n, map_keys, map_rows, map_freq, map_version, static_cast<uint64_t>(map_slots - 1), overflow);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_kv_bucket(const er_kv_route_job* jobs_dev, const int32_t* blk_start_dev, int n_jobs, int total_blocks, int world,
                 int64_t block_stride, int64_t* send, int32_t* counts, er_stream_t stream) {
  ER_REQUIRE(jobs_dev && blk_start_dev && send && counts && n_jobs >= 1 && total_blocks >= 1 && world >= 1 && block_stride >= 1 &&
                 block_stride * world < (1ll << 31),
             "er_kv_bucket: bad arguments (the send buffer is addressed with 31 bits)");
  hipStream_t s = er::as_stream(stream);
  hipLaunchKernelGGL(er::kv_bucket_clear_kernel, dim3(256), dim3(er::kBlock), 0, s, send, block_stride * world, counts,
                     static_cast<int64_t>(n_jobs) * world);
  ER_LAUNCH_CHECK();
  hipLaunchKernelGGL(er::kv_bucket_kernel, dim3(total_blocks), dim3(er::kBlock), 0, s, jobs_dev, blk_start_dev, n_jobs, world,
                     block_stride, send, counts);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_kv_unbucket(const er_kv_route_job* jobs_dev, const int32_t* blk_start_dev, int n_jobs, int total_blocks, int world,
                   int64_t block_stride, const int64_t* back, er_stream_t stream) {
  ER_REQUIRE(jobs_dev && blk_start_dev && back && n_jobs >= 1 && total_blocks >= 1 && world >= 1 && block_stride >= 1,
             "er_kv_unbucket: bad arguments");
  hipLaunchKernelGGL(er::kv_unbucket_kernel, dim3(total_blocks), dim3(er::kBlock), 0, er::as_stream(stream), jobs_dev,
                     blk_start_dev, n_jobs, world, block_stride, back);
  ER_LAUNCH_CHECK();
  return 0;
}

int er_kv_export(const int64_t* map_keys, const int32_t* map_rows, int64_t map_slots, int64_t* out_keys,
                 int32_t* out_rows, int32_t* count, er_stream_t stream) {
  ER_REQUIRE(map_keys && map_rows && out_keys && out_rows && count && map_slots > 0, "er_kv_export: null argument");
  hipLaunchKernelGGL(er::kv_export_kernel, dim3(er::blocks_for(map_slots)), dim3(er::kBlock), 0, er::as_stream(stream),
                     map_keys, map_rows, map_slots, out_keys, out_rows, count);
  ER_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
