// Sharded embedding checkpoint files: host-side C++ equivalents of the reference's two native ops.
//
//   easy_rec/python/ops/src/load_dense_embed.cc:54-135 (LoadEmbedOp) - re-shard `embed-<var>-part-<k>.bin` files
//     written by ANY number of workers onto (task_index, task_num): the file of old worker k holds rows
//     k, k + P, k + 2 P, ... (P = number of part files); new worker r keeps the rows with id % task_num == r at
//     local position id / task_num.
//   easy_rec/python/ops/src/load_kv_embed.cc:60-163 (LoadKVEmbedOp) - the same for key/value tables
//     (`.key` int64 + `.val` float32 files): a worker keeps the keys with key mod task_num == task_index.
//   easy_rec/python/compat/embedding_parallel_saver.py:99-127 (_save_dense_embedding) - the writer: raw float32
//     rows of the worker's shard; worker 0 removes part files of workers that no longer exist.
//
// No device code: checkpoints are read into host (pinned) buffers the caller uploads.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "er_common.h"

namespace {

// embed-input_layer__all_fea__embedding_weights:0-part-12.bin -> 12   (load_dense_embed.cc:40-50)
int part_id_of(const std::string& path) {
  if (path.size() < 6) return -1;
  const size_t pos = path.rfind('-', path.size() - 5);
  if (pos == std::string::npos) return -1;
  return std::atoi(path.c_str() + pos + 1);
}

bool ends_with(const std::string& s, const char* suffix) {
  const size_t n = std::strlen(suffix);
  return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// regular files `<folder>/<prefix><id><suffix>`, sorted by part id
int list_parts(const std::string& folder, const std::string& prefix, const char* suffix,
               std::vector<std::string>* out) {
  DIR* dir = opendir(folder.c_str());
  if (!dir) return 1;
  while (dirent* ent = readdir(dir)) {
    const std::string name = ent->d_name;
    if (name.compare(0, prefix.size(), prefix) != 0 || !ends_with(name, suffix)) continue;
    struct stat st;
    const std::string path = folder + name;
    if (stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode)) out->push_back(path);
  }
  closedir(dir);
  std::sort(out->begin(), out->end(),
            [](const std::string& a, const std::string& b) { return part_id_of(a) < part_id_of(b); });
  return 0;
}

bool read_all(const std::string& path, std::vector<char>* buf) {
  std::ifstream fin(path.c_str(), std::ios::binary);
  if (!fin) return false;
  fin.seekg(0, fin.end);
  const std::streamoff len = fin.tellg();
  fin.seekg(0, fin.beg);
  buf->resize(static_cast<size_t>(len));
  if (len > 0) fin.read(buf->data(), len);
  return static_cast<bool>(fin) || fin.eof();
}

int make_dir(const std::string& d) {
  struct stat st;
  if (stat(d.c_str(), &st) == 0) return S_ISDIR(st.st_mode) ? 0 : 1;
  return mkdir(d.c_str(), 0755) == 0 ? 0 : (stat(d.c_str(), &st) == 0 && S_ISDIR(st.st_mode) ? 0 : 1);
}

}  // namespace

extern "C" {

int er_save_dense_embed(const char* ckpt_path, const char* var_name, int32_t task_index, int32_t task_num,
                        const float* vals, int64_t rows, int32_t embed_dim) {
  ER_REQUIRE(ckpt_path && var_name && vals && rows >= 0 && embed_dim > 0 && task_index >= 0 && task_index < task_num,
             "er_save_dense_embed: bad arguments");
  const std::string folder = std::string(ckpt_path) + "-embedding/";
  ER_REQUIRE(make_dir(folder) == 0, "er_save_dense_embed: cannot create %s", folder.c_str());
  const std::string prefix = std::string(var_name) + "-part-";
  const std::string path = folder + prefix + std::to_string(task_index) + ".bin";
  {
    std::ofstream fout(path.c_str(), std::ios::binary | std::ios::trunc);
    ER_REQUIRE(static_cast<bool>(fout), "er_save_dense_embed: cannot open %s", path.c_str());
    fout.write(reinterpret_cast<const char*>(vals), static_cast<std::streamsize>(sizeof(float)) * rows * embed_dim);
    ER_REQUIRE(static_cast<bool>(fout), "er_save_dense_embed: short write to %s", path.c_str());
  }
  if (task_index == 0) {  // clear the tables of workers that no longer exist (embedding_parallel_saver.py:115-122)
    std::vector<std::string> parts;
    if (list_parts(folder, prefix, ".bin", &parts) == 0)
      for (const auto& p : parts)
        if (part_id_of(p) >= task_num) std::remove(p.c_str());
  }
  return 0;
}

int er_load_dense_embed(const char* ckpt_path, const char* var_name, int32_t task_index, int32_t task_num,
                        int32_t embed_dim, int64_t embed_part_size, float* out_vals, int64_t* rows_loaded) {
  ER_REQUIRE(ckpt_path && var_name && out_vals && embed_dim > 0 && embed_part_size > 0 && task_index >= 0 &&
                 task_index < task_num,
             "er_load_dense_embed: bad arguments");
  const std::string folder = std::string(ckpt_path) + "-embedding/";
  std::vector<std::string> parts;
  ER_REQUIRE(list_parts(folder, std::string(var_name) + "-part-", ".bin", &parts) == 0,
             "er_load_dense_embed: cannot list %s", folder.c_str());
  ER_REQUIRE(!parts.empty(), "er_load_dense_embed: no %s-part-*.bin under %s", var_name, folder.c_str());
  // the reference leaves rows it does not find uninitialised except the last one (load_dense_embed.cc:88,127-130):
  // every row is cleared here
  std::memset(out_vals, 0, sizeof(float) * static_cast<size_t>(embed_part_size) * embed_dim);
  const int64_t total = embed_part_size * task_num;
  const int64_t n_parts = static_cast<int64_t>(parts.size());
  int64_t updated = 0;
  std::vector<char> buf;
  for (const auto& path : parts) {
    ER_REQUIRE(read_all(path, &buf), "er_load_dense_embed: cannot read %s", path.c_str());
    const int64_t part_id_o = part_id_of(path);
    ER_REQUIRE(part_id_o >= 0 && part_id_o < n_parts, "er_load_dense_embed: part id of %s outside the %lld parts found",
               path.c_str(), (long long)n_parts);
    const float* src = reinterpret_cast<const float*>(buf.data());
    const int64_t n_o = static_cast<int64_t>(buf.size() / sizeof(float)) / embed_dim;
    for (int64_t i = 0; i < n_o; ++i) {
      const int64_t id = i * n_parts + part_id_o;  // the global row this old local row is
      if (id % task_num == task_index && id < total) {
        std::memcpy(out_vals + (id / task_num) * embed_dim, src + i * embed_dim, sizeof(float) * embed_dim);
        ++updated;
      }
    }
  }
  if (rows_loaded) *rows_loaded = updated;
  // load_dense_embed.cc:121-126: the old shards must cover this shard, up to the one padding row at its end
  ER_REQUIRE(updated == embed_part_size || updated + 1 == embed_part_size,
             "er_load_dense_embed: %s: %lld rows found for a shard of %lld", var_name, (long long)updated,
             (long long)embed_part_size);
  return 0;
}

int er_load_kv_embed(const char* ckpt_path, const char* var_name, int32_t task_index, int32_t task_num,
                     int32_t embed_dim, int64_t capacity, int64_t* out_keys, float* out_vals, int64_t* n_keys) {
  ER_REQUIRE(ckpt_path && var_name && n_keys && embed_dim > 0 && task_index >= 0 && task_index < task_num,
             "er_load_kv_embed: bad arguments");
  const std::string folder = std::string(ckpt_path) + "-embedding/";
  std::vector<std::string> key_files;
  ER_REQUIRE(list_parts(folder, std::string(var_name) + "-part-", ".key", &key_files) == 0,
             "er_load_kv_embed: cannot list %s", folder.c_str());
  int64_t count = 0;
  std::vector<char> kbuf, vbuf;
  for (const auto& kf : key_files) {
    ER_REQUIRE(read_all(kf, &kbuf), "er_load_kv_embed: cannot read %s", kf.c_str());
    const int64_t n = static_cast<int64_t>(kbuf.size() / sizeof(int64_t));
    const int64_t* keys = reinterpret_cast<const int64_t*>(kbuf.data());
    const float* vals = nullptr;
    if (out_keys) {
      const std::string vf = kf.substr(0, kf.size() - 4) + ".val";
      ER_REQUIRE(read_all(vf, &vbuf), "er_load_kv_embed: cannot read %s", vf.c_str());
      ER_REQUIRE(static_cast<int64_t>(vbuf.size()) == n * embed_dim * static_cast<int64_t>(sizeof(float)),
                 "er_load_kv_embed: key_num(%lld) does not match the size of %s", (long long)n, vf.c_str());
      vals = reinterpret_cast<const float*>(vbuf.data());
    }
    for (int64_t j = 0; j < n; ++j) {
      int64_t a = keys[j] % task_num;  // load_kv_embed.cc:123-127
      if (a < 0) a += task_num;
      if (a != task_index) continue;
      if (out_keys) {
        ER_REQUIRE(count < capacity, "er_load_kv_embed: more than %lld keys for this worker", (long long)capacity);
        out_keys[count] = keys[j];
        std::memcpy(out_vals + count * embed_dim, vals + j * embed_dim, sizeof(float) * embed_dim);
      }
      ++count;
    }
  }
  *n_keys = count;
  return 0;
}


// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of n bytes, continuing from `crc` (0 to start): the checksum of
// TensorFlow's tensor-bundle entries and table blocks (tensorflow/core/lib/hash/crc32c.h; the files a TF Saver writes for
// the dense variables, reference model/easy_rec_model.py:219-351 restores from them) - easyrec_amd/utils/tensor_bundle.py.
uint32_t er_crc32c(uint32_t crc, const void* data, int64_t n) {
  // (a function-local static with a constructor: initialised once, thread-safely, on first use - the first calls may come
  // from two threads at once, e.g. an asynchronous checkpoint writer next to a reader)
  struct Tables {
    uint32_t t[8][256];
    Tables() {
      for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        t[0][i] = c;
      }
      for (uint32_t i = 0; i < 256; ++i)
        for (int q = 1; q < 8; ++q) t[q][i] = (t[q - 1][i] >> 8) ^ t[0][t[q - 1][i] & 0xFFu];
    }
  };
  static const Tables tables;
  const uint32_t (*table)[256] = tables.t;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = ~crc;
  while (n >= 8) {  // slicing-by-8
    const uint32_t lo = c ^ (static_cast<uint32_t>(p[0]) | static_cast<uint32_t>(p[1]) << 8 | static_cast<uint32_t>(p[2]) << 16 |
                             static_cast<uint32_t>(p[3]) << 24);
    c = table[7][lo & 0xFFu] ^ table[6][(lo >> 8) & 0xFFu] ^ table[5][(lo >> 16) & 0xFFu] ^ table[4][lo >> 24] ^
        table[3][p[4]] ^ table[2][p[5]] ^ table[1][p[6]] ^ table[0][p[7]];
    p += 8;
    n -= 8;
  }
  while (n-- > 0) c = table[0][(c ^ *p++) & 0xFFu] ^ (c >> 8);
  return ~c;
}

}  // extern "C"
