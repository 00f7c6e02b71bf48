// Host-side decoder of separator-delimited text: the C++ half of CSVInput.
//
// Reference: easy_rec/python/input/csv_input.py:33-76 (`tf.decode_csv(line, field_delim=separator,
// record_defaults=...)`, a TensorFlow C++ kernel) - every line is split on ONE separator character into exactly
// n_fields cells; an empty cell takes the field's default (applied by the caller: utils/input_utils.py:11-36).
// Here one pass over the text of a batch writes column-major results: parsed int64 / double values for numeric
// fields, (begin, length) of the cell inside the text for string fields (no copy - the strings are packed later,
// straight from the text buffer), and an "empty cell" mask.  No device code.
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "er_common.h"

namespace {

// Host worker threads that outlive a call: creating and joining a thread costs 20 - 50 us on a plain Linux host and
// 0.3 - 0.5 ms inside a sandboxed container - per thread and per 4096-line batch that is the whole gain.  One job at a
// time (callers are serialised); the pool is never destroyed (no destructor races at exit) and is rebuilt after a fork
// (a child process has none of the parent's threads).
class HostPool {
 public:
  static HostPool* get() {
    // lock-free (a mutex here could be inherited LOCKED by a child forked while another thread held it): the pool of this
    // process, replaced after a fork - the parent's worker threads do not exist in the child; the old object is leaked on
    // purpose
    static std::atomic<HostPool*> pool{nullptr};
    HostPool* p = pool.load(std::memory_order_acquire);
    if (!p || p->pid_ != getpid()) {
      HostPool* fresh = new HostPool();
      if (pool.compare_exchange_strong(p, fresh, std::memory_order_acq_rel)) {
        p = fresh;
      } else {
        delete fresh;  // (another thread of this process installed one first; p holds it)
        if (p->pid_ != getpid()) p = pool.load(std::memory_order_acquire);
      }
    }
    return p;
  }
  // fn(0 .. n - 1), fn(0) on the calling thread
  void run(int n, const std::function<void(int)>& fn) {
    std::lock_guard<std::mutex> serial(run_mu_);
    if (n <= 1) {
      fn(0);
      return;
    }
    {
      std::unique_lock<std::mutex> lock(mu_);
      while (static_cast<int>(workers_.size()) < n - 1) workers_.emplace_back([this] { loop(); });
      fn_ = &fn;
      n_ = n;
      next_ = 1;
      pending_ = n - 1;
      ++epoch_;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lock(mu_);
    done_cv_.wait(lock, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  HostPool() : pid_(getpid()) {}
  void loop() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lock(mu_);
    for (;;) {
      cv_.wait(lock, [&] { return epoch_ != seen; });
      seen = epoch_;
      while (fn_ && next_ < n_) {
        const int i = next_++;
        const std::function<void(int)>* fn = fn_;
        lock.unlock();
        (*fn)(i);
        lock.lock();
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  pid_t pid_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> workers_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0, next_ = 0, pending_ = 0;
  uint64_t epoch_ = 0;
};

// A decimal integer as strtoll reads it when nothing unusual is in the cell: [+-] 1 - 18 digits.  Anything else (spaces,
// 19+ digits, other characters) is left to strtoll itself - the results and the error behaviour stay its.
inline bool parse_int_fast(const uint8_t* p, int64_t len, int64_t* out) {
  int64_t i = 0;
  bool neg = false;
  if (p[0] == '-' || p[0] == '+') {
    neg = p[0] == '-';
    i = 1;
  }
  if (i == len || len - i > 18) return false;
  uint64_t v = 0;
  for (; i < len; ++i) {
    const unsigned d = static_cast<unsigned>(p[i]) - '0';
    if (d > 9) return false;
    v = v * 10 + d;
  }
  *out = neg ? -static_cast<int64_t>(v) : static_cast<int64_t>(v);
  return true;
}

// [+-] digits [. digits] with at most 15 significant digits and at most 22 fraction digits: mantissa and 10^k are both
// exact doubles, so ONE IEEE division is the correctly rounded value - what glibc's strtod returns (Clinger's fast path).
// Exponents, inf / nan, hex floats, longer mantissas: strtod.
inline bool parse_double_fast(const uint8_t* p, int64_t len, double* out) {
  static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  int64_t i = 0;
  bool neg = false;
  if (p[0] == '-' || p[0] == '+') {
    neg = p[0] == '-';
    i = 1;
  }
  uint64_t m = 0;
  int digits = 0, frac = 0, sig = 0;
  bool seen_dot = false;
  for (; i < len; ++i) {
    const uint8_t c = p[i];
    if (c == '.') {
      if (seen_dot) return false;
      seen_dot = true;
      continue;
    }
    const unsigned d = static_cast<unsigned>(c) - '0';
    if (d > 9) return false;
    ++digits;
    if (m != 0 || d != 0) ++sig;
    if (sig > 15) return false;
    m = m * 10 + d;
    if (seen_dot) ++frac;
  }
  if (digits == 0 || frac > 22) return false;
  const double v = static_cast<double>(m) / kPow10[frac];
  *out = neg ? -v : v;
  return true;
}

struct CsvLine { int64_t b, e; };  // [b, e): the line without its "\n" / "\r\n"

// rows [r0, r1) of `lines` into the column-major outputs; the first failure (if any) as (row, message)
void decode_rows(const uint8_t* text, const CsvLine* lines, int64_t r0, int64_t r1, uint8_t sep, int32_t n_fields,
                 const int32_t* kinds, int64_t out_stride, int64_t* int_out, double* flt_out, uint8_t* empty_out,
                 int64_t* str_begin, int32_t* str_len, int64_t* bad_row, std::string* bad_msg) {
  char num[64], msg[256];
  for (int64_t row = r0; row < r1; ++row) {
    int64_t b = lines[row].b;
    const int64_t eol = lines[row].e;
    for (int32_t f = 0; f < n_fields; ++f) {
      const uint8_t* hit = b < eol ? static_cast<const uint8_t*>(memchr(text + b, sep, static_cast<size_t>(eol - b))) : nullptr;
      const int64_t e = hit ? hit - text : eol;
      if (!(f + 1 < n_fields ? e < eol : e == eol)) {
        snprintf(msg, sizeof(msg), "er_decode_csv_host: line %lld has %s than %d fields", (long long)row,
                 f + 1 < n_fields ? "fewer" : "more", n_fields);
        *bad_row = row;
        *bad_msg = msg;
        return;
      }
      const int64_t o = static_cast<int64_t>(f) * out_stride + row;
      const int64_t len = e - b;
      empty_out[o] = len == 0;
      str_begin[o] = b;
      str_len[o] = static_cast<int32_t>(len);
      int_out[o] = 0;
      flt_out[o] = 0.0;
      if (len > 0 && kinds[f] != 0) {
        bool ok = false;
        if (kinds[f] == 1) {
          ok = parse_int_fast(text + b, len, &int_out[o]);
          if (ok) flt_out[o] = static_cast<double>(int_out[o]);
        } else {
          ok = parse_double_fast(text + b, len, &flt_out[o]);
        }
        if (!ok) {  // the general path: strtoll / strtod on a terminated copy
          if (len >= static_cast<int64_t>(sizeof(num))) {
            snprintf(msg, sizeof(msg), "er_decode_csv_host: line %lld field %d: number too long", (long long)row, f);
            *bad_row = row;
            *bad_msg = msg;
            return;
          }
          memcpy(num, text + b, static_cast<size_t>(len));
          num[len] = 0;
          char* endp = nullptr;
          errno = 0;
          if (kinds[f] == 1) {
            int_out[o] = strtoll(num, &endp, 10);
            flt_out[o] = static_cast<double>(int_out[o]);
          } else {
            flt_out[o] = strtod(num, &endp);
          }
          if (!(endp == num + len && errno == 0)) {
            snprintf(msg, sizeof(msg), "er_decode_csv_host: line %lld field %d: '%s' is not a number", (long long)row, f, num);
            *bad_row = row;
            *bad_msg = msg;
            return;
          }
        }
      }
      b = e + 1;
    }
  }
}

}  // namespace

extern "C" {

// er_decode_csv_host on `n_threads` host threads (<= 0: one per hardware thread, at most 8; a thread takes at least 256
// rows).  out_stride (>= max_rows): elements between consecutive fields of the column-major outputs - NOT a power of two
// where it matters: a row's 40 fields x 5 arrays at a 4096-element pitch all map to the same cache sets (one 4096-line
// batch: 3.2 ms at pitch 4096, 2.4 at 4100 on one thread; 0.9 against 0.7 on eight).  One pass finds the lines (memchr: a few GB/s), the threads parse disjoint row ranges into the same column-major
// outputs; numbers take an inline fast path (a plain decimal integer; a decimal of <= 15 significant digits = one exact
// division, the value glibc's strtod returns) and strtoll / strtod otherwise.  Results and errors: er_decode_csv_host's.
int er_decode_csv_host_mt(const uint8_t* text, int64_t n_bytes, uint8_t sep, int32_t n_fields, const int32_t* kinds,
                          int64_t max_rows, int64_t out_stride, int64_t* int_out, double* flt_out, uint8_t* empty_out,
                          int64_t* str_begin, int32_t* str_len, int64_t* n_rows_out, int64_t* consumed_out, int32_t n_threads) {
  ER_REQUIRE(text && kinds && int_out && flt_out && empty_out && str_begin && str_len && n_rows_out && consumed_out &&
                 n_bytes >= 0 && n_fields > 0 && max_rows >= 0 && out_stride >= max_rows && sep != '\n' && sep != '\r',
             "er_decode_csv_host: bad arguments");
  std::vector<CsvLine> lines;
  lines.reserve(static_cast<size_t>(max_rows < (1 << 20) ? max_rows : (1 << 20)));
  int64_t pos = 0;
  while (pos < n_bytes && static_cast<int64_t>(lines.size()) < max_rows) {
    const uint8_t* nl = static_cast<const uint8_t*>(memchr(text + pos, '\n', static_cast<size_t>(n_bytes - pos)));
    if (!nl) break;  // an unterminated last line is left to the caller (it may continue in the next chunk)
    int64_t eol = nl - text;
    const int64_t next = eol + 1;
    if (eol > pos && text[eol - 1] == '\r') --eol;
    if (eol > pos) lines.push_back(CsvLine{pos, eol});  // (a blank line is skipped)
    pos = next;
  }
  const int64_t rows = static_cast<int64_t>(lines.size());
  int T = n_threads;
  if (T <= 0) {
    // (at most 8: the decode of a 4096-line batch stops scaling there, several reader threads of one process serialise on
    // the pool, and a multi-process loader would otherwise put 16 workers per reader on the host)
    T = static_cast<int>(std::thread::hardware_concurrency());
    if (T > 8) T = 8;
  }
  if (T > rows / 256) T = static_cast<int>(rows / 256);
  if (T < 1) T = 1;
  std::vector<int64_t> bad_row(static_cast<size_t>(T), -1);
  std::vector<std::string> bad_msg(static_cast<size_t>(T));
  const int64_t per = (rows + T - 1) / T;
  auto work = [&](int t) {
    const int64_t r0 = t * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    if (r0 < r1)
      decode_rows(text, lines.data(), r0, r1, sep, n_fields, kinds, out_stride, int_out, flt_out, empty_out, str_begin, str_len,
                  &bad_row[t], &bad_msg[t]);
  };
  if (T == 1) {
    work(0);
  } else {
    HostPool::get()->run(T, work);
  }
  for (int t = 0; t < T; ++t)  // the ranges ascend: the first thread with a failure holds the smallest row
    if (bad_row[t] >= 0) {
      er::set_error("%s", bad_msg[t].c_str());
      return 2;
    }
  *n_rows_out = rows;
  *consumed_out = pos;
  return 0;
}

int er_decode_csv_host(const uint8_t* text, int64_t n_bytes, uint8_t sep, int32_t n_fields, const int32_t* kinds,
                       int64_t max_rows, int64_t* int_out, double* flt_out, uint8_t* empty_out, int64_t* str_begin,
                       int32_t* str_len, int64_t* n_rows_out, int64_t* consumed_out) {
  ER_REQUIRE(text && kinds && int_out && flt_out && empty_out && str_begin && str_len && n_rows_out && consumed_out &&
                 n_bytes >= 0 && n_fields > 0 && max_rows >= 0 && sep != '\n' && sep != '\r',
             "er_decode_csv_host: bad arguments");
  int64_t pos = 0, row = 0;
  char num[64];
  while (pos < n_bytes && row < max_rows) {
    // one line: [pos, eol), without the trailing "\r"
    const uint8_t* nl = static_cast<const uint8_t*>(memchr(text + pos, '\n', static_cast<size_t>(n_bytes - pos)));
    if (!nl) break;  // an unterminated last line is left to the caller (it may continue in the next chunk)
    int64_t eol = nl - text;
    const int64_t next = eol + 1;
    if (eol > pos && text[eol - 1] == '\r') --eol;
    if (eol == pos) {  // blank line
      pos = next;
      continue;
    }
    int64_t b = pos;
    for (int32_t f = 0; f < n_fields; ++f) {
      int64_t e = b;
      while (e < eol && text[e] != sep) ++e;
      ER_REQUIRE(f + 1 < n_fields ? e < eol : e == eol, "er_decode_csv_host: line %lld has %s than %d fields",
                 (long long)row, f + 1 < n_fields ? "fewer" : "more", n_fields);
      const int64_t o = static_cast<int64_t>(f) * max_rows + row;
      const int64_t len = e - b;
      empty_out[o] = len == 0;
      str_begin[o] = b;
      str_len[o] = static_cast<int32_t>(len);
      int_out[o] = 0;
      flt_out[o] = 0.0;
      if (len > 0 && kinds[f] != 0) {
        ER_REQUIRE(len < static_cast<int64_t>(sizeof(num)), "er_decode_csv_host: line %lld field %d: number too long",
                   (long long)row, f);
        memcpy(num, text + b, static_cast<size_t>(len));
        num[len] = 0;
        char* endp = nullptr;
        errno = 0;
        if (kinds[f] == 1) {
          int_out[o] = strtoll(num, &endp, 10);
          flt_out[o] = static_cast<double>(int_out[o]);
        } else {
          flt_out[o] = strtod(num, &endp);
        }
        ER_REQUIRE(endp == num + len && errno == 0, "er_decode_csv_host: line %lld field %d: '%s' is not a number",
                   (long long)row, f, num);
      }
      b = e + 1;
    }
    ++row;
    pos = next;
  }
  *n_rows_out = row;
  *consumed_out = pos;
  return 0;
}

// The cells (begin, length) of a decoded text batch, in the order given, as one packed byte string + offsets[n + 1]:
// what er_hash_bucket_fast(_host) takes.  out_bytes must hold sum(length) bytes.
int er_pack_cells_host(const uint8_t* text, const int64_t* begin, const int32_t* length, int64_t n, uint8_t* out_bytes,
                       int64_t* out_offsets) {
  ER_REQUIRE(text && begin && length && out_offsets && n >= 0, "er_pack_cells_host: bad arguments");
  int64_t o = 0;
  for (int64_t i = 0; i < n; ++i) {  // (the copies on the host pool: measured no faster - 0.85 against 0.75 ms for 26 x 4096 cells)
    out_offsets[i] = o;
    if (length[i] > 0) {
      ER_REQUIRE(out_bytes, "er_pack_cells_host: null output");
      memcpy(out_bytes + o, text + begin[i], static_cast<size_t>(length[i]));
      o += length[i];
    }
  }
  out_offsets[n] = o;
  return 0;
}

// The cells (begin, length) of a text buffer split into tokens - what tf.string_split / tf.strings.split do to a TagFeature's
// or a SequenceFeature's column (reference input/input.py:488-530, 680-690) - as (begin, length) views of the same buffer
// plus row offsets [n + 1]: the ragged layout the lookup kernels take, with no per-row Python work.
//   keep_empty 0 (tf.string_split, tags): EVERY byte of `seps` is a delimiter, empty tokens are skipped, an empty cell has
//     no token;
//   keep_empty 1 (tf.strings.split with a one-byte separator, sequences): empty tokens stay, an empty cell is ONE empty
//     token; at most max_tokens per cell are kept (max_seq_len truncation; <= 0: all).
// tok_begin / tok_len hold `capacity` entries (sum(length) + n always suffices).  Returns 0 and the token count in
// row_offsets[n].
int er_split_cells_host(const uint8_t* text, const int64_t* begin, const int32_t* length, int64_t n, const uint8_t* seps,
                        int32_t n_seps, int32_t keep_empty, int32_t max_tokens, int64_t* tok_begin, int32_t* tok_len,
                        int64_t capacity, int64_t* row_offsets) {
  ER_REQUIRE(begin && length && seps && tok_begin && tok_len && row_offsets && n >= 0 && n_seps >= 1 && capacity >= 0 &&
                 (n == 0 || text) && (!keep_empty || n_seps == 1),
             "er_split_cells_host: bad arguments");
  bool is_sep[256];
  memset(is_sep, 0, sizeof(is_sep));
  for (int32_t k = 0; k < n_seps; ++k) is_sep[seps[k]] = true;
  int64_t t = 0;
  for (int64_t i = 0; i < n; ++i) {
    row_offsets[i] = t;
    const int64_t b = begin[i], e = b + length[i];
    int64_t kept = 0, tb = b;
    for (int64_t p = b; p <= e; ++p) {
      if (p < e && !is_sep[text[p]]) continue;
      const int64_t len = p - tb;  // a token ends at a delimiter or at the end of the cell
      if ((keep_empty || len > 0) && (max_tokens <= 0 || kept < max_tokens)) {
        ER_REQUIRE(t < capacity, "er_split_cells_host: more than %lld tokens", (long long)capacity);
        tok_begin[t] = tb;
        tok_len[t] = static_cast<int32_t>(len);
        ++t;
        ++kept;
      }
      tb = p + 1;
    }
  }
  row_offsets[n] = t;
  return 0;
}

// n int64 values as decimal strings ("-12", "0", "4294967295": Python's str(int), the reference's `_as_string` of an
// integer column, input/input.py:356-376) packed like er_pack_cells_host's output: what er_hash_bucket_fast(_host) takes
// for hashed IdFeatures fed from integer columns (the Criteo binary format's uint32 categories,
// input/criteo_input.py:75-85).  out_bytes must hold 20 bytes per value.  One pass, no per-value Python object.
int er_pack_int_decimal_host(const int64_t* values, int64_t n, uint8_t* out_bytes, int64_t* out_offsets) {
  ER_REQUIRE(out_offsets && n >= 0 && (n == 0 || (values && out_bytes)), "er_pack_int_decimal_host: bad arguments");
  static const char kPairs[201] =
      "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960"
      "616263646566676869707172737475767778798081828384858687888990919293949596979899";
  int64_t o = 0;
  char tmp[24];
  for (int64_t i = 0; i < n; ++i) {
    out_offsets[i] = o;
    const int64_t v = values[i];
    uint64_t u = v < 0 ? (~static_cast<uint64_t>(v) + 1u) : static_cast<uint64_t>(v);
    int k = 24;  // digits are written backwards, two at a time (a table of the 100 pairs), 32-bit arithmetic once they fit
    while (u > 0xFFFFFFFFull) {
      const uint64_t q = u / 100;
      const unsigned r = static_cast<unsigned>(u - q * 100);
      u = q;
      tmp[--k] = kPairs[2 * r + 1];
      tmp[--k] = kPairs[2 * r];
    }
    uint32_t w = static_cast<uint32_t>(u);
    while (w >= 100) {
      const uint32_t q = w / 100;
      const unsigned r = w - q * 100;
      w = q;
      tmp[--k] = kPairs[2 * r + 1];
      tmp[--k] = kPairs[2 * r];
    }
    if (w >= 10) {
      tmp[--k] = kPairs[2 * w + 1];
      tmp[--k] = kPairs[2 * w];
    } else {
      tmp[--k] = static_cast<char>('0' + w);
    }
    if (v < 0) out_bytes[o++] = '-';
    memcpy(out_bytes + o, tmp + k, static_cast<size_t>(24 - k));
    o += 24 - k;
  }
  out_offsets[n] = o;
  return 0;
}

}  // extern "C"
