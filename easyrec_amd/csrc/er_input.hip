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

#include "er_common.h"

extern "C" {

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
  for (int64_t i = 0; i < n; ++i) {
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

}  // extern "C"
