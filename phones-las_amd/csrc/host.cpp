// Host-only C++ of liblas_hip (no HIP include): the error state, CRC-32C and the input path's walk over TFRecord framing and
// SequenceExample protobuf bytes (utils/dataset_utils.py:138-283 of the reference; written by preprocess_all.py:31-50):
//   las_tfrecord_index        TFRecord framing of a file image (length, masked crc32c, payload, masked crc32c) + the frame /
//                             label counts of every SequenceExample, one pass, optional CRC check (hardware crc32c)
//   las_tfrecord_parse_batch  SequenceExample{feature_lists{'inputs': float_list per frame, 'labels': bytes_list per step}}
//                             -> packed [sum T, F] float frames + the label tokens as bytes with offsets
//   las_vocab_lookup          token bytes -> ids through an open-addressing table of FNV-1a hashes
// These functions read bytes that come from files: tests/test_host_sanitized.py builds this file alone with
// -fsanitize=address,undefined and runs it over every truncation and seeded mutations of a corpus (tests/native/tfrecord_fuzz.cpp).
#include "las_host.h"
#include <string.h>
#if defined(__x86_64__)
#include <nmmintrin.h>
#endif

static thread_local char g_err[512] = "";

void las_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* las_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// Host helper: CRC-32C (Castagnoli) as used by the TFRecord framing (masked crc of length and payload).
// ------------------------------------------------------------------------------------------------
extern "C" uint32_t las_crc32c(const void* data, size_t n) {
  struct Table {
    uint32_t t[8][256];
    Table() {
      for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
        t[0][i] = c;
      }
      for (uint32_t i = 0; i < 256; ++i)
        for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
    }
  };
  static const Table tab;                       // (a function-local static: initialised once, thread-safe -- the input path calls this from its prefetch thread)
  const uint32_t (*table)[256] = tab.t;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = 0xffffffffu;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = table[7][lo & 0xff] ^ table[6][(lo >> 8) & 0xff] ^ table[5][(lo >> 16) & 0xff] ^ table[4][lo >> 24] ^
        table[3][hi & 0xff] ^ table[2][(hi >> 8) & 0xff] ^ table[1][(hi >> 16) & 0xff] ^ table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

namespace {

// ---- crc32c ---------------------------------------------------------------------------------------
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) uint32_t crc32c_hw(const unsigned char* p, size_t n) {
  uint64_t c = 0xffffffffu;
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    c = _mm_crc32_u64(c, v);
    p += 8;
    n -= 8;
  }
  uint32_t c32 = (uint32_t)c;
  while (n--) c32 = _mm_crc32_u8(c32, *p++);
  return c32 ^ 0xffffffffu;
}
#endif

uint32_t crc32c(const unsigned char* p, size_t n) {
#if defined(__x86_64__)
  static int hw = -1;
  if (hw < 0) hw = __builtin_cpu_supports("sse4.2") ? 1 : 0;
  if (hw) return crc32c_hw(p, n);
#endif
  return las_crc32c(p, n);
}

inline uint32_t masked(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }

// ---- protobuf wire walking ---------------------------------------------------------------------------
struct Span {
  const unsigned char* p;
  const unsigned char* end;
};

inline bool varint(Span& s, uint64_t* out) {
  uint64_t v = 0;
  int shift = 0;
  while (s.p < s.end && shift < 64) {
    const unsigned char b = *s.p++;
    v |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) { *out = v; return true; }
    shift += 7;
  }
  return false;
}

// next field of a message: returns false at the end or on a malformed field (ok tells which); length-delimited payloads
// come back as a sub-span, the other wire types are skipped over with val holding varints
inline bool next_field(Span& s, int* field, int* wt, Span* sub, uint64_t* val, bool* ok) {
  *ok = true;
  if (s.p >= s.end) return false;
  uint64_t key;
  if (!varint(s, &key)) { *ok = false; return false; }
  *field = (int)(key >> 3);
  *wt = (int)(key & 7);
  switch (*wt) {
    case 0: if (!varint(s, val)) { *ok = false; return false; } return true;
    case 1: if (s.end - s.p < 8) { *ok = false; return false; } sub->p = s.p; sub->end = s.p + 8; s.p += 8; return true;
    case 5: if (s.end - s.p < 4) { *ok = false; return false; } sub->p = s.p; sub->end = s.p + 4; s.p += 4; return true;
    case 2: {
      uint64_t ln;
      if (!varint(s, &ln) || ln > (uint64_t)(s.end - s.p)) { *ok = false; return false; }
      sub->p = s.p;
      sub->end = s.p + ln;
      s.p += ln;
      return true;
    }
    default: *ok = false; return false;
  }
}

// the two feature lists of a SequenceExample (feature_lists = field 2; map entries = field 1 {key 1, value 2})
bool find_lists(Span ex, Span* inputs, Span* labels) {
  inputs->p = inputs->end = labels->p = labels->end = nullptr;
  int f, wt; Span sub; uint64_t v; bool ok;
  while (next_field(ex, &f, &wt, &sub, &v, &ok)) {
    if (f != 2 || wt != 2) continue;
    Span lists = sub;
    int f2, wt2; Span entry;
    while (next_field(lists, &f2, &wt2, &entry, &v, &ok)) {
      if (f2 != 1 || wt2 != 2) continue;
      Span key = {nullptr, nullptr}, value = {nullptr, nullptr};
      int f3, wt3; Span s3;
      Span e = entry;
      while (next_field(e, &f3, &wt3, &s3, &v, &ok)) {
        if (wt3 != 2) continue;
        if (f3 == 1) key = s3;
        else if (f3 == 2) value = s3;
      }
      if (!ok) return false;
      const size_t kl = (size_t)(key.end - key.p);
      if (kl == 6 && memcmp(key.p, "inputs", 6) == 0) *inputs = value;
      else if (kl == 6 && memcmp(key.p, "labels", 6) == 0) *labels = value;
    }
    if (!ok) return false;
  }
  return ok && inputs->p != nullptr && labels->p != nullptr;
}

// The token of a label step: Feature{bytes_list(1){value(1)}}, the FIRST value of the first bytes_list that holds one
// (FixedLenSequenceFeature([], string)).  tok->p stays null when there is none; false: malformed bytes.  The index (sizes) and
// the batch parser (copies) both come through here, so they cannot disagree about a record -- round 6: they did on records with
// a repeated bytes_list field (only reachable with the CRC check off; found by tests/native/tfrecord_fuzz.cpp).
bool first_token(Span feat, Span* tok) {
  tok->p = tok->end = nullptr;
  int f, wt; Span bl; uint64_t v; bool ok;
  while (next_field(feat, &f, &wt, &bl, &v, &ok)) {
    if (f != 1 || wt != 2) continue;               // Feature.bytes_list
    int f2, wt2; Span tv; bool ok2;
    Span b = bl;
    while (next_field(b, &f2, &wt2, &tv, &v, &ok2))
      if (f2 == 1 && wt2 == 2) { *tok = tv; return true; }
    if (!ok2) return false;
  }
  return ok;
}

// number of Feature messages of a FeatureList (field 1), and for bytes features the total token bytes
bool count_features(Span list, int* count, int64_t* token_bytes) {
  int f, wt; Span feat; uint64_t v; bool ok;
  *count = 0;
  if (token_bytes) *token_bytes = 0;
  while (next_field(list, &f, &wt, &feat, &v, &ok)) {
    if (f != 1 || wt != 2) continue;
    ++*count;
    if (!token_bytes) continue;
    Span tok;
    if (!first_token(feat, &tok)) return false;
    if (tok.p != nullptr) *token_bytes += tok.end - tok.p;
  }
  return ok;
}

// one frame: Feature{float_list(2){value(1): packed bytes or repeated fixed32}} -> F floats
bool read_frame(Span feat, int F, float* out) {
  int f, wt; Span fl; uint64_t v; bool ok;
  int n = 0;
  while (next_field(feat, &f, &wt, &fl, &v, &ok)) {
    if (f != 2 || wt != 2) continue;
    int f2, wt2; Span val; bool ok2;
    Span l = fl;
    while (next_field(l, &f2, &wt2, &val, &v, &ok2)) {
      if (f2 != 1) continue;
      if (wt2 == 2) {
        const int64_t k = (val.end - val.p) / 4;
        if (n + k > F) return false;
        memcpy(out + n, val.p, (size_t)k * 4);
        n += (int)k;
      } else if (wt2 == 5) {
        if (n + 1 > F) return false;
        memcpy(out + n, val.p, 4);
        ++n;
      }
    }
    if (!ok2) return false;
  }
  return ok && n == F;
}

}  // namespace

extern "C" int64_t las_tfrecord_index(const uint8_t* data, size_t nbytes, int verify_crc, int64_t max_records, int64_t* offsets,
                                      int64_t* lengths, int32_t* n_frames, int32_t* n_labels, int64_t* label_bytes) {
  if (!data && nbytes) { las_set_error("las_tfrecord_index: null buffer"); return LAS_ERR_ARG; }
  size_t pos = 0;
  int64_t n = 0;
  while (pos < nbytes) {
    if (nbytes - pos < 12) { las_set_error("las_tfrecord_index: truncated record header at byte %zu", pos); return LAS_ERR_ARG; }
    uint64_t ln;
    uint32_t crc_len;
    memcpy(&ln, data + pos, 8);
    memcpy(&crc_len, data + pos + 8, 4);
    if (verify_crc && masked(crc32c(data + pos, 8)) != crc_len) {
      las_set_error("las_tfrecord_index: corrupt length crc at byte %zu", pos);
      return LAS_ERR_ARG;
    }
    if (ln > nbytes - pos - 12 || nbytes - pos - 12 - ln < 4) {
      las_set_error("las_tfrecord_index: truncated record payload at byte %zu", pos);
      return LAS_ERR_ARG;
    }
    const uint8_t* payload = data + pos + 12;
    if (verify_crc) {
      uint32_t crc_data;
      memcpy(&crc_data, payload + ln, 4);
      if (masked(crc32c(payload, ln)) != crc_data) {
        las_set_error("las_tfrecord_index: corrupt payload crc at byte %zu", pos);
        return LAS_ERR_ARG;
      }
    }
    if (n < max_records) {
      if (offsets) offsets[n] = (int64_t)(pos + 12);
      if (lengths) lengths[n] = (int64_t)ln;
      if (n_frames || n_labels || label_bytes) {
        Span in, lab;
        int nf = -1, nl = -1;
        int64_t lb = 0;
        if (find_lists(Span{payload, payload + ln}, &in, &lab)) {
          if (!count_features(in, &nf, nullptr) || !count_features(lab, &nl, &lb)) nf = nl = -1;
        }
        if (n_frames) n_frames[n] = nf;
        if (n_labels) n_labels[n] = nl;
        if (label_bytes) label_bytes[n] = lb;
      }
    }
    ++n;
    pos += 12 + ln + 4;
  }
  return n;
}

extern "C" int las_tfrecord_parse_batch(const uint8_t* data, const int64_t* offsets, const int64_t* lengths, int n,
                                        int num_channels, float* frames, int64_t frame_rows_capacity, int64_t* frame_row_offsets,
                                        uint8_t* label_bytes, int64_t label_bytes_capacity, int32_t* token_offsets,
                                        int64_t token_capacity, int32_t* label_counts) {
  LAS_REQUIRE(offsets && lengths && n >= 0 && num_channels > 0 && frames && frame_row_offsets && token_offsets && label_counts,
              "las_tfrecord_parse_batch: null argument");
  int64_t row = 0, tok = 0, lb = 0;
  frame_row_offsets[0] = 0;
  token_offsets[0] = 0;
  for (int i = 0; i < n; ++i) {
    const uint8_t* p = data + offsets[i];
    Span in, lab;
    LAS_REQUIRE(find_lists(Span{p, p + lengths[i]}, &in, &lab), "las_tfrecord_parse_batch: record %d is not a SequenceExample with inputs / labels", i);
    int f, wt; Span feat; uint64_t v; bool ok;
    while (next_field(in, &f, &wt, &feat, &v, &ok)) {
      if (f != 1 || wt != 2) continue;
      LAS_REQUIRE(row < frame_rows_capacity, "las_tfrecord_parse_batch: frame buffer too small");
      LAS_REQUIRE(read_frame(feat, num_channels, frames + row * num_channels),
                  "las_tfrecord_parse_batch: record %d has a frame that does not hold num_channels=%d floats", i, num_channels);
      ++row;
    }
    LAS_REQUIRE(ok, "las_tfrecord_parse_batch: malformed inputs list in record %d", i);
    frame_row_offsets[i + 1] = row;
    int count = 0;
    while (next_field(lab, &f, &wt, &feat, &v, &ok)) {
      if (f != 1 || wt != 2) continue;
      Span tokspan;
      LAS_REQUIRE(first_token(feat, &tokspan) && tokspan.p != nullptr, "las_tfrecord_parse_batch: record %d has a label that is not a bytes token", i);
      const int64_t tl = tokspan.end - tokspan.p;
      LAS_REQUIRE(tok < token_capacity && lb + tl <= label_bytes_capacity, "las_tfrecord_parse_batch: label buffer too small");
      if (tl) memcpy(label_bytes + lb, tokspan.p, (size_t)tl);
      lb += tl;
      ++tok;
      token_offsets[tok] = (int32_t)lb;
      ++count;
    }
    LAS_REQUIRE(ok, "las_tfrecord_parse_batch: malformed labels list in record %d", i);
    label_counts[i] = count;
  }
  return LAS_OK;
}

extern "C" int las_vocab_lookup(const uint8_t* label_bytes, const int32_t* token_offsets, int64_t n_tokens, const uint64_t* keys,
                                const int32_t* vals, int64_t table_size, int32_t default_id, int32_t* ids) {
  LAS_REQUIRE(n_tokens >= 0 && table_size > 0 && (table_size & (table_size - 1)) == 0 && keys && vals && ids &&
                  (n_tokens == 0 || (label_bytes && token_offsets)),
              "las_vocab_lookup: bad arguments (table_size must be a power of two)");
  const uint64_t mask = (uint64_t)table_size - 1;
  for (int64_t k = 0; k < n_tokens; ++k) {
    uint64_t h = 1469598103934665603ull;                       // FNV-1a
    for (int32_t i = token_offsets[k]; i < token_offsets[k + 1]; ++i) h = (h ^ label_bytes[i]) * 1099511628211ull;
    if (h == 0) h = 1;
    int32_t id = default_id;
    for (uint64_t slot = h & mask, probes = 0; probes < (uint64_t)table_size; slot = (slot + 1) & mask, ++probes) {
      if (keys[slot] == h) { id = vals[slot]; break; }
      if (keys[slot] == 0) break;
    }
    ids[k] = id;
  }
  return LAS_OK;
}

