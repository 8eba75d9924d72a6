// Sanitizer driver for the host half of the input path (csrc/host.cpp: las_tfrecord_index, las_tfrecord_parse_batch -- a
// hand-written walk over TFRecord framing and SequenceExample protobuf bytes, preprocess_all.py:31-50 /
// utils/dataset_utils.py:141-153 of the reference).  Built by tests/test_host_sanitized.py with
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all  csrc/host.cpp tfrecord_fuzz.cpp
// (host code only: no GPU needed) and run on a corpus the test writes.  Every buffer the parser sees is an EXACT-SIZE heap
// block, so a read or write one byte past an end trips AddressSanitizer; UBSan watches the shifts and the pointer arithmetic.
//   tfrecord_fuzz <file.tfrecord> <num_channels> <mutations> <seed>
// prints: `full <records> <frames> <labels> <label bytes> <fnv of the parsed frames> <fnv of the label bytes>`, then one line
// `prefix <length> <records>` for every proper prefix of the file the CRC-checking index ACCEPTS (the test expects exactly the
// record boundaries), then `mutations <n> accepted_with_crc <k> parsed_without_crc <m>` (k must be 0).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/las_hip.h"

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

struct Parsed { int64_t records = -1, frames = 0, labels = 0, label_bytes = 0; uint64_t hf = 0, hl = 0; bool parsed = false; };

// index (+ parse of every record the index could size) of an exact-size copy of buf[0:n]
static Parsed run(const uint8_t* buf, size_t n, int verify, int F) {
  Parsed out;
  uint8_t* d = (uint8_t*)malloc(n ? n : 1);
  if (n) memcpy(d, buf, n);
  const int64_t cnt = las_tfrecord_index(d, n, verify, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
  out.records = cnt;
  if (cnt > 0) {
    int64_t* off = (int64_t*)malloc(sizeof(int64_t) * cnt);
    int64_t* len = (int64_t*)malloc(sizeof(int64_t) * cnt);
    int32_t* nf = (int32_t*)malloc(sizeof(int32_t) * cnt);
    int32_t* nl = (int32_t*)malloc(sizeof(int32_t) * cnt);
    int64_t* lb = (int64_t*)malloc(sizeof(int64_t) * cnt);
    const int64_t again = las_tfrecord_index(d, n, verify, cnt, off, len, nf, nl, lb);
    if (again != cnt) { fprintf(stderr, "index is not repeatable: %lld then %lld\n", (long long)cnt, (long long)again); exit(3); }
    std::vector<int64_t> o2, l2;
    int64_t rows = 0, toks = 0, bytes = 0;
    for (int64_t i = 0; i < cnt; ++i) {
      if (off[i] < 12 || off[i] + len[i] + 4 > (int64_t)n) { fprintf(stderr, "record %lld outside the buffer\n", (long long)i); exit(3); }
      if (nf[i] < 0 || nl[i] < 0) continue;          // not a SequenceExample with inputs / labels: the index says so
      o2.push_back(off[i]); l2.push_back(len[i]);
      rows += nf[i]; toks += nl[i]; bytes += lb[i];
    }
    const int m = (int)o2.size();
    if (m > 0) {
      float* frames = (float*)malloc(sizeof(float) * (size_t)(rows > 0 ? rows : 1) * F);
      int64_t* row_off = (int64_t*)malloc(sizeof(int64_t) * (m + 1));
      uint8_t* lbytes = (uint8_t*)malloc(bytes > 0 ? bytes : 1);
      int32_t* tok_off = (int32_t*)malloc(sizeof(int32_t) * (toks + 1));
      int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * m);
      const int rc = las_tfrecord_parse_batch(d, o2.data(), l2.data(), m, F, frames, rows, row_off, lbytes, bytes, tok_off, toks, counts);
      if (rc == 0) {
        out.parsed = true;
        out.frames = row_off[m]; out.labels = 0;
        for (int i = 0; i < m; ++i) out.labels += counts[i];
        out.label_bytes = tok_off[out.labels];
        out.hf = fnv(frames, sizeof(float) * (size_t)out.frames * F);
        out.hl = fnv(lbytes, (size_t)out.label_bytes);
        if (out.frames != rows || out.labels != toks || out.label_bytes != bytes) {
          fprintf(stderr, "parse and index disagree: %lld/%lld frames, %lld/%lld labels, %lld/%lld bytes\n", (long long)out.frames,
                  (long long)rows, (long long)out.labels, (long long)toks, (long long)out.label_bytes, (long long)bytes);
          exit(3);
        }
      }
      free(frames); free(row_off); free(lbytes); free(tok_off); free(counts);
    }
    free(off); free(len); free(nf); free(nl); free(lb);
  }
  free(d);
  return out;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: tfrecord_fuzz file num_channels mutations seed\n"); return 2; }
  const int F = atoi(argv[2]), n_mut = atoi(argv[3]);
  uint64_t rng = strtoull(argv[4], nullptr, 10) * 2654435761ull + 88172645463325252ull;
  auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  std::vector<uint8_t> base;
  uint8_t tmp[65536];
  size_t k;
  while ((k = fread(tmp, 1, sizeof tmp, f)) > 0) base.insert(base.end(), tmp, tmp + k);
  fclose(f);
  const size_t n = base.size();

  const Parsed full = run(base.data(), n, 1, F);
  if (full.records < 0 || !full.parsed) { fprintf(stderr, "the unmodified file does not parse: %s\n", las_last_error()); return 3; }
  printf("full %lld %lld %lld %lld %llu %llu\n", (long long)full.records, (long long)full.frames, (long long)full.labels,
         (long long)full.label_bytes, (unsigned long long)full.hf, (unsigned long long)full.hl);

  for (size_t len = 0; len < n; ++len) {                 // truncation at every byte offset
    const Parsed p = run(base.data(), len, 1, F);
    if (p.records >= 0) printf("prefix %zu %lld\n", len, (long long)p.records);
    (void)run(base.data(), len, 0, F);                   // without the CRC check: anything but a crash
  }

  int accepted = 0, parsed = 0;
  std::vector<uint8_t> m;
  for (int i = 0; i < n_mut && n > 0; ++i) {
    m = base;
    const int kind = (int)(next() % 4);
    const size_t at = (size_t)(next() % n);
    if (kind == 0) m[at] ^= (uint8_t)(1u << (next() % 8));                       // one bit
    else if (kind == 1) m[at] = (uint8_t)next();                                 // one byte (may repeat the old value: skipped below)
    else if (kind == 2) { for (int j = 0; j < 4 && at + j < n; ++j) m[at + j] = 0xff; }   // a run of 0xff: endless varints, huge lengths
    else { const uint64_t huge = next(); memcpy(&m[at], &huge, n - at < 8 ? n - at : 8); } // a garbage 64-bit word (length fields)
    if (m == base) continue;
    if (run(m.data(), n, 1, F).records >= 0) ++accepted;                         // a changed byte must fail one of the two CRCs
    if (run(m.data(), n, 0, F).parsed) ++parsed;
  }
  printf("mutations %d accepted_with_crc %d parsed_without_crc %d\n", n_mut, accepted, parsed);
  return 0;
}
