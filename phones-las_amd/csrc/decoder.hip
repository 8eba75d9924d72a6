// Speller step kernels: one AttentionWrapper(LSTMCell, Luong|Bahdanau) step of las/model.py:145-202 /
// SURVEY.md Appendix A.5-A.7, forward and backward, one 256-thread workgroup per utterance.
//
// The dense parts of a step ([attention_{t-1}, h_{t-1}] * K and its transpose) run in las_gemm_nt; these
// kernels fuse everything else of the step: token-row gather + bias + LSTM gate math, the score of h_t
// against the keys, the length-masked softmax (wave-shuffle + LDS reductions) and the context
// sum_t' align * values.  keys/values rows are read as 16-byte bf16x8 pieces (16 lanes per memory frame
// for scores, 4 columns per lane for the context).
//
// For the fused Speller (single cell, Hd 128/256, B <= 256) the whole decode loop runs in ONE launch per direction
// (dec_persist_fwd_kernel / dec_persist_bwd_kernel below): groups of 32 workgroups on one XCD own 8 utterances,
// keep their slice of the cell kernel in registers as MFMA fragments, and trade the per-step vectors through L2
// behind flag barriers.  The step kernels above remain the general path (stacked cells, wider shapes, greedy and
// beam decoding).  The file also holds the masked sequence loss, the log-probs loss of the binary-feature decoder
// and the beam-search step.
#include "las_common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  // 256 threads = 4 waves
  v = is_max ? las_wave_max(v) : las_wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  return r;
}

__device__ __forceinline__ bool att_additive(int a) { return a == LAS_ATT_BAHDANAU || a == LAS_ATT_BAHDANAU_MONOTONIC; }
__device__ __forceinline__ bool att_uses_wq(int a) { return att_additive(a) || a == LAS_ATT_CUSTOM; }

// In-place scan of arr[0..n) (LDS) by the 256 threads of the workgroup: sum or product, inclusive or exclusive,
// left-to-right or right-to-left.  Thread i owns a contiguous chunk; chunk totals are scanned through `tmp` (256 floats).
template <bool MUL>
__device__ void block_scan(float* arr, int n, float* tmp, bool exclusive, bool reverse) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ch = (n + 255) / 256;
  const int lo = tid * ch, hi = min(n, lo + ch);
  const float ident = MUL ? 1.f : 0.f;
  float tot = ident;
  // the callers fill arr[t] with thread t % 256; a chunk (or, right-to-left, any element) belongs to another thread as soon
  // as n exceeds one wave: without this barrier the reverse scans of the monotonic backward read what the previous phase had
  // left in the array (cfg5 at T' = 200: gradients of 1e36 after ten decoder steps, NaN after twenty)
  __syncthreads();
  for (int i = lo; i < hi; ++i) {
    const float v = arr[reverse ? n - 1 - i : i];
    tot = MUL ? tot * v : tot + v;
  }
  // exclusive scan of the 256 chunk totals: shuffles inside a wave, the four wave totals through `tmp` (a serial pass of
  // thread 0 over 256 LDS words was 6 us per scan: two of them per forward step of the monotonic normaliser, four per
  // backward step)
  float inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float nb = __shfl_up(inc, o, 64);
    if (lane >= o) inc = MUL ? inc * nb : inc + nb;
  }
  if (lane == 63) tmp[wave] = inc;
  float run = __shfl_up(inc, 1, 64);
  if (lane == 0) run = ident;
  __syncthreads();
  for (int w = 0; w < wave; ++w) run = MUL ? tmp[w] * run : tmp[w] + run;
  for (int i = lo; i < hi; ++i) {
    const int j = reverse ? n - 1 - i : i;
    const float v = arr[j];
    const float nxt = MUL ? run * v : run + v;
    arr[j] = exclusive ? run : nxt;
    run = nxt;
  }
  __syncthreads();
}

// The same scans for n <= 256 with element t in a REGISTER of thread t (v = the identity for t >= n): a wave-shuffle scan, the four
// wave totals through tmp4 (4 floats; the caller alternates between two such buffers from one scan to the next, so a buffer is
// rewritten only behind the barrier of the scan in between) and ONE barrier.  block_scan keeps its array in LDS and needs three;
// the monotonic normaliser runs two scans per forward and four per backward step.  Forward scans add in block_scan's order.
template <bool MUL, bool REVERSE, bool EXCLUSIVE>
__device__ __forceinline__ float scan256(const float v, float* tmp4, const int lane, const int wave) {
  const float ident = MUL ? 1.f : 0.f;
  float inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float nb = REVERSE ? __shfl_down(inc, o, 64) : __shfl_up(inc, o, 64);
    if (REVERSE ? (lane + o < 64) : (lane >= o)) inc = MUL ? inc * nb : inc + nb;
  }
  if (lane == (REVERSE ? 0 : 63)) tmp4[wave] = inc;
  float run = REVERSE ? __shfl_down(inc, 1, 64) : __shfl_up(inc, 1, 64);
  if (lane == (REVERSE ? 63 : 0)) run = ident;
  __syncthreads();
  if (REVERSE) { for (int w = 3; w > wave; --w) run = MUL ? tmp4[w] * run : tmp4[w] + run; }
  else { for (int w = 0; w < wave; ++w) run = MUL ? tmp4[w] * run : tmp4[w] + run; }
  return EXCLUSIVE ? run : (MUL ? run * v : run + v);
}

__device__ __forceinline__ float dot8(const uint4& k, const float* q) {
  const unsigned short* e = reinterpret_cast<const unsigned short*>(&k);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += las_bf2f(e[j]) * q[j];
  return s;
}

__device__ __forceinline__ uint4 ld16(const unsigned short* p);
// out[c] = sum_r W[r][c] * x[r] for a square bf16 matrix W [n, n] (row-major) by the 256 threads of the workgroup; x, out and
// scratch (2048 floats) in LDS.  The query layer of the Bahdanau / Custom attentions: forward pq = h Wq (W = Wq), backward
// dh = dpq Wq^T (W = the transposed copy).  Thread (column group of 8, row phase) walks its rows with 16-byte loads, eight
// in flight, and the row phases meet in scratch.  (A column at a time with 2-byte loads -- n dependent L2 round trips per
// thread -- made this product the longest phase of a decoder step at 512 units: 75 of 100 us.)
__device__ void square_matvec_bf16(const unsigned short* __restrict__ W, const float* x, float* out, float* scratch, int n) {
  const int tid = threadIdx.x;
  const int AG = n / 8;
  if (AG > 256 || (n & 7)) {            // wider than 2048: a column per thread
    for (int c = tid; c < n; c += 256) {
      float acc = 0.f;
      for (int r = 0; r < n; ++r) acc += las_bf2f(W[(int64_t)r * n + c]) * x[r];
      out[c] = acc;
    }
    __syncthreads();
    return;
  }
  const int UG = 256 / AG;
  const int ag = tid % AG, ug = tid / AG;
  if (ug < UG) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const unsigned short* wp = W + ag * 8;
    int r = ug;
    for (; r + 7 * UG < n; r += 8 * UG) {
      uint4 w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = ld16(wp + (int64_t)(r + i * UG) * n);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned short* e = reinterpret_cast<const unsigned short*>(&w[i]);
        const float xv = x[r + i * UG];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += xv * las_bf2f(e[j]);
      }
    }
    for (; r < n; r += UG) {
      const uint4 w = ld16(wp + (int64_t)r * n);
      const unsigned short* e = reinterpret_cast<const unsigned short*>(&w);
      const float xv = x[r];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += xv * las_bf2f(e[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) scratch[(ug * AG + ag) * 8 + j] = acc[j];
  }
  __syncthreads();
  for (int c = tid; c < n; c += 256) {
    float t = 0.f;
    for (int g = 0; g < UG; ++g) t += scratch[(g * AG + (c >> 3)) * 8 + (c & 7)];
    out[c] = t;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Set inside the persistent decoder only: the context-column parts of an utterance then SPLIT THE FRAMES of the score
// phase (each part reads a quarter of the keys instead of all of them), publish their raw scores in a per-step row of
// global scratch and meet at the group's flag barrier before the softmax.
typedef unsigned long long pu64;
// Workgroup barrier that orders LDS traffic only (what the workgroups exchange through global memory is ordered by the granule
// tags and by persist_barrier's own vmcnt(0); threads of one workgroup exchange through LDS only).  Used inside the one-launch
// decoder's step loops; measured: no difference to __syncthreads() there (hipcc's barrier does not drain vmcnt on gfx950
// either -- the vmcnt(0) waits of the phases come from the compiler's own tracking of loads carried across iterations).
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) only (vmcnt, expcnt untouched)
  __builtin_amdgcn_s_barrier();
}
// 8-byte {tag, fp32} granule of the persistent decoder's utterance-local exchanges (see persist_exchange_words)
__device__ __forceinline__ void pgranule_store(pu64* p, unsigned tag, float value, bool local) {
  const pu64 x = ((pu64)tag << 32) | __float_as_uint(value);
  if (local) *p = x;
  else __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ pu64 pgranule_load(const pu64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16-byte loads from global memory or, through an address-space-qualified pointer, from LDS (a pointer that travelled
// through a struct is otherwise a FLAT access: slower than either)
typedef const __attribute__((address_space(3))) unsigned short* lds_cu16;
__device__ __forceinline__ uint4 ld16(const unsigned short* p) { return *reinterpret_cast<const uint4*>(p); }
typedef __attribute__((ext_vector_type(4))) unsigned int las_u32x4;
__device__ __forceinline__ uint4 ld16(lds_cu16 p) {
  const las_u32x4 v = *(const __attribute__((address_space(3))) las_u32x4*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}

typedef __attribute__((ext_vector_type(8))) __bf16 sq_bf16x8;
typedef __attribute__((ext_vector_type(4))) float sq_f32x4;
// out[n] = sum_k a[k] W[n][k] for a square matrix of 32 KC rows handed in as its LAS_IMAGE_PACK_MFMA_B image (2 KC tiles of 16
// rows x KC chunks of 32), a (bf16) in LDS with 8 zeros at `zeros`: a is row 0 of the MFMA A tile, wave w takes the tiles w,
// w + 4, ..., TPB of them -- TPB KC fragments, one contiguous KB each -- in flight at a time.  The query layer of the Bahdanau /
// Custom attentions in the one-launch decoders (square_matvec_bf16 walks the row-major matrix with eight 16-byte loads in
// flight: 2.5 us per step at 256 units).
template <int KC, int TPB>
__device__ __forceinline__ void square_matvec_mfma(const unsigned short* packed, const unsigned short* a_lds, const unsigned short* zeros,
                                                   float* out, const int lane, const int wave) {
  constexpr int TPW = KC / 2;
  const int l15 = lane & 15, lq = lane >> 4;
  const unsigned short* azp = l15 == 0 ? a_lds + 8 * lq : zeros;
  const int azs = l15 == 0 ? 32 : 0;
  uint4 av[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) av[kc] = *reinterpret_cast<const uint4*>(azp + kc * azs);
#pragma unroll
  for (int i0 = 0; i0 < TPW; i0 += TPB) {
    uint4 bv[TPB][KC];
#pragma unroll
    for (int i = 0; i < TPB; ++i)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) bv[i][kc] = ld16(packed + (((int64_t)(wave + 4 * (i0 + i)) * KC + kc) * 64 + lane) * 8);
#pragma unroll
    for (int i = 0; i < TPB; ++i) {
      sq_f32x4 acc = sq_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sq_bf16x8, av[kc]), __builtin_bit_cast(sq_bf16x8, bv[i][kc]), acc, 0, 0, 0);
      if (lq == 0) out[(wave + 4 * (i0 + i)) * 16 + l15] = acc[0];
    }
  }
}

#ifdef LAS_STAMPS
// diagnostics build (LAS_CXXFLAGS=-DLAS_STAMPS): wall-clock (100 MHz) stamps of the phases of every step of workgroup 0
__device__ unsigned long long las_stamps[2 * 256 * 16];        // forward launch, then backward launch
#define LAS_STAMP(step, k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (step) < 256) las_stamps[(step) * 16 + (k)] = wall_clock64(); } while (0)
#define LAS_STAMPB(step, k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (step) < 256) las_stamps[4096 + (step) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define LAS_STAMP(step, k) do { } while (0)
#define LAS_STAMPB(step, k) do { } while (0)
#endif

// DropoutWrapper mask on 8 consecutive elements of a cell's input row (elements idx0 .. idx0 + 7 of generator stream `stream`):
// exactly what las_dropout_bf16 writes, applied where a product role reads the operand row (two-cell one-launch decoders)
__device__ __forceinline__ unsigned drop2(unsigned x, unsigned seed, unsigned stream, unsigned long long idx, float keep, float inv) {
  const float lo = __uint_as_float(x << 16) * (las_uniform(seed, stream, idx) < keep ? inv : 0.f);
  const float hi = __uint_as_float(x & 0xffff0000u) * (las_uniform(seed, stream, idx + 1) < keep ? inv : 0.f);
  return (unsigned)las_f2bf(lo) | ((unsigned)las_f2bf(hi) << 16);
}
__device__ __forceinline__ uint4 drop8(uint4 v, unsigned seed, unsigned stream, unsigned long long idx0, float keep, float inv) {
  // (register arithmetic only: through a pointer to its halves the vector -- and the array it came from -- went to scratch)
  return make_uint4(drop2(v.x, seed, stream, idx0, keep, inv), drop2(v.y, seed, stream, idx0 + 2, keep, inv),
                    drop2(v.z, seed, stream, idx0 + 4, keep, inv), drop2(v.w, seed, stream, idx0 + 6, keep, inv));
}

struct PersistHook {
  pu64* xsc;            // raw-score granules of this utterance and step parity: [ld] {tag, fp32}
  pu64* xz;             // z_t granules of this utterance and step parity: [4Hd] {tag, fp32}
  unsigned xtag;        // tag of this step (1, 2, ...)
  pu64* flags;
  int member;
  unsigned* epoch;
  bool local;
  int* fail;
  // LDS copies of what this workgroup reads from the memory at every step (nullptr: read from global memory):
  const unsigned short* wq_pk;   // B-fragment image of the query layer's kernel transposed (nullable): pq = h Wq on MFMA
  const unsigned short* lkeys;   // keys of its score frames [fq][Hd], row 0 = frame part*fq
  const unsigned short* lvals;   // its context columns of the frames [0, vres): [vres][cols_per]
  // operands of the cell that do not depend on this step's product, fetched while the product was running
  // (unit = threadIdx.x; Hd <= 256): token id, its row of the cell kernel, c_{t-1}; the bias once per launch
  bool pre;
  int tok;
  float tok4[4], bias4[4], cprev;
  int vres = 1 << 30;  // value frames [0, vres) are in LDS (lvals), the rest is read from memory (partial residency: long memories)
};
__device__ bool persist_barrier(pu64* flags, int member, unsigned epoch, bool local, int* lds_fail);
// Exchange discipline of the persistent decoder: every (utterance, step) row of an exchanged tensor occupies WHOLE
// 128-byte cache lines and is read only after it is complete, so a workgroup never holds a line in its L1 that somebody
// else is still going to write (a row sharing a line with the next step's row would be served stale from L1 later; the
// host pads the small rows - raw scores, partial dots - to 32 floats).

// Bahdanau scores of the persistent decoders' own frames at a compile-time width (NK = Hd / 32 sixteen-byte pieces per lane and
// frame; 4 lanes per frame, 64 frames per pass): score[t] = sum_a v[a] tanh(keys[t, a] + pq[a]).  pq and v come from LDS as
// 16-byte reads, every piece has its own partial sum, nothing in the pass is conditional but the final store.  (The general
// loop below read pq two floats and v four floats at a time, each read waited for before its two tanh, one running sum:
// 7.5 us per step at cfg5 for 64 tanh per lane.)
template <int NK, typename KR>
__device__ __forceinline__ void score_additive_rows(KR krows, const int row0, const int Hd, const int f0, const int f1, const int len,
                                                    const float* pql, const float* vl, const int lane, const int wave,
                                                    const PersistHook* ph) {
  const int sub = lane & 3, fr = lane >> 2;
  for (int t0 = f0; t0 < f1; t0 += 64) {
    const int t = t0 + wave * 16 + fr;
    const auto krow = krows + (int64_t)(min(t, f1 - 1) - row0) * Hd + sub * 8;
    uint4 kv[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) kv[j] = ld16(krow + j * 32);
    float pe = 0.f, po = 0.f;                       // even and odd pieces: two independent chains
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const int k = sub * 8 + j * 32;
      const float4 q0 = *reinterpret_cast<const float4*>(pql + k), q1 = *reinterpret_cast<const float4*>(pql + k + 4);
      const float4 v0 = *reinterpret_cast<const float4*>(vl + k), v1 = *reinterpret_cast<const float4*>(vl + k + 4);
      const uint4 kk = kv[j];
      const float a0 = v0.x * las_tanh(__uint_as_float(kk.x << 16) + q0.x) + v0.y * las_tanh(__uint_as_float(kk.x & 0xffff0000u) + q0.y);
      const float a1 = v0.z * las_tanh(__uint_as_float(kk.y << 16) + q0.z) + v0.w * las_tanh(__uint_as_float(kk.y & 0xffff0000u) + q0.w);
      const float a2 = v1.x * las_tanh(__uint_as_float(kk.z << 16) + q1.x) + v1.y * las_tanh(__uint_as_float(kk.z & 0xffff0000u) + q1.y);
      const float a3 = v1.z * las_tanh(__uint_as_float(kk.w << 16) + q1.z) + v1.w * las_tanh(__uint_as_float(kk.w & 0xffff0000u) + q1.w);
      if (j & 1) po += (a0 + a1) + (a2 + a3);
      else pe += (a0 + a1) + (a2 + a3);
    }
    float part_sum = pe + po;
    part_sum += __shfl_xor(part_sum, 1, 64);
    part_sum += __shfl_xor(part_sum, 2, 64);
    if (sub == 0 && t < f1) pgranule_store(ph->xsc + t, ph->xtag, (t < len) ? part_sum : -INFINITY, ph->local);
  }
}

// RES: the persistent kernel's LDS copies of keys / values are in use (compile-time: one load flavour per instantiation)
template <bool RES = false>
__device__ __forceinline__ void dec_step_fwd_body(const las_dec_step& s, const int b, const int part, const int nparts, float* sm,
                                  const PersistHook* ph = nullptr) {
  float* hq = sm;                 // [Hd] h_t (bf16-rounded) as float
  float* pq = hq + s.Hd;          // [Hd] processed query (Bahdanau)
  float* sc = pq + s.Hd;          // [Tm] scores -> probabilities
  float* red = sc + s.Tm;         // [8]

  float* cred = red + 8;          // [256][8] floats (context phases); monotonic work arrays follow it
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hd = s.Hd, M = s.M, Tm = s.Tm;
  const int len = (s.mode == LAS_DEC_CELL_ONLY) ? 0 : min(s.mem_len[b], Tm);
  const bool writer = (part == 0);

  if (s.mode == LAS_DEC_ATTENTION_ONLY) {
    // the query comes from another cell (top of a MultiRNNCell stack, las/model.py:194-200)
    for (int u = tid; u < Hd; u += 256) {
      const unsigned short qv = s.query[(int64_t)b * s.ldq + u];
      hq[u] = las_bf2f(qv);
      if (writer && s.h_out2) s.h_out2[(int64_t)b * s.ldh2 + u] = qv;      // the query next to the context: [query | context]
    }
  } else {
  // ---- LSTM cell (Appendix A.1) ----
  // (persistent decoder with scheduled sampling: the id was written by another workgroup during this launch and shares
  // a cache line with ids read earlier: bypass L1)
  const bool pre = ph && ph->pre;
  const int tok = !s.tok_rows ? 0
                  : (pre ? ph->tok
                         : (ph ? __hip_atomic_load(s.tok_ids + (int64_t)b * s.tok_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : s.tok_ids[(int64_t)b * s.tok_stride]));
  // DropoutWrapper on the cell input: the one-hot feed keeps/loses its single non-zero entry
  float tok_scale = 1.0f;
  if (s.drop_keep < 1.0f)
    tok_scale = las_uniform(s.drop_seed, s.drop_stream, ((unsigned long long)s.step * s.B + b) * s.feed_width + tok) < s.drop_keep
                    ? 1.0f / s.drop_keep : 0.f;
  for (int u = tid; u < Hd; u += 256) {
    float z[4];
    if (ph) {
      // persistent decoder: the product slices arrive as granules from the 32 workgroups of the group (whole waves
      // are in this loop: Hd is a multiple of 64); wave-uniform, bounded polling
      unsigned spins = 0;
      for (;;) {
        pu64 gq[4];
        bool got = true;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          gq[g] = pgranule_load(ph->xz + g * Hd + u);
          got = got && ((unsigned)(gq[g] >> 32) == ph->xtag);
        }
        if (__all(got)) {
#pragma unroll
          for (int g = 0; g < 4; ++g) z[g] = __uint_as_float((unsigned)gq[g]);
          break;
        }
        if (++spins > (1u << 22) || *ph->fail) { *ph->fail = 1; z[0] = z[1] = z[2] = z[3] = 0.f; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = s.z[(int64_t)b * 4 * Hd + g * Hd + u];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (pre) {
        z[g] += ph->bias4[g] + tok_scale * ph->tok4[g];
      } else {
        z[g] += s.bias[g * Hd + u];
        if (s.tok_rows) z[g] += tok_scale * las_bf2f(s.tok_rows[(int64_t)tok * 4 * Hd + g * Hd + u]);
      }
    }
    const float gi = las_sigmoid(z[0]), gj = las_tanh(z[1]), gf = las_sigmoid(z[2] + 1.0f), go = las_sigmoid(z[3]);
    const float cn = gf * (pre ? ph->cprev : s.c_prev[(int64_t)b * s.ldcp + u]) + gi * gj;
    const unsigned short hb = las_f2bf(go * las_tanh(cn));
    hq[u] = las_bf2f(hb);
    if (writer) {
      float* gp = s.gates_out + (int64_t)b * s.ldg + u;
      gp[0] = gi; gp[Hd] = gj; gp[2 * Hd] = gf; gp[3 * Hd] = go;
      s.c_out[(int64_t)b * s.ldco + u] = cn;
      s.h_out[(int64_t)b * s.ldh + u] = hb;
      if (s.h_out2) s.h_out2[(int64_t)b * s.ldh2 + u] = hb;
    }
  }
  }
  if (s.mode == LAS_DEC_CELL_ONLY) return;
  if (ph) LAS_STAMP(s.step, 3);
  __syncthreads();
  if (ph) LAS_STAMP(s.step, 4);

  // ---- processed query (Bahdanau, Custom): pq[a] = sum_u Wq[u][a] h[u]  (TF Dense kernel layout [in,out]);
  //      CustomAttention applies relu and, like every AttentionWrapper query, it is a GEMM operand: bf16-rounded ----
  if (att_uses_wq(s.attention)) {
    if (ph && ph->wq_pk && (Hd == 256 || Hd == 128)) {
      // on the matrix cores: h_t (bf16, exactly what hq holds) as row 0 of the A tile
      unsigned short* hb = reinterpret_cast<unsigned short*>(cred);
      for (int u = tid; u < Hd; u += 256) hb[u] = (unsigned short)(__float_as_uint(hq[u]) >> 16);
      if (tid < 4) reinterpret_cast<unsigned*>(hb + Hd)[tid] = 0u;
      __syncthreads();
      if (Hd == 256) square_matvec_mfma<8, 2>(ph->wq_pk, hb, hb + Hd, pq, lane, wave);
      else square_matvec_mfma<4, 2>(ph->wq_pk, hb, hb + Hd, pq, lane, wave);
      __syncthreads();
    } else
    square_matvec_bf16(s.wq, hq, pq, cred, Hd);            // (cred: free until the context phase)
    for (int a = tid; a < Hd; a += 256) {
      float acc = pq[a];
      if (s.attention == LAS_ATT_CUSTOM) acc = las_bf2f(las_f2bf(fmaxf(acc, 0.f)));
      pq[a] = acc;
      if (writer && s.pq_out) s.pq_out[(int64_t)b * s.ldpq + a] = acc;
    }
    // (persistent decoders, Bahdanau scores: attention_v next to it in LDS -- cred is free until the context phase)
    if (ph && att_additive(s.attention))
      for (int a = tid; a < Hd; a += 256) (sm + ((2 * Hd + Tm + 8 + 3) & ~3))[a] = s.att_v[a];
    __syncthreads();
  }

  // ---- scores: 4 lanes per memory frame, 64 frames per pass; the 16-byte key loads of SC_PASSES passes are all issued
  //      before the first is used (every launch starts with cold L2s: each dependent round trip goes to Infinity Cache) ----
  const unsigned short* keys = s.keys + (int64_t)b * Tm * Hd;
  {
    constexpr int SC_PASSES = 4, KMAX = 8;            // KMAX x 16 B per lane and frame: Hd <= 256 in one go
    const int sub = lane & 3, fr = lane >> 2;
    const bool additive = att_additive(s.attention);
    const float* qv = s.attention == LAS_ATT_CUSTOM ? pq : hq;
    // frames this workgroup scores: all of them, or its share when the parts exchange scores (persistent decoder)
    const int fq = (Tm + nparts - 1) / nparts;
    const int f0 = ph ? part * fq : 0, f1 = ph ? min(Tm, f0 + fq) : Tm;
    const int flen = min(len, f1);
    if (ph && additive && (Hd == 256 || Hd == 128)) {
      const float* vl = sm + ((2 * Hd + Tm + 8 + 3) & ~3);
      auto run = [&](auto krows, int row0) {
        if (Hd == 256) score_additive_rows<8>(krows, row0, Hd, f0, f1, len, pq, vl, lane, wave, ph);
        else score_additive_rows<4>(krows, row0, Hd, f0, f1, len, pq, vl, lane, wave, ph);
      };
      if constexpr (RES) run((lds_cu16)ph->lkeys, f0);
      else run(keys, 0);
    } else if (Hd <= 32 * KMAX) {
      const int nk = Hd / 32;                          // loads per lane and frame
      // krows: key rows from memory (row 0 = frame 0) or from the workgroup's LDS copy (row 0 = frame f0)
      auto score_pass = [&](auto krows, int row0) {
      for (int t0 = f0; t0 < f1; t0 += 64 * SC_PASSES) {
        uint4 kv[SC_PASSES][KMAX];
#pragma unroll
        for (int p = 0; p < SC_PASSES; ++p) {
          const int t = t0 + p * 64 + wave * 16 + fr;
          const auto krow = krows + (int64_t)(min(t, f1 - 1) - row0) * Hd;
#pragma unroll
          for (int j = 0; j < KMAX; ++j)
            if (j < nk && t < flen) kv[p][j] = ld16(krow + sub * 8 + j * 32);
        }
#pragma unroll
        for (int p = 0; p < SC_PASSES; ++p) {
          const int t = t0 + p * 64 + wave * 16 + fr;
          float part_sum = 0.f;
          if (t < flen) {
#pragma unroll
            for (int j = 0; j < KMAX; ++j)
              if (j < nk) {
                const int k = sub * 8 + j * 32;
                if (!additive) part_sum += dot8(kv[p][j], qv + k);
                else {
                  const unsigned short* e = reinterpret_cast<const unsigned short*>(&kv[p][j]);
#pragma unroll
                  for (int i = 0; i < 8; ++i) part_sum += s.att_v[k + i] * las_tanh(las_bf2f(e[i]) + pq[k + i]);
                }
              }
          }
          part_sum += __shfl_xor(part_sum, 1, 64);
          part_sum += __shfl_xor(part_sum, 2, 64);
          if (sub == 0 && t < f1) {
            const float sv = (t < len) ? part_sum : -INFINITY;
            if (ph) pgranule_store(ph->xsc + t, ph->xtag, sv, ph->local);      // the other three parts are waiting for it
            else sc[t] = sv;
          }
        }
      }
      };
      if constexpr (RES) score_pass((lds_cu16)ph->lkeys, f0);
      else score_pass(keys, 0);
    } else {
      for (int t0 = f0; t0 < f1; t0 += 64) {
        const int t = t0 + wave * 16 + fr;
        float part_sum = 0.f;
        if (t < flen) {
          const unsigned short* krow = keys + (int64_t)t * Hd;
          for (int k = sub * 8; k < Hd; k += 32) {
            const uint4 kvv = *reinterpret_cast<const uint4*>(krow + k);
            if (!additive) part_sum += dot8(kvv, qv + k);
            else {
              const unsigned short* e = reinterpret_cast<const unsigned short*>(&kvv);
#pragma unroll
              for (int i = 0; i < 8; ++i) part_sum += s.att_v[k + i] * las_tanh(las_bf2f(e[i]) + pq[k + i]);
            }
          }
        }
        part_sum += __shfl_xor(part_sum, 1, 64);
        part_sum += __shfl_xor(part_sum, 2, 64);
        if (sub == 0 && t < f1) {
            const float sv = (t < len) ? part_sum : -INFINITY;
            if (ph) pgranule_store(ph->xsc + t, ph->xtag, sv, ph->local);      // the other three parts are waiting for it
            else sc[t] = sv;
          }
      }
    }
    if (ph) LAS_STAMP(s.step, 5);
    if (ph) {            // gather the whole score row: the data is its own flag (no group barrier)
      for (int t0 = 0; t0 < Tm; t0 += 256) {
        const int t = t0 + tid;
        unsigned spins = 0;
        for (;;) {                                      // wave-uniform, bounded
          const pu64 gq = t < Tm ? pgranule_load(ph->xsc + t) : ((pu64)ph->xtag << 32);
          if (__all((unsigned)(gq >> 32) == ph->xtag)) {
            if (t < Tm) sc[t] = __uint_as_float((unsigned)gq);
            break;
          }
          if (++spins > (1u << 22) || *ph->fail) { *ph->fail = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
  }
  __syncthreads();
  if (ph) LAS_STAMP(s.step, 6);

  if (s.norm != LAS_NORM_SOFTMAX) {
    // ---- monotonic attention (tf.contrib.seq2seq.monotonic_attention) ----
    float* wa = cred + 2048;          // [Tm]
    float* wb = wa + Tm;              // [Tm]
    float* tmp = cred;                // 256 floats of scan scratch (cred is free until the context phase)
    const float bias = s.score_bias ? *s.score_bias : 0.f;
    const float* prev = s.prev_align ? s.prev_align + (int64_t)b * s.ldpa : nullptr;
    if (Tm <= 256 && s.norm == LAS_NORM_MONOTONIC_PARALLEL) {
      // frame t in the registers of thread t from the score to the alignment: two scans, two barriers (see scan256)
      const int t = tid;
      float p = 0.f, pr = 0.f;
      if (t < Tm) {
        if (t < len) {
          float sv = sc[t] + bias;
          if (s.noise_scale > 0.f)
            sv += s.noise_scale * las_normal(s.noise_seed, s.noise_stream, ((unsigned long long)s.step * s.B + b) * Tm + t);
          p = las_sigmoid(sv);
        }
        pr = prev ? (ph ? __hip_atomic_load(prev + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : prev[t]) : (t == 0 ? 1.f : 0.f);
        if (writer && s.p_out) s.p_out[(int64_t)b * s.ldp + t] = p;
      }
      const float lc = t < Tm ? __logf(fminf(fmaxf(1.f - p, 1.17549435e-38f), 1.f)) : 0.f;
      const float c = __expf(scan256<false, false, true>(lc, tmp, lane, wave));
      const float S = scan256<false, false, false>(t < Tm ? pr / fminf(fmaxf(c, 1e-10f), 1.f) : 0.f, tmp + 4, lane, wave);
      if (t < Tm) {
        const float a = p * c * S;
        sc[t] = a;
        if (writer) {
          s.align_out[(int64_t)b * s.lda + t] = a;
          if (s.align_bf16) s.align_bf16[(int64_t)b * s.lda + t] = las_f2bf(a);
        }
      }
      __syncthreads();
    } else {
    for (int t = tid; t < Tm; t += 256) {
      float p = 0.f;
      if (t < len) {
        float sv = sc[t] + bias;
        if (s.noise_scale > 0.f)
          sv += s.noise_scale * las_normal(s.noise_seed, s.noise_stream, ((unsigned long long)s.step * s.B + b) * Tm + t);
        p = s.norm == LAS_NORM_MONOTONIC_HARD ? (sv > 0.f ? 1.f : 0.f) : las_sigmoid(sv);
      }
      sc[t] = p;
      // (one-launch decoder: the previous step's alignments were written by ANOTHER workgroup of this launch and their rows
      //  share cache lines with rows read earlier: bypass L1)
      wb[t] = prev ? (ph ? __hip_atomic_load(prev + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : prev[t]) : (t == 0 ? 1.f : 0.f);
      if (writer && s.p_out) s.p_out[(int64_t)b * s.ldp + t] = p;
    }
    __syncthreads();
    if (s.norm == LAS_NORM_MONOTONIC_PARALLEL) {
      // a = p * c * cumsum(prev / clip(c, 1e-10, 1)),  c = exclusive cumprod(1 - p) in log space (safe_cumprod)
      for (int t = tid; t < Tm; t += 256) wa[t] = __logf(fminf(fmaxf(1.f - sc[t], 1.17549435e-38f), 1.f));
      block_scan<false>(wa, Tm, tmp, true, false);
      for (int t = tid; t < Tm; t += 256) {
        const float c = __expf(wa[t]);
        wa[t] = c;
        wb[t] = wb[t] / fminf(fmaxf(c, 1e-10f), 1.f);
      }
      block_scan<false>(wb, Tm, tmp, false, false);
      for (int t = tid; t < Tm; t += 256) sc[t] = sc[t] * wa[t] * wb[t];
    } else {
      // hard: p *= cumsum(prev); a = p * exclusive cumprod(1 - p)
      block_scan<false>(wb, Tm, tmp, false, false);
      for (int t = tid; t < Tm; t += 256) { sc[t] *= wb[t]; wa[t] = 1.f - sc[t]; }
      block_scan<true>(wa, Tm, tmp, true, false);
      for (int t = tid; t < Tm; t += 256) sc[t] *= wa[t];
    }
    __syncthreads();
    for (int t = tid; t < Tm; t += 256) {
      if (writer) {
        s.align_out[(int64_t)b * s.lda + t] = sc[t];
        if (s.align_bf16) s.align_bf16[(int64_t)b * s.lda + t] = las_f2bf(sc[t]);
      }
    }
    __syncthreads();
    }
  } else {
  // ---- masked softmax over t' ----
  float mx = -INFINITY;
  for (int t = tid; t < Tm; t += 256) mx = fmaxf(mx, sc[t]);
  mx = block_reduce(mx, red, true);
  float sum = 0.f;
  for (int t = tid; t < Tm; t += 256) {
    const float e = (t < len) ? __expf(sc[t] - mx) : 0.f;
    sc[t] = e;
    sum += e;
  }
  sum = block_reduce(sum, red, false);
  const float inv = len > 0 ? 1.0f / sum : 0.f;
  for (int t = tid; t < Tm; t += 256) {
    const float p = sc[t] * inv;
    sc[t] = p;
    if (writer) {
      s.align_out[(int64_t)b * s.lda + t] = p;
      if (s.align_bf16) s.align_bf16[(int64_t)b * s.lda + t] = las_f2bf(p);
    }
  }
  __syncthreads();
  }

  if (ph) LAS_STAMP(s.step, 7);
  // ---- context = sum_t' p[t'] * values[b,t',:] for this workgroup's column range.  L = cols/8 lanes cover one
  // frame with 16-byte loads; the 256/L frame phases are reduced through LDS. ----
  const unsigned short* vals = s.values + (int64_t)b * Tm * M;
  const int cols_per = ((M / 8 + nparts - 1) / nparts) * 8;
  const int c_begin = part * cols_per, c_end = min(M, c_begin + cols_per);
  for (int cb = c_begin; cb < c_end; cb += 2048) {
    const int ncols = min(2048, c_end - cb);
    int L = 1;
    while (L * 8 < ncols) L <<= 1;           // lanes per frame (power of two <= 256)
    const int P = 256 / L;                    // frame phases
    const int phase = tid / L, cl = tid % L;
    const int col = cb + cl * 8;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    if (col < c_end) {
      constexpr int VB = 16;                 // value loads in flight per thread (one round trip to Infinity Cache)
      // vrows + t * vstride: this thread's 8 columns of frame t (memory, or the workgroup's LDS copy of its columns)
      // frames [ta, te) of the pass; vrows + t * vstride addresses frame t
      auto context_pass = [&](auto vrows, int64_t vstride, int ta, int te) {
        // (no branches in the pass: frames past the utterance re-read its last frame with weight 0 -- guarded loads were 16
        //  branches per pass, each on a per-lane condition)
        const int last = max(te - 1, ta);
        for (int tb = ta + phase; tb < te; tb += P * VB) {
          uint4 vv[VB];
          float pw[VB];
#pragma unroll
          for (int i = 0; i < VB; ++i) {
            const int t = tb + i * P;
            vv[i] = ld16(vrows + (int64_t)min(t, last) * vstride);
            pw[i] = t < te ? sc[min(t, last)] : 0.f;
          }
#pragma unroll
          for (int i = 0; i < VB; ++i) {
            const uint4 q = vv[i];
            const float p = pw[i];
            a[0] += p * __uint_as_float(q.x << 16); a[1] += p * __uint_as_float(q.x & 0xffff0000u);
            a[2] += p * __uint_as_float(q.y << 16); a[3] += p * __uint_as_float(q.y & 0xffff0000u);
            a[4] += p * __uint_as_float(q.z << 16); a[5] += p * __uint_as_float(q.z & 0xffff0000u);
            a[6] += p * __uint_as_float(q.w << 16); a[7] += p * __uint_as_float(q.w & 0xffff0000u);
          }
        }
      };
      if constexpr (RES) {
        context_pass((lds_cu16)ph->lvals + (col - c_begin), cols_per, 0, min(len, ph->vres));
        if (ph->vres < len) context_pass(vals + col, M, ph->vres, len);      // long memories: the frames that did not fit the LDS
      } else context_pass(vals + col, M, 0, len);
    }
    if (ph) LAS_STAMP(s.step, 8);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) cred[tid * 8 + j] = a[j];
    __syncthreads();
    // thread j < ncols sums the P phases of column j
    for (int j = tid; j < ncols; j += 256) {
      const int cl2 = j >> 3, e = j & 7;
      float acc = 0.f;
      for (int ph = 0; ph < P; ++ph) acc += cred[(ph * L + cl2) * 8 + e];
      const unsigned short o = las_f2bf(acc);
      s.ctx_out[(int64_t)b * s.ldc + cb + j] = o;
      if (s.ctx_out2) {
        unsigned short o2 = o;
        if (s.drop_keep < 1.0f && s.feed_plain != 1) {   // the copy that feeds step t+1's cell goes through that step's input dropout
          // (feed_plain 2: the step-by-step path's draws -- one generator stream per step, element index b * M + column)
          // (feed_plain 2 without token rows -- a dense token vector in front of the feed --: the masked window is feed_width wide)
          const int W2 = s.tok_rows ? M : max(s.feed_width, M);
          const unsigned long long idx = s.feed_plain == 2 ? (unsigned long long)b * W2 + (W2 - M) + cb + j
                                         : ((unsigned long long)(s.step + 1) * s.B + b) * s.feed_width + (s.feed_width - M) + cb + j;
          const unsigned stream = s.feed_plain == 2 ? s.feed_stream0 + (unsigned)(s.step + 1) : s.drop_stream;
          o2 = s.feed_plain == 2 ? (las_uniform(s.drop_seed, stream, idx) < s.drop_keep ? las_f2bf(las_bf2f(o) * (1.0f / s.drop_keep)) : (unsigned short)0)
                                 : (las_uniform(s.drop_seed, stream, idx) < s.drop_keep ? las_f2bf(las_bf2f(o) / s.drop_keep) : (unsigned short)0);
        }
        s.ctx_out2[(int64_t)b * s.ldc2 + cb + j] = o2;
      }
    }
  }
  if (ph) LAS_STAMP(s.step, 9);
}


__global__ __launch_bounds__(256) void dec_step_fwd_kernel(las_dec_step s) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  dec_step_fwd_body(s, blockIdx.x, blockIdx.y, gridDim.y, sm);
}

// ------------------------------------------------------------------------------------------------
// PERSISTENT forward decoder: all U steps of the fused AttentionWrapper(LSTMCell) decoder in ONE launch.
// Why: every kernel launch starts with cold per-XCD L2s, so a per-step kernel pays an Infinity-Cache round trip for
// each dependent phase and re-fetches all keys/values (~20 MB) every step; here they stay in the XCD's L2.
// Decomposition: utterances are cut into groups of 8; a group is served by 32 co-resident workgroups that the
// block index places on ONE XCD (block = group + 8k*member).  Per step each member plays two roles:
//   G: columns [member*4Hd/32, ...) of z_t = [attention_{t-1}, h_{t-1}] K for the group's 8 utterances (its slice of
//      K stays in registers as MFMA B fragments for all steps; M = 16 rows, 8 used);
//   S: the step kernel's body for utterance member/4, context-column part member%4.
// Once per step (after S) the 32 members meet at a flag barrier in global memory (each writes its own epoch word,
// wave 0 polls all 32): what crosses it (c_t, the next operand row) is written once and read later at distinct
// addresses, so no stale L1 line can exist, and with the group on one XCD its L2 is the coherence point: plain
// stores, no fences.  If the members find themselves on different XCDs (XCC_ID handshake) they add agent-scope
// release/acquire fences around the barrier instead: slower, same results.  Everything else that travels inside a
// step -- z_t from the 32 product slices to the utterance's workgroups, the raw scores among the four workgroups of an
// utterance -- goes as 8-byte {step tag, fp32} granules that the consumer polls (two parity slots in the workspace):
// the data is its own flag, one L2 round trip instead of store + barrier + load.  Every spin is bounded (status word).
// ------------------------------------------------------------------------------------------------
constexpr unsigned P_SPIN_LIMIT = 1u << 22;
constexpr int P_MEMBERS = 32;
// LDS of the persistent backward kernel: its scratch, and -- when it fits the CU's 160 KiB -- this workgroup's quarter
// of the utterance's memory frames (values [fq, M] and keys [fq, Hd], bf16), which every step reads again.
// Workspace of a persistent launch: [64 B status][group flags and XCC table: groups * 2 * 32 words][exchange granules].
// The four workgroups of an UTTERANCE hand small results to each other as 8-byte {tag, fp32} granules (the data is its
// own flag, as in the recurrent kernels; two parity slots, tag = step count) instead of meeting all 32 workgroups of
// the group at a flag barrier: forward the raw scores [2][B][ld] (ld = frames rounded up to 32) and the gate
// pre-activations z_t [2][B][4Hd] (from the 32 product slices to the utterance's workgroups), backward the partial
// dots [2][B][4], the partial dh [2][B][3][Hd] and the feed gradients d[attention, h]_t [2][B][M+Hd] (from the 32 product
// slices to the utterance's workgroups of the step before).
__host__ __device__ inline size_t persist_flag_words(int B) { return (size_t)((B + 7) / 8) * 2 * P_MEMBERS; }
__host__ __device__ inline size_t persist_exchange_words(int B, int Tm, int Hd, int M) {
  const size_t ld = (size_t)((Tm + 31) / 32) * 32;
  // (backward with a second cell: + [2][B][W1] granules of its product, W1 <= 2 M + Hd)
  const size_t fwd = 2 * (size_t)B * (ld + 9 * (size_t)Hd), bwd = 2 * (size_t)B * (4 + 5 * (size_t)Hd + (size_t)(M + Hd) + (size_t)(2 * M + Hd));
  return fwd > bwd ? fwd : bwd;
}
// forward: scratch, and this workgroup's score frames of the keys [fq, Hd] + its context columns of the values [Tm, M/4]
// (red: [4 waves][16 rows][columns per member + 1] partial z tiles; 4 Hd / 32 columns per member, at least 32)
__host__ __device__ inline int persist_red_stride(int Hd) { return (Hd / 8 > 32 ? Hd / 8 : 32) + 1; }
__host__ __device__ inline size_t persist_fwd_scratch_floats(int Hd, int Tm) {
  return ((size_t)2 * Hd + Tm + 16 + 2048 + 4 * 16 * persist_red_stride(Hd) + 8 + 3) & ~(size_t)3;
}
__host__ __device__ inline int persist_cols_per(int M) { return ((M / 8 + 3) / 4) * 8; }
__host__ __device__ inline size_t persist_fwd_resident_bytes(int M, int Hd, int Tm) {
  return ((size_t)((Tm + 3) / 4) * Hd + (size_t)Tm * persist_cols_per(M)) * 2;
}
__host__ __device__ inline bool persist_fwd_resident(int M, int Hd, int Tm) {
  return persist_fwd_scratch_floats(Hd, Tm) * 4 + persist_fwd_resident_bytes(M, Hd, Tm) <= 158 * 1024;   // (2 KiB of margin below the CU's 160 KiB)
}
// Long memories (general kernel): the workgroup's key frames and the value frames [0, vres) of its columns in LDS, the rest of
// the values streamed at every step.  Returns vres: Tm (everything fits), a multiple of 16 below it, or 0 (nothing resident).
__host__ __device__ inline int persist_fwd_resident_frames(int M, int Hd, int Tm) {
  if (persist_fwd_resident(M, Hd, Tm)) return Tm;
  const long long room = 158 * 1024 - (long long)persist_fwd_scratch_floats(Hd, Tm) * 4 - (long long)((Tm + 3) / 4) * Hd * 2;
  const long long fr = room / ((long long)persist_cols_per(M) * 2);
  return fr >= 64 ? (int)(fr & ~15LL) : 0;
}
__host__ __device__ inline size_t persist_fwd_resident_bytes_partial(int M, int Hd, int Tm, int vres) {
  return ((size_t)((Tm + 3) / 4) * Hd + (size_t)vres * persist_cols_per(M)) * 2;
}
// G role's partial tiles: [4 waves][8 utterances][16-column tiles per member * 16 + 1] (W / 16 tiles over 32 members: <= 3, or 5)
__host__ __device__ inline size_t persist_bwd_red2_floats(int W) {
  const int nt = (W / 16 + 31) / 32;
  return (size_t)4 * 8 * ((nt <= 3 ? 3 : 5) * 16 + 1);
}
__host__ __device__ inline int persist_bwd_ds_pad(int Tm) { return (((Tm + 3) / 4 + 31) / 32) * 32; }
// Luong scores with the matrix-core passes: the keys lie TRANSPOSED in LDS, [Hd][FS] (a unit's frames are contiguous,
// as a B fragment reads them); FS = the frame share rounded up to an odd number of 16-byte pieces, so that the 16 units
// of a fragment fall on 16 different bank groups
__host__ __device__ inline int persist_bwd_kt_stride(int Tm) {
  int r = ((Tm + 3) / 4 + 7) & ~7;
  if (((r / 8) & 1) == 0) r += 8;
  return r;
}
// partial-tile area of the product roles: the cell product's tiles, at least 2048 floats; with a second cell its product's
// [4 waves][8][5 * 16 + 1] (up to 5 column tiles of 2 M + Hd columns per member)
__host__ __device__ inline size_t persist_bwd_red_area(int M, int Hd, bool two) {
  const size_t r2 = persist_bwd_red2_floats(M + Hd), lo = two ? (size_t)4 * 8 * (5 * 16 + 1) : 2048;
  return r2 > lo ? r2 : lo;
}
__host__ __device__ inline size_t persist_bwd_scratch_floats(int M, int Hd, int Tm, bool two = false) {
  // (the Bahdanau query-layer product borrows the partial-tile area as 2048 floats of scratch)
  const size_t r2 = persist_bwd_red_area(M, Hd, two);
  // (d(context): M floats, or two bf16 rows of M -- high and low halves -- for the matrix-core d(alignments) pass; + 16 B of zeros)
  // (+ ds of the own frames as two bf16 rows, high and low halves, padded to whole 32-frame chunks with zeros)
  return ((size_t)M + 4 + 2 * (size_t)Tm + 2048 + 16 + Hd + (r2 > 2048 ? r2 : 2048) + 8 + persist_bwd_ds_pad(Tm) + 3) & ~(size_t)3;
}
// rows of the LDS-resident values are P_VPAD elements apart from a multiple of 64 banks: the 16 frames of a matrix-core
// fragment read 16 different rows at the same column
#define P_VPAD 8
__host__ __device__ inline size_t persist_bwd_resident_bytes(int M, int Hd, int Tm, bool keys_t) {
  const size_t fq = (Tm + 3) / 4;
  return (fq * (size_t)(M + P_VPAD) + (keys_t ? (size_t)Hd * persist_bwd_kt_stride(Tm) : fq * (size_t)Hd)) * 2;
}
__host__ __device__ inline bool persist_bwd_resident(int M, int Hd, int Tm, bool keys_t, bool two = false) {
  return persist_bwd_scratch_floats(M, Hd, Tm, two) * 4 + persist_bwd_resident_bytes(M, Hd, Tm, keys_t) <= 158 * 1024;
}
// Long memories: the workgroup's key frames [fq, Hd] (row-major) and the first `rows` of its value frames in LDS, the rest of the
// values streamed at every step.  Returns fq (everything fits), a multiple of 8 below it, or 0 (nothing resident).
__host__ __device__ inline int persist_bwd_resident_rows(int M, int Hd, int Tm, bool two) {
  const int fq = (Tm + 3) / 4;
  if (persist_bwd_resident(M, Hd, Tm, false, two)) return fq;
  const long long room = 158 * 1024 - (long long)persist_bwd_scratch_floats(M, Hd, Tm, two) * 4 - (long long)fq * Hd * 2;
  const long long rows = room / ((long long)(M + P_VPAD) * 2);
  return rows >= 32 ? (int)(rows & ~7LL) : 0;
}

__device__ bool persist_barrier(pu64* flags, int member, unsigned epoch, bool local, int* lds_fail) {
  __builtin_amdgcn_s_waitcnt(0x0070);                      // vmcnt(0) lgkmcnt(0): this wave's stores are acknowledged
  if (!local) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (local) flags[member] = epoch;
    else __hip_atomic_store(flags + member, (pu64)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    unsigned spins = 0;
    for (;;) {
      const pu64 v = lane < P_MEMBERS ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (pu64)epoch;
      if (__all(v >= epoch)) break;
      if (++spins > P_SPIN_LIMIT) { if (lane == 0) *lds_fail = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  if (!local) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return *lds_fail == 0;
}

// NTL_MAX 16-column tiles of z per member (4 Hd / 32 / 16) and KCW_MAX 32-deep K chunks per wave (K_in / 32 / 4) stay in
// registers as MFMA B fragments: <2, 12> for decoder_units <= 256 (96 VGPRs), <4, 20, 8> for 512 units with a 2048-deep
// memory
// KRES: K chunks per wave that stay in registers; the chunks beyond are re-read from L2 / Infinity Cache at every step
// (512 units: 8 of 20 resident = 128 VGPRs; all 20 would need 320 of the 512 next to the step body's ~270).
// AL: the decoder has an attention layer (attention_layer_size / --binf_projection, las/model.py:179-200): attention_t =
// [h_t | context_t] W_al is its output and the next step's feed.  After the S role a group barrier, then members 0 .. A/16-1
// form one 16-column tile of attention_t each for the group's 8 utterances (K = Hd + M over the four waves, W_al slice in
// registers: KAL_MAX chunks per wave) and write it to `att_out` and into the next operand row.  The monotonic normalisers
// (las/model.py:157-164) run inside the shared step body; here they only get their per-step pointers.
// TWO (round 4): a second decoder cell (decoder_layers = 2, the reference's default depth) inside the launch.  p.wiring 0 = a
// MultiRNNCell inside the AttentionWrapper (las/model.py:194-200): per step G0 -> cell 0 -> barrier -> G1: z1 = [h0_t | h1_{t-1}] K1
// -> cell 1 + attention queried with h1_t; p.wiring 1 = AttentionMultiCell (--bottom_only, las/model.py:36-69): G0 -> cell 0 +
// attention queried with h0_t -> barrier -> G1: z1 = [attention_t | attention_{t-1} | h1_{t-1}] K1 -> cell 1.  G1's operand row is
// gathered chunk by chunk from where its pieces already lie (h0_t in s.h_out, attention_t in s.ctx_out, attention_{t-1} in the
// operand row of cell 0, h1_{t-1} in p.h1): no copies; K1's chunks are streamed from L2 at every step (at 2M + Hd columns in
// the --bottom_only wiring they would not fit the registers next to K0's).
template <bool SAMPLING, bool RES, int NTL_MAX = 2, int KCW_MAX = 12, int KRES = KCW_MAX, bool AL = false, int KAL_MAX = 10, bool TWO = false>
__global__ __launch_bounds__(256) void dec_persist_fwd_kernel(las_dec_persist p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const las_dec_step& s0 = p.s;
  const int B = s0.B, Hd = s0.Hd, M = s0.M, Tm = s0.Tm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  // blocks in chunks of 8 groups (block = chunk * 256 + member * 8 + group % 8): a group's 32 members are 8 blocks apart (one
  // XCD under round-robin dispatch), and the in-order dispatcher completes the 8 groups of a chunk -- one workgroup per CU of
  // a 256-CU device -- before it starts the next chunk: a batch of more than 64 utterances runs chunk after chunk
  const int groups = (B + 7) / 8;
  const int group = (blockIdx.x / (8 * P_MEMBERS)) * 8 + (blockIdx.x & 7), member = (blockIdx.x % (8 * P_MEMBERS)) >> 3;
  if (group >= groups) return;
  unsigned* status = reinterpret_cast<unsigned*>(p.workspace);
  pu64* flags = reinterpret_cast<pu64*>(reinterpret_cast<char*>(p.workspace) + 64) + (size_t)group * 2 * P_MEMBERS;
  pu64* xcc_tab = flags + P_MEMBERS;
  float* red = sm + (2 * Hd + Tm + 16 + 2048);             // [4 waves][16][CPM+1] partial z tiles
  const int RS = persist_red_stride(Hd);
  int* fail = reinterpret_cast<int*>(red + 4 * 16 * RS);
  int* colo = fail + 1;
  if (tid == 0) { *fail = 0; *colo = 0; }
  __syncthreads();

  // are the 32 members on one XCD?  (same handshake as the recurrent kernels; decides fences only)
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    __hip_atomic_store(xcc_tab + member, ((pu64)1 << 32) | (xcc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool same = true;
    for (int m = 0; m < P_MEMBERS; ++m) {
      pu64 v = 0;
      unsigned spins = 0;
      do {
        v = __hip_atomic_load(xcc_tab + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 32) == 1) break;
        __builtin_amdgcn_s_sleep(2);
      } while (++spins < P_SPIN_LIMIT);
      same = same && ((v >> 32) == 1) && ((unsigned)v == xcc + 1);
    }
    *colo = same ? 1 : 0;
  }
  __syncthreads();
  const bool local = *colo != 0;

  // G role: this member's columns of z and its register-resident slice of K ([4Hd, K_in] bf16, row = output column)
  const int CPM = 4 * Hd / P_MEMBERS, NTL = CPM / 16;      // columns per member, 16-column tiles
  const int KC = p.K_in / 32;                               // 32-deep K chunks; wave w takes chunks w, w+4, ...
  bf16x8 wf[NTL_MAX][KRES];
  const unsigned short* wrow[NTL_MAX];          // this lane's row of kT per column tile (the streamed chunks' source)
#pragma unroll
  for (int nt = 0; nt < NTL_MAX; ++nt) {
    wrow[nt] = p.kT + (int64_t)(member * CPM + min(nt, NTL - 1) * 16 + l15) * p.ldk + 8 * lq;
#pragma unroll
    for (int i = 0; i < KRES; ++i) {
      const int kc = wave + 4 * i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (nt < NTL && kc < KC) v = *reinterpret_cast<const uint4*>(wrow[nt] + kc * 32);
      wf[nt][i] = __builtin_bit_cast(bf16x8, v);
    }
  }
  const int bg = group * 8 + (l15 & 7);                    // utterance of A-fragment row l15 (rows 8..15 repeat 0..7)
  // A role (AL): this member's 16 columns of W_al^T [A, Hd + M], K chunks wave, wave + 4, ...
  bf16x8 wal[AL ? KAL_MAX : 1];
  const int KA = (Hd + M) / 32, NA = AL ? p.A / 16 : 0;
  if constexpr (AL) {
#pragma unroll
    for (int i = 0; i < KAL_MAX; ++i) {
      const int kc = wave + 4 * i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (member < NA && kc < KA) v = *reinterpret_cast<const uint4*>(p.walT + (int64_t)(member * 16 + l15) * p.ld_wal + kc * 32 + 8 * lq);
      wal[i] = __builtin_bit_cast(bf16x8, v);
    }
  }
  const int bs = group * 8 + member / 4, part = member & 3;  // S role
  // what the S role reads from the encoder memory never changes over the U steps: keep it in LDS when it fits
  const unsigned short* lkeys = nullptr;
  const unsigned short* lvals = nullptr;
  const int vres = RES ? persist_fwd_resident_frames(M, Hd, Tm) : 0;      // value frames in LDS (Tm unless the memory is long)
  if (RES && bs < B) {
    unsigned short* lk = reinterpret_cast<unsigned short*>(sm + persist_fwd_scratch_floats(Hd, Tm));
    const int fq = (Tm + 3) / 4, f0 = part * fq, f1 = min(Tm, f0 + fq);
    unsigned short* lv = lk + (size_t)fq * Hd;
    const int cols_per = persist_cols_per(M), c0 = part * cols_per, cn = max(0, min(M, c0 + cols_per) - c0);
    const unsigned short* gk = s0.keys + (int64_t)bs * Tm * Hd;
    const unsigned short* gv = s0.values + (int64_t)bs * Tm * M;
    for (int e = tid; e < max(f1 - f0, 0) * (Hd / 8); e += 256) {
      const int r = e / (Hd / 8), c = e % (Hd / 8);
      *reinterpret_cast<uint4*>(lk + (size_t)r * Hd + c * 8) = *reinterpret_cast<const uint4*>(gk + (int64_t)(f0 + r) * Hd + c * 8);
    }
    for (int e = tid; e < vres * (cn / 8); e += 256) {
      const int r = e / (cn / 8), c = e % (cn / 8);
      *reinterpret_cast<uint4*>(lv + (size_t)r * cols_per + c * 8) = *reinterpret_cast<const uint4*>(gv + (int64_t)r * M + c0 + c * 8);
    }
    lkeys = lk;
    lvals = lv;
  }
  __syncthreads();
  unsigned epoch = 0;

  // cell bias of this thread's unit: the same at every step
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  const bool pre_ok = s0.tok_rows != nullptr && tid < Hd && Hd <= 256;     // (one unit per thread; wider cells load in the body)
  if (pre_ok)
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = s0.bias[g * Hd + tid];

  for (int t = 0; t < p.U; ++t) {
    // the S role's cell operands that are already final (token id of this step -- teacher's, or drawn before the last
    // barrier --, its kernel row, c_{t-1}): requested now, used after the product and its barrier
    int tok_pre = 0;
    float tok4[4] = {0.f, 0.f, 0.f, 0.f}, cprev_pre = 0.f;
    if (pre_ok && bs < B) {
      tok_pre = __hip_atomic_load(s0.tok_ids + (int64_t)bs * s0.tok_stride + t * p.inc_tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int g = 0; g < 4; ++g) tok4[g] = las_bf2f(s0.tok_rows[(int64_t)tok_pre * 4 * Hd + g * Hd + tid]);
      cprev_pre = (s0.c_prev + t * p.inc_cprev)[(int64_t)bs * s0.ldcp + tid];
    }
    LAS_STAMP(t, 0);
    // ---- G: z_t[group's utterances, my columns] ----
    const unsigned xtag = (unsigned)(t + 1);
    pu64* const xbase = reinterpret_cast<pu64*>(reinterpret_cast<char*>(p.workspace) + 64) + persist_flag_words(B);
    const size_t ldsc = (size_t)((Tm + 31) / 32 * 32);
    pu64* const xzb = xbase + 2 * (size_t)B * ldsc;           // [2][B][4Hd] after the score granules [2][B][ldsc]
    {
      f32x4 acc[NTL_MAX];
#pragma unroll
      for (int nt = 0; nt < NTL_MAX; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned short* arow = p.x + (int64_t)min(bg, B - 1) * p.ldx + (int64_t)t * p.inc_x + 8 * lq;
      uint4 av[KRES];
#pragma unroll
      for (int i = 0; i < KRES; ++i) {
        const int kc = wave + 4 * i;
        av[i] = make_uint4(0, 0, 0, 0);
        if (kc < KC && bg < B) av[i] = *reinterpret_cast<const uint4*>(arow + kc * 32);
      }
#pragma unroll
      for (int i = 0; i < KRES; ++i)
#pragma unroll
        for (int nt = 0; nt < NTL_MAX; ++nt)      // (absent tiles / chunks hold zero weights: no guard -- 32 branches otherwise)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[i]), wf[nt][i], acc[nt], 0, 0, 0);
      LAS_STAMP(t, 11);
#ifndef DEC_SKIP_STREAM
      if constexpr (KRES < KCW_MAX) {
        // the chunks that do not fit the register file, SB at a time: their operand and weight fragments (SB (1 + NTL_MAX)
        // loads) are all requested before the first product.  (One chunk at a time under `#pragma unroll 2` the compiler,
        // short of VGPRs, had made every weight fragment its own load -> wait -> product: ~18 dependent L2 round trips, 6 of
        // the 8 us of this phase at 512 units.)  Chunks past the end of K: clamped address, zero operand.
        constexpr int SB = 4;
#pragma unroll 1
        for (int i0 = KRES; i0 < KCW_MAX; i0 += SB) {
          if (wave + 4 * i0 >= KC) break;
          uint4 a[SB], w[SB][NTL_MAX];
#pragma unroll
          for (int j = 0; j < SB; ++j) {
            const int kc = wave + 4 * (i0 + j), kcc = min(kc, KC - 1);
            a[j] = *reinterpret_cast<const uint4*>(arow + kcc * 32);
            if (kc >= KC || bg >= B) a[j] = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NTL_MAX; ++nt) w[j][nt] = *reinterpret_cast<const uint4*>(wrow[nt] + kcc * 32);
          }
#pragma unroll
          for (int j = 0; j < SB; ++j)
#pragma unroll
            for (int nt = 0; nt < NTL_MAX; ++nt)  // (an absent tile repeats the last one's weights; its sums are never stored)
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, w[j][nt]), acc[nt], 0, 0, 0);
        }
      }
#endif
      LAS_STAMP(t, 12);
#pragma unroll
      for (int nt = 0; nt < NTL_MAX; ++nt)
        if (nt < NTL)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(wave * 16 + lq * 4 + r) * RS + nt * 16 + l15] = acc[nt][r];
      LAS_STAMP(t, 1);
      __syncthreads();
      for (int e = tid; e < 8 * CPM; e += 256) {
        const int row = e / CPM, col = e % CPM;
        const int b = group * 8 + row;
        if (b < B)
          pgranule_store(xzb + ((size_t)(xtag & 1) * B + b) * 4 * Hd + member * CPM + col, xtag,
                         red[(0 * 16 + row) * RS + col] + red[(1 * 16 + row) * RS + col] + red[(2 * 16 + row) * RS + col] + red[(3 * 16 + row) * RS + col],
                         local);
      }
    }
    // (no group barrier here: the S role polls the granules of its utterance)
    LAS_STAMP(t, 2);

    // ---- S: cell + attention of utterance bs, context columns of `part` ----
    las_dec_step st = s0;
    {
      const bool last = (t + 1 == p.U);
      st.tok_ids = s0.tok_ids + t * p.inc_tok;
      st.c_prev = s0.c_prev + t * p.inc_cprev;
      st.gates_out = s0.gates_out + t * p.inc_gates;
      st.c_out = s0.c_out + t * p.inc_cout;
      st.h_out = s0.h_out + t * p.inc_h;
      st.h_out2 = last ? nullptr : s0.h_out2 + t * p.inc_h2;
      st.align_out = s0.align_out + t * p.inc_align;
      st.align_bf16 = s0.align_bf16 ? s0.align_bf16 + t * p.inc_align : nullptr;
      st.pq_out = s0.pq_out ? s0.pq_out + t * p.inc_pq : nullptr;
      st.ctx_out = s0.ctx_out + t * p.inc_ctx;
      st.ctx_out2 = (last || !s0.ctx_out2) ? nullptr : s0.ctx_out2 + t * p.inc_ctx2;
      st.step = t;
      if (s0.norm != LAS_NORM_SOFTMAX) {      // monotonic normalisers: alignments of the step before, p_choose for the backward
        st.prev_align = t > 0 ? s0.align_out + (t - 1) * p.inc_align : nullptr;
        st.ldpa = s0.lda;
        st.p_out = s0.p_out ? s0.p_out + t * p.inc_p : nullptr;
      }
    }
    pu64* const xsc = xbase + ((size_t)(xtag & 1) * B + (bs < B ? bs : 0)) * ldsc;
    pu64* const xz = xzb + ((size_t)(xtag & 1) * B + (bs < B ? bs : 0)) * 4 * Hd;
    if constexpr (!TWO) {
      if (bs < B) {
        PersistHook hook{xsc, xz, xtag, flags, member, &epoch, local, fail, p.wq_packed, lkeys, lvals,
                         pre_ok, tok_pre, {tok4[0], tok4[1], tok4[2], tok4[3]}, {bias4[0], bias4[1], bias4[2], bias4[3]}, cprev_pre};
        hook.vres = vres;
        dec_step_fwd_body<RES>(st, bs, part, 4, sm, &hook);     // (a timed-out poll leaves through the barrier below)
      }
    } else {
      // the second cell's step: no token rows, its own bias / state / saved gates; the attention fields are cell 0's (one attention)
      las_dec_step st1 = st;
      st1.tok_rows = nullptr;
      st1.bias = p.bias1;
      st1.c_prev = p.c1 + (int64_t)t * Hd;            st1.ldcp = (int64_t)(p.U + 1) * Hd;
      st1.c_out = p.c1 + (int64_t)(t + 1) * Hd;       st1.ldco = (int64_t)(p.U + 1) * Hd;
      st1.gates_out = p.gates1 + (int64_t)t * 4 * Hd; st1.ldg = (int64_t)p.U * 4 * Hd;
      st1.h_out = p.h1 + (int64_t)(t + 1) * Hd;       st1.ldh = (int64_t)(p.U + 1) * Hd;
      st1.h_out2 = nullptr;
      // input dropout: cell 0's token row is scaled in its cell (stA keeps drop_keep); its attention feed is written masked
      // with step t+1's draws by whichever body writes ctx_out2 (feed_plain 2); cell 1's row is masked where G1 reads it
      las_dec_step stA = st;
      stA.feed_plain = st1.feed_plain = 2;
      stA.feed_stream0 = st1.feed_stream0 = p.in_stream0;
      st1.feed_width = p.win0;       // (cell 1 has no token rows: the body takes the feed's mask window from here; cell 0 keeps
                                     //  the width its token-row draws are indexed with)
      if (p.wiring == 0) {           // cell 0 alone now; the attention runs with cell 1
        stA.mode = LAS_DEC_CELL_ONLY;
        stA.ctx_out2 = nullptr;
        st1.mode = LAS_DEC_FUSED;
      } else {                       // cell 0 with the attention now; cell 1 alone afterwards
        stA.mode = LAS_DEC_FUSED;
        st1.mode = LAS_DEC_CELL_ONLY;
        st1.ctx_out2 = nullptr;
        st1.drop_keep = 1.0f;
      }
      // one instance of the step body for both cells (a loop, not two inlined copies: the second copy had pushed the kernel 150
      // VGPRs over the register file, into scratch)
      bool dead = false;
#pragma nounroll
      for (int cell = 0; cell < 2; ++cell) {
        if (cell == 1) {
          // every piece of G1's operand rows (h0_t, attention_t of all 8 utterances) is in memory behind this barrier
          LAS_STAMP(t, 13);
          if (!persist_barrier(flags, member, ++epoch, local, fail)) { dead = true; break; }
          LAS_STAMP(t, 14);
        // ---- G1: z1_t[group's utterances, my columns] = [pieces] K1, all chunks streamed ----
        pu64* const xz1b = xzb + 2 * (size_t)B * 4 * Hd;
        {
          f32x4 acc[NTL_MAX];
  #pragma unroll
          for (int nt = 0; nt < NTL_MAX; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int64_t bgc = min(bg, B - 1);
          const int KC1 = p.K1_in / 32, nH = Hd / 32, nM = M / 32;
          // this lane's row of each piece (8 lq = its 16 bytes of a 32-deep chunk)
          const unsigned short* r_h0 = s0.h_out + bgc * s0.ldh + (int64_t)t * p.inc_h + 8 * lq;                    // h0_t
          const unsigned short* r_h1 = p.h1 + (bgc * (p.U + 1) + t) * Hd + 8 * lq;                                  // h1_{t-1}
          const unsigned short* r_at = s0.ctx_out + bgc * s0.ldc + (int64_t)t * p.inc_ctx + 8 * lq;                // attention_t
          const unsigned short* r_ap = s0.ctx_out + bgc * s0.ldc + (int64_t)max(t - 1, 0) * p.inc_ctx + 8 * lq;    // attention_{t-1} (unmasked; zeros at t = 0)
          auto piece = [&](int kc) -> const unsigned short* {
            if (p.wiring == 0) return kc < nH ? r_h0 + kc * 32 : r_h1 + (kc - nH) * 32;
            if (kc < nM) return r_at + kc * 32;
            if (kc < 2 * nM) return r_ap + (kc - nM) * 32;
            return r_h1 + (kc - 2 * nM) * 32;
          };
          const unsigned short* w1row[NTL_MAX];
  #pragma unroll
          for (int nt = 0; nt < NTL_MAX; ++nt) w1row[nt] = p.k1T + (int64_t)(member * CPM + min(nt, NTL - 1) * 16 + l15) * p.ldk1 + 8 * lq;
          constexpr int SB = 4;
  #pragma unroll 1
          for (int i0 = 0; wave + 4 * i0 < KC1; i0 += SB) {
            uint4 a[SB], w[SB][NTL_MAX];
  #pragma unroll
            for (int j = 0; j < SB; ++j) {
              const int kc = wave + 4 * (i0 + j), kcc = min(kc, KC1 - 1);
              a[j] = *reinterpret_cast<const uint4*>(piece(kcc));
              if (kc >= KC1 || bg >= B || (t == 0 && p.wiring == 1 && kc >= nM && kc < 2 * nM)) a[j] = make_uint4(0, 0, 0, 0);
              if (s0.drop_keep < 1.0f && kcc * 32 + 8 * lq < p.win1)       // input dropout of cell 1 (columns [0, win1) of its row)
                a[j] = drop8(a[j], s0.drop_seed, p.in_stream1 + (unsigned)t, (unsigned long long)bgc * p.win1 + kcc * 32 + 8 * lq, s0.drop_keep, 1.0f / s0.drop_keep);
  #pragma unroll
              for (int nt = 0; nt < NTL_MAX; ++nt) w[j][nt] = *reinterpret_cast<const uint4*>(w1row[nt] + kcc * 32);
            }
  #pragma unroll
            for (int j = 0; j < SB; ++j)
  #pragma unroll
              for (int nt = 0; nt < NTL_MAX; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, w[j][nt]), acc[nt], 0, 0, 0);
          }
          __syncthreads();                         // (the S role above is done with the LDS scratch `red` lies behind)
  #pragma unroll
          for (int nt = 0; nt < NTL_MAX; ++nt)
            if (nt < NTL)
  #pragma unroll
              for (int r = 0; r < 4; ++r) red[(wave * 16 + lq * 4 + r) * RS + nt * 16 + l15] = acc[nt][r];
          __syncthreads();
          for (int e = tid; e < 8 * CPM; e += 256) {
            const int row = e / CPM, col = e % CPM;
            const int b = group * 8 + row;
            if (b < B)
              pgranule_store(xz1b + ((size_t)(xtag & 1) * B + b) * 4 * Hd + member * CPM + col, xtag,
                             red[(0 * 16 + row) * RS + col] + red[(1 * 16 + row) * RS + col] + red[(2 * 16 + row) * RS + col] + red[(3 * 16 + row) * RS + col],
                             local);
          }
        }
        }
        if (cell == 1) LAS_STAMP(t, 15);
        if (bs < B) {
          const bool c0 = cell == 0;
          const las_dec_step& stc = c0 ? stA : st1;
          pu64* const xzc = c0 ? xz : xzb + 2 * (size_t)B * 4 * Hd + ((size_t)(xtag & 1) * B + bs) * 4 * Hd;
          PersistHook hookc{xsc, xzc, xtag, flags, member, &epoch, local, fail, p.wq_packed, lkeys, lvals,
                            c0 && pre_ok, c0 ? tok_pre : 0, {tok4[0], tok4[1], tok4[2], tok4[3]}, {bias4[0], bias4[1], bias4[2], bias4[3]}, cprev_pre};
          hookc.vres = vres;
          dec_step_fwd_body<RES>(stc, bs, part, 4, sm, &hookc);
        }
      }
      if (dead) break;
    }
    if constexpr (SAMPLING) {
      // ---- scheduled sampling (utils/training_helper.py:48-87): logits_t = context_t W_proj + b from the four parts'
      //      partial products, then the next fed token = Categorical(logits_t) with probability p, else the teacher's ----
      // The draws are counter-based, so every member knows which utterances of its group are selected at this step:
      // a group without a selection skips the phase and both of its barriers (57 % of the steps at p = 0.1), and only
      // the selected utterances' workgroups compute logits.  (The logits the model returns come from one GEMM after
      // the loop, as without sampling.)
      const int V = p.V, Vp = p.Vp;
      bool any_sel = false, my_sel = false;
      if (t + 1 < p.U) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int bj = group * 8 + j;
          const bool sel = bj < B && las_uniform(p.seed, 0x5e1ec7u, (unsigned long long)t * B + bj) < p.sampling_prob;
          any_sel = any_sel || sel;
          if (bj == bs) my_sel = sel;
        }
      }
      if (any_sel) {
      float* plog_t = p.plog + ((int64_t)t * B + (bs < B ? bs : 0)) * 4 * Vp;
      if (my_sel) {
        // the decoder's output row: the context (this part's columns, as it wrote them), or -- second cell in the --bottom_only
        // wiring -- h1_t, which every part holds in LDS (hq of the cell it just ran)
        const bool out_h1 = TWO && p.wiring == 1;
        const int cols = (out_h1 ? Hd : M) / 4, c0 = part * cols;
        float* cx = out_h1 ? sm + c0 : sm;                // [cols] my columns of the output as floats
        __builtin_amdgcn_s_waitcnt(0x0070);               // my context stores are done
        __syncthreads();
        if (!out_h1) {
          const unsigned short* crow = s0.ctx_out + (int64_t)bs * s0.ldc + (int64_t)t * p.inc_ctx + c0;
          for (int c = tid; c < cols; c += 256)
            cx[c] = las_bf2f(__builtin_bit_cast(unsigned short, (unsigned short)__hip_atomic_load(
                        reinterpret_cast<const unsigned short*>(crow) + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
          __syncthreads();
        }
        for (int v = tid; v < Vp; v += 256) {
          float acc = 0.f;
          if (v < V) {
            const unsigned short* wrow = p.wprojT + (int64_t)v * p.ldw + c0;
            for (int c = 0; c < cols; c += 8) {
              const uint4 w = *reinterpret_cast<const uint4*>(wrow + c);
              acc += dot8(w, cx + c);
            }
          }
          plog_t[part * Vp + v] = acc;
        }
      }
      if (!persist_barrier(flags, member, ++epoch, local, fail)) break;
      if (my_sel && part == 0) {
        float* lg = sm;                                   // [Vp]
        for (int v = tid; v < Vp; v += 256)
          lg[v] = v < V ? plog_t[v] + plog_t[Vp + v] + plog_t[2 * Vp + v] + plog_t[3 * Vp + v] + p.bproj[v] : p.bproj[v];
        __syncthreads();
        if (tid < 64) {
          const unsigned long long sidx = (unsigned long long)t * B + bs;
          int out;
          {
            float best = -INFINITY;
            int arg = 0;
            for (int v = lane; v < V; v += 64) {
              const float u = fmaxf(las_uniform(p.seed, 0x9a3b1eu, sidx * V + v), 1e-12f);
              const float gmb = lg[v] - __logf(-__logf(u));
              if (gmb > best) { best = gmb; arg = v; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
              const float ob = __shfl_xor(best, o, 64);
              const int oa = __shfl_xor(arg, o, 64);
              if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
            }
            out = arg;
          }
          if (lane == 0) const_cast<int32_t*>(s0.tok_ids)[(int64_t)bs * s0.tok_stride + t + 1] = out;   // else: the teacher's, already there
          if constexpr (TWO) {
            if (p.emb) {
              // dense token feed (embedding_size > 0 under input dropout): the host filled the operand rows with the teacher's
              // embedded tokens; a sampled token replaces the next row's token columns (through that step's input mask)
              const float kp = s0.drop_keep;
              for (int c = lane; c < p.T0; c += 64) {
                float v = las_bf2f(p.emb[(int64_t)out * p.ld_emb + c]);
                if (kp < 1.0f)
                  v = las_uniform(s0.drop_seed, p.in_stream0 + (unsigned)(t + 1), (unsigned long long)bs * p.win0 + c) < kp ? v * (1.0f / kp) : 0.f;
                const_cast<unsigned short*>(p.x)[(int64_t)bs * p.ldx + (int64_t)(t + 1) * p.inc_x + c] = las_f2bf(v);
              }
            }
          }
        }
      }
      }
    }
    if constexpr (AL) {
      // ---- A: attention_t = [h_t | context_t] W_al for the group's utterances (every part's context columns and h_t are
      //      in memory behind this barrier; their lines were never read by this workgroup before) ----
      if (!persist_barrier(flags, member, ++epoch, local, fail)) break;
      if (member < NA) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const int64_t bgc = min(bg, B - 1);
        const unsigned short* hrow = s0.h_out + bgc * s0.ldh + (int64_t)t * p.inc_h + 8 * lq;
        const unsigned short* crow = s0.ctx_out + bgc * s0.ldc + (int64_t)t * p.inc_ctx + 8 * lq;
        uint4 av[KAL_MAX];
#pragma unroll
        for (int i = 0; i < KAL_MAX; ++i) {
          const int kc = wave + 4 * i;
          av[i] = make_uint4(0, 0, 0, 0);
          if (kc < KA && bg < B) av[i] = *reinterpret_cast<const uint4*>(kc < Hd / 32 ? hrow + kc * 32 : crow + (kc - Hd / 32) * 32);
        }
#pragma unroll
        for (int i = 0; i < KAL_MAX; ++i)
          if (wave + 4 * i < KA) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[i]), wal[i], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 16 + lq * 4 + r) * RS + l15] = acc[r];
        __syncthreads();
        if (tid < 8 * 16) {
          const int row = tid >> 4, col = tid & 15;
          const int b = group * 8 + row;
          if (b < B) {
            const unsigned short v = las_f2bf(red[(0 * 16 + row) * RS + col] + red[(1 * 16 + row) * RS + col] + red[(2 * 16 + row) * RS + col] +
                                              red[(3 * 16 + row) * RS + col]);
            p.att_out[(int64_t)b * p.ld_att + (int64_t)t * p.A + member * 16 + col] = v;
            if (t + 1 < p.U)
              const_cast<unsigned short*>(p.x)[(int64_t)b * p.ldx + (int64_t)(t + 1) * p.inc_x + p.x_att_off + member * 16 + col] = v;
          }
        }
      }
    }
    if (!persist_barrier(flags, member, ++epoch, local, fail)) break;
    LAS_STAMP(t, 10);
  }
  if (*fail && tid == 0) atomicOr(status, 8u);
}

// ------------------------------------------------------------------------------------------------
// The same launch with the S role WRITTEN OUT for the case the train step runs (teacher forcing without scheduled sampling,
// softmax attention Luong or Bahdanau, decoder_units <= 256, keys / values resident in LDS): dec_step_fwd_body serves every
// mode of the step kernels and, inlined into the persistent loop, cost 6 600 instructions for the score phase alone, 420
// spilled scalar registers and a full vmcnt(0) drain in front of most phases (phase stamps of the diagnostics build,
// scripts/gpu_dec_stamps.py: 13.9 us per step, of which scores 2.9, context 2.7, operands + product 3.0, softmax 1.4).
// Differences from the general body, none of them visible in the results' layout:
//   * the attention type is a template parameter; the Luong score is v_dot2c_f32_bf16 on (key, h) bf16 pairs (h_t is
//     bf16-rounded before it is a query in every path, so the products are the same numbers);
//   * c_t stays in a register (all four workgroups of an utterance compute the same cell), the token's kernel row and id
//     of step t+1 are requested during step t (the ids are the teacher's: known up front);
//   * the saved tensors are written by four workgroups instead of one: part p writes gate p, parts 0 / 1 / 2 write
//     c_t / h_t / the operand copy of h_t, part 3 the processed query, and every part its own frames of the alignments;
//   * the product's operand loads are issued before anything else in the step; LDS key rows are padded against bank
//     conflicts (4 lanes per frame, 16 frames per wave).
// ------------------------------------------------------------------------------------------------
constexpr int LEAN_KPAD = 32;                       // bf16 elements of padding per LDS key row (row stride = 64 mod 128 bytes)
struct LeanLayout { size_t hq, pq, vq, sc, red, cred, zred, flags, hqb, tok, psb, zero, keys, vals, total_bytes; int fq, kst, cols_per, kp, ts, cp; };
__host__ __device__ inline LeanLayout lean_layout(int Hd, int Tm, int M, int U) {
  LeanLayout l;
  size_t o = 0;                                     // in floats
  l.hq = o; o += Hd;
  l.pq = o; o += Hd;
  l.vq = o; o += Hd;
  l.sc = o; o += (size_t)((Tm + 3) & ~3);
  l.red = o; o += 16;
  l.cred = o; o += 2048;
  l.zred = o; o += (size_t)4 * 8 * persist_red_stride(Hd);      // [4 waves][8 utterances][columns per member + 1]
  l.flags = o; o += 8;
  l.hqb = o; o += Hd / 2;
  l.tok = o; o += (size_t)((U + 3) & ~3);            // the utterance's fed token ids, all steps
  o = (o + 3) & ~(size_t)3;
  // the alignments as two bf16 rows (high and low halves; zeros up to a whole number of 32-frame chunks) and 16 B of zeros:
  // the A operand of the context product on the matrix cores
  l.kp = (Tm + 31) / 32 * 32;
  l.psb = o; o += (size_t)l.kp;
  l.zero = o; o += 4;
  l.fq = (Tm + 3) / 4;
  l.kst = Hd + LEAN_KPAD;
  l.cols_per = persist_cols_per(M);
  l.keys = o; o += ((size_t)l.fq * l.kst + 1) / 2;
  o = (o + 3) & ~(size_t)3;
  // the values TRANSPOSED, [cp columns][ts frames]: a column's frames are contiguous, as a B fragment reads them (ts: the
  // frames rounded up to an odd number of 16-byte pieces -- the 16 columns of a fragment on 16 different bank groups;
  // cp: whole 64-column shares of the four waves)
  l.ts = (Tm + 7) & ~7;
  if (((l.ts / 8) & 1) == 0) l.ts += 8;
  l.cp = (l.cols_per + 63) & ~63;
  l.vals = o; o += ((size_t)l.cp * l.ts + 1) / 2;
  l.total_bytes = ((o + 3) & ~(size_t)3) * sizeof(float);
  return l;
}
__host__ __device__ inline bool persist_fwd_lean_ok(int Hd, int M, int Tm, int U, int att, int norm) {
  return (att == LAS_ATT_LUONG || att == LAS_ATT_BAHDANAU) && norm == LAS_NORM_SOFTMAX && (Hd == 128 || Hd == 256 || Hd == 512) && M % 32 == 0 &&
         lean_layout(Hd, Tm, M, U).total_bytes <= 158 * 1024;
}

typedef __attribute__((ext_vector_type(2))) __bf16 las_bf16x2;
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(las_bf16x2, a), __builtin_bit_cast(las_bf16x2, b), acc, false);
}

// NTL: 16-column tiles of z per member (decoder_units / 128: 1, 2 or 4).  KRES of the KCWM 32-deep K chunks a wave owns stay in
// registers as MFMA B fragments; the chunks beyond (512 units: 12 of 20) are streamed from L2 at every step, three in flight.
template <int ATT, int NTL, int KRES = 12, int KCWM = 12, bool SAMPLING = false>
__global__ __launch_bounds__(256) void dec_persist_fwd_lean_kernel(las_dec_persist p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int UPT = NTL == 4 ? 2 : 1;                    // hidden units per thread (512 units: tid and tid + 256)
  constexpr int NK = NTL * 4;                              // 16-byte key pieces per lane and frame (4 lanes per frame)
  const int B = p.s.B, Hd = p.s.Hd, M = p.s.M, Tm = p.s.Tm, U = p.U;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int groups = (B + 7) / 8;
  const int group = (blockIdx.x / (8 * P_MEMBERS)) * 8 + (blockIdx.x & 7), member = (blockIdx.x % (8 * P_MEMBERS)) >> 3;
  if (group >= groups) return;
  unsigned* status = reinterpret_cast<unsigned*>(p.workspace);
  pu64* flags = reinterpret_cast<pu64*>(reinterpret_cast<char*>(p.workspace) + 64) + (size_t)group * 2 * P_MEMBERS;
  pu64* xcc_tab = flags + P_MEMBERS;
  const LeanLayout L = lean_layout(Hd, Tm, M, U);
  float* hq = sm + L.hq;
  float* pq = sm + L.pq;
  float* vq = sm + L.vq;
  float* sc = sm + L.sc;
  float* red = sm + L.red;
  float* cred = sm + L.cred;
  float* zred = sm + L.zred;
  int* fail = reinterpret_cast<int*>(sm + L.flags);
  int* colo = fail + 1;
  unsigned short* hqb = reinterpret_cast<unsigned short*>(sm + L.hqb);
  int* ltok = reinterpret_cast<int*>(sm + L.tok);
  unsigned short* lk = reinterpret_cast<unsigned short*>(sm + L.keys);
  unsigned short* lv = reinterpret_cast<unsigned short*>(sm + L.vals);
  unsigned short* psb = reinterpret_cast<unsigned short*>(sm + L.psb);
  const int RS = persist_red_stride(Hd);
  for (int i = tid; i < 2 * L.kp + 8; i += 256) psb[i] = 0;            // (the 16 zero bytes lie right behind)
  if (tid == 0) { *fail = 0; *colo = 0; }
  __syncthreads();
  if (tid == 0) {                                          // are the 32 members on one XCD?  (decides fences only)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    __hip_atomic_store(xcc_tab + member, ((pu64)1 << 32) | (xcc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool same = true;
    for (int m = 0; m < P_MEMBERS; ++m) {
      pu64 v = 0;
      unsigned spins = 0;
      do {
        v = __hip_atomic_load(xcc_tab + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 32) == 1) break;
        __builtin_amdgcn_s_sleep(2);
      } while (++spins < P_SPIN_LIMIT);
      same = same && ((v >> 32) == 1) && ((unsigned)v == xcc + 1);
    }
    *colo = same ? 1 : 0;
  }
  __syncthreads();
  const bool local = *colo != 0;
#ifdef LAS_STAMPS
  if (tid == 0 && (blockIdx.x % 37) == 0) atomicAdd(&las_stamps[255 * 16 + (local ? 15 : 14)], 1ull);   // diagnostics: groups on one XCD / not
#endif

  // G role: this member's columns of z and its register-resident slice of K ([4Hd, K_in] bf16, row = output column)
  constexpr int CPM = NTL * 16;
  const int KC = p.K_in / 32;
  bf16x8 wf[NTL][KRES];                                    // chunks past the end of K: zero weights (their products add nothing)
  const unsigned short* wrow[NTL];                         // this lane's row of kT per column tile (the streamed chunks' source)
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) {
    wrow[nt] = p.kT + (int64_t)(member * CPM + nt * 16 + l15) * p.ldk + 8 * lq;
#pragma unroll
    for (int i = 0; i < KRES; ++i) {
      const int kc = wave + 4 * i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kc < KC) v = *reinterpret_cast<const uint4*>(wrow[nt] + kc * 32);
      wf[nt][i] = __builtin_bit_cast(bf16x8, v);
    }
  }
  const int bg = group * 8 + (l15 & 7);                    // utterance of A-fragment row l15 (rows 8..15 repeat 0..7)
  const int bs = group * 8 + member / 4, part = member & 3;  // S role: utterance, part
  const bool have = bs < B;
  const int bsc = have ? bs : 0;
  // S role constants
  const int fq = L.fq, f0 = part * fq, f1 = min(Tm, f0 + fq), KST = L.kst;
  const int cols_per = L.cols_per, c_begin = part * cols_per, ncols = max(0, min(M, c_begin + cols_per) - c_begin);
  const int KP = L.kp, TS = L.ts, CP = L.cp;
  const int len = have ? min(p.s.mem_len[bsc], Tm) : 0;
  const int flen = min(len, f1);
  if (have) {                                              // this workgroup's share of the encoder memory: resident for all steps
    const unsigned short* gk = p.s.keys + (int64_t)bs * Tm * Hd;
    const unsigned short* gv = p.s.values + (int64_t)bs * Tm * M;
    for (int e = tid; e < max(f1 - f0, 0) * (Hd / 8); e += 256) {
      const int r = e / (Hd / 8), c = e % (Hd / 8);
      *reinterpret_cast<uint4*>(lk + (size_t)r * KST + c * 8) = *reinterpret_cast<const uint4*>(gk + (int64_t)(f0 + r) * Hd + c * 8);
    }
    // (consecutive lanes take consecutive frames of the same 8 columns: the 2-byte LDS stores of a wave fall on 32 banks;
    // frames past the utterance and columns past its share are zeros -- their products must add nothing)
    for (int e = tid; e < TS * (CP / 8); e += 256) {
      const int r = e % TS, c = e / TS;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < len && c * 8 < ncols) v = *reinterpret_cast<const uint4*>(gv + (int64_t)r * M + c_begin + c * 8);
      const unsigned short* e8 = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
      for (int j = 0; j < 8; ++j) lv[(size_t)(c * 8 + j) * TS + r] = e8[j];
    }
    if (ATT == LAS_ATT_BAHDANAU)
      for (int u = tid; u < Hd; u += 256) vq[u] = p.s.att_v[u];
  }
  // the fed token ids of all steps (the teacher's: known up front).  From LDS the next step's id costs an LDS read; as a global
  // load it was waited for at once (a workgroup-uniform value goes to a scalar register): an L2 round trip in every step
  for (int i = tid; i < U; i += 256) ltok[i] = p.s.tok_ids[(int64_t)bsc * p.s.tok_stride + (int64_t)i * p.inc_tok];
  unsigned epoch = 0;
  const bool unit = tid < Hd && have;                      // this thread owns hidden units tid (+ 256) of utterance bs
  float bias4[UPT][4], c_reg[UPT];
  int tok_cur = p.s.tok_ids[(int64_t)bsc * p.s.tok_stride];          // (every thread: the loads of the loop are unconditional)
#pragma unroll
  for (int q = 0; q < UPT; ++q) {
    c_reg[q] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[q][g] = 0.f;
    if (unit) {
#pragma unroll
      for (int g = 0; g < 4; ++g) bias4[q][g] = p.s.bias[g * Hd + tid + q * 256];
      c_reg[q] = p.s.c_prev[(int64_t)bs * p.s.ldcp + tid + q * 256];
    }
  }
  const float keep = p.s.drop_keep;
  const int sub = lane & 3, fr = lane >> 2;                // score phase: 4 lanes per frame, 16 frames per wave
  pu64* const xbase = reinterpret_cast<pu64*>(reinterpret_cast<char*>(p.workspace) + 64) + persist_flag_words(B);
  const size_t ldsc = (size_t)((Tm + 31) / 32 * 32);
  pu64* const xzb = xbase + 2 * (size_t)B * ldsc;           // [2][B][4Hd] after the score granules [2][B][ldsc]
  pu64* const xpqb = xzb + 2 * (size_t)B * 4 * Hd;          // [2][B][Hd]: the four quarters of the processed query (Bahdanau)
  __syncthreads();

  for (int t = 0; t < U; ++t) {
    LAS_STAMP(t, 0);
    const bool last = (t + 1 == U);
    const unsigned xtag = (unsigned)(t + 1);
    // ---- G: z_t[group's utterances, my columns]; its operand row first, everything else of the step behind it ----
    // (no branches around the loads: chunks past the end of K re-read the last one against zero weights, rows of absent
    //  utterances re-read the last utterance's and are dropped below; a branch would cost a vmcnt(0) at its join)
    // Rows 8..15 of the 16-row MFMA tile carry nothing (8 utterances per group; their outputs are never read): those
    // lanes all read one 16-byte piece of zeros (workspace header, words 4..7: never written) instead of a second copy of
    // the operand rows.
    const unsigned short* arow = (l15 < 8) ? p.x + (int64_t)min(bg, B - 1) * p.ldx + (int64_t)t * p.inc_x + 8 * lq
                                           : reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(p.workspace) + 16);
    const int astep = (l15 < 8) ? 32 : 0;
    uint4 av[KRES];
#pragma unroll
    for (int i = 0; i < KRES; ++i) av[i] = *reinterpret_cast<const uint4*>(arow + min(wave + 4 * i, KC - 1) * astep);
    // the streamed chunks: operand piece and NTL weight pieces each, SD of them in flight
    constexpr int NS = KCWM - KRES, SD = 4;
    uint4 sa[NS > 0 ? SD : 1], sw[NS > 0 ? SD : 1][NTL];
    auto stream_issue = [&](int slot, int i) {              // i: chunk index of this wave (KRES ...)
#ifdef LEAN_STREAM_SAME                                      // (diagnostics: every streamed chunk re-reads the first one -- wrong results, L1 hits)
      const int kcc = min(wave + 4 * KRES, KC - 1) + 0 * i;
#else
      const int kcc = min(wave + 4 * i, KC - 1);
#endif
      sa[slot] = *reinterpret_cast<const uint4*>(arow + kcc * astep);
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) sw[slot][nt] = *reinterpret_cast<const uint4*>(wrow[nt] + kcc * 32);
    };
    if constexpr (NS > 0) {
#pragma unroll
      for (int q = 0; q < SD; ++q) stream_issue(q, KRES + q);
    }
    // the token's row of the cell kernel (its id was requested a step ago; used in the cell phase), and the next step's id
    unsigned short tokraw[UPT][4];
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
      const unsigned short* trow = p.s.tok_rows + (int64_t)tok_cur * 4 * Hd + min(tid + q * 256, Hd - 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) tokraw[q][g] = trow[g * Hd];
    }
    // every load of the step's head is in flight before the first product instruction (left to itself the scheduler
    // re-used four registers for the twelve operand pieces: three dependent round trips to L2 instead of one)
    __builtin_amdgcn_sched_barrier(0);
    {
      f32x4 acc[NTL];
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KRES; ++i)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[i]), wf[nt][i], acc[nt], 0, 0, 0);
      LAS_STAMP(t, 11);
      if constexpr (NS > 0) {
        // KCWM is the instantiation's bound, not the model's K: at metric-L (K = 1536) 4 of the 12 streamable chunks of a wave
        // exist, and walking all 12 three at a time was 4 L2 round trips of clamped loads (5 of the 8 us of this phase).  When
        // the chunks that exist were all requested above, their products are all that is left (scalar branch).
        const int nreal = __builtin_amdgcn_readfirstlane((KC - (tid >> 6) + 3) / 4) - KRES;
        if (nreal <= SD) {
#pragma unroll
          for (int q = 0; q < SD; ++q) {
            const uint4 a4 = q < nreal ? sa[q] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, sw[q][nt]), acc[nt], 0, 0, 0);
          }
        } else
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int slot = i % SD;
          const bool in = wave + 4 * (KRES + i) < KC;                         // past the end of K: nothing to add
          const uint4 a4 = in ? sa[slot] : make_uint4(0, 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NTL; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, sw[slot][nt]), acc[nt], 0, 0, 0);
          if (i + SD < NS) stream_issue(slot, KRES + i + SD);
        }
      }
      LAS_STAMP(t, 12);
      if (lq < 2) {                                        // rows 0..7 of the tile: the group's 8 utterances
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) zred[(wave * 8 + lq * 4 + r) * RS + nt * 16 + l15] = acc[nt][r];
      }
      LAS_STAMP(t, 1);
      lds_barrier();
      for (int e = tid; e < 8 * CPM; e += 256) {
        const int row = e / CPM, col = e % CPM;
        const int b = group * 8 + row;
        if (b < B)
          pgranule_store(xzb + ((size_t)(xtag & 1) * B + b) * 4 * Hd + member * CPM + col, xtag,
                         zred[(0 * 8 + row) * RS + col] + zred[(1 * 8 + row) * RS + col] + zred[(2 * 8 + row) * RS + col] + zred[(3 * 8 + row) * RS + col],
                         local);
      }
    }
    LAS_STAMP(t, 2);
    const int tok_next = ltok[min(t + 1, U - 1)];            // the next step's token id

    // ---- S: cell + attention of utterance bs, frames / context columns of `part` ----
    if (have) {
      pu64* const xsc = xbase + ((size_t)(xtag & 1) * B + bs) * ldsc;
      const pu64* const xz = xzb + ((size_t)(xtag & 1) * B + bs) * 4 * Hd;
      if (tid < Hd) {                                      // whole waves: Hd is a multiple of 64
        float tok_scale = 1.0f;                            // DropoutWrapper on the one-hot feed: its single entry is kept or lost
        if (keep < 1.0f)
          tok_scale = las_uniform(p.s.drop_seed, p.s.drop_stream, ((unsigned long long)t * B + bs) * p.s.feed_width + tok_cur) < keep
                          ? 1.0f / keep : 0.f;
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
          const int u = tid + q * 256;
          float z[4];
          unsigned spins = 0;
          for (;;) {                                       // the 32 product slices arrive as granules: wave-uniform, bounded
            pu64 gq[4];
            bool got = true;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              gq[g] = pgranule_load(xz + g * Hd + u);
              got = got && ((unsigned)(gq[g] >> 32) == xtag);
            }
            if (__all(got)) {
#pragma unroll
              for (int g = 0; g < 4; ++g) z[g] = __uint_as_float((unsigned)gq[g]);
              break;
            }
            if (++spins > P_SPIN_LIMIT || *fail) { *fail = 1; z[0] = z[1] = z[2] = z[3] = 0.f; break; }
            __builtin_amdgcn_s_sleep(1);
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) z[g] += bias4[q][g] + tok_scale * las_bf2f(tokraw[q][g]);
          const float gi = las_sigmoid(z[0]), gj = las_tanh(z[1]), gf = las_sigmoid(z[2] + 1.0f), go = las_sigmoid(z[3]);
          const float cn = gf * c_reg[q] + gi * gj;
          c_reg[q] = cn;
          const unsigned short hb = las_f2bf(go * las_tanh(cn));
          hq[u] = las_bf2f(hb);
          hqb[u] = hb;
          // saved for the backward pass: one quarter per workgroup of the utterance
          const float gsel = part == 0 ? gi : (part == 1 ? gj : (part == 2 ? gf : go));
          p.s.gates_out[(int64_t)bs * p.s.ldg + (int64_t)t * p.inc_gates + part * Hd + u] = gsel;
          if (part == 0) p.s.c_out[(int64_t)bs * p.s.ldco + (int64_t)t * p.inc_cout + u] = cn;
          else if (part == 1) p.s.h_out[(int64_t)bs * p.s.ldh + (int64_t)t * p.inc_h + u] = hb;
          else if (part == 2 && !last) p.s.h_out2[(int64_t)bs * p.s.ldh2 + (int64_t)t * p.inc_h2 + u] = hb;
        }
      }
      LAS_STAMP(t, 3);
      lds_barrier();
      LAS_STAMP(t, 4);
      if (ATT == LAS_ATT_BAHDANAU) {
        // processed query pq = h Wq: every part forms ITS quarter of the columns (a quarter of Wq from L2 instead of all of
        // it -- at 512 units the full product in each of the four workgroups was 22 of the step's 49 us) and hands it to
        // the other three as granules
        const int qc = Hd / 4, c0 = part * qc;
        pu64* const xpq = xpqb + ((size_t)(xtag & 1) * B + bs) * Hd;
        const int AG = qc / 8, UG = 256 / AG;               // column groups of 8, row phases
        {
          const int ag = tid % AG, ug = tid / AG;
          float acc[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = 0.f;
          const unsigned short* wp = p.s.wq + c0 + ag * 8;
          int r = ug;
          for (; r + 15 * UG < Hd; r += 16 * UG) {           // (16 rows in flight: the loop is L2 round trips)
            uint4 w[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = ld16(wp + (int64_t)(r + i * UG) * Hd);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const unsigned short* e = reinterpret_cast<const unsigned short*>(&w[i]);
              const float xv = hq[r + i * UG];
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[j] += xv * las_bf2f(e[j]);
            }
          }
          for (; r + 7 * UG < Hd; r += 8 * UG) {
            uint4 w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = ld16(wp + (int64_t)(r + i * UG) * Hd);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const unsigned short* e = reinterpret_cast<const unsigned short*>(&w[i]);
              const float xv = hq[r + i * UG];
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[j] += xv * las_bf2f(e[j]);
            }
          }
          for (; r < Hd; r += UG) {
            const uint4 w = ld16(wp + (int64_t)r * Hd);
            const unsigned short* e = reinterpret_cast<const unsigned short*>(&w);
            const float xv = hq[r];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += xv * las_bf2f(e[j]);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) cred[(ug * AG + ag) * 8 + j] = acc[j];
        }
        lds_barrier();
        for (int c = tid; c < qc; c += 256) {
          float v = 0.f;
          for (int g = 0; g < UG; ++g) v += cred[(g * AG + (c >> 3)) * 8 + (c & 7)];
          pq[c0 + c] = v;
          pgranule_store(xpq + c0 + c, xtag, v, local);
          if (p.s.pq_out) p.s.pq_out[(int64_t)bs * p.s.ldpq + (int64_t)t * p.inc_pq + c0 + c] = v;
        }
        for (int a0 = 0; a0 < Hd; a0 += 256) {             // the other parts' quarters
          const int a = a0 + tid;
          const bool need = a < Hd && (a < c0 || a >= c0 + qc);
          unsigned spins = 0;
          for (;;) {
            const pu64 gq = need ? pgranule_load(xpq + a) : ((pu64)xtag << 32);
            if (__all((unsigned)(gq >> 32) == xtag)) {
              if (need) pq[a] = __uint_as_float((unsigned)gq);
              break;
            }
            if (++spins > P_SPIN_LIMIT || *fail) { *fail = 1; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        lds_barrier();
      }
      // ---- raw scores of my frames ----
      if constexpr (ATT == LAS_ATT_BAHDANAU && NTL == 4) {
        // 512 units: SIXTEEN lanes per frame (4 frames per wave, 16 per pass).  With four, the 25 own frames of a T' = 100
        // utterance kept 100 lanes busy with 128 tanh each -- 4.7 of the 7 us of this phase; now 64 tanh on every lane.
        constexpr int LPF = 16, NKL = NK * 4 / LPF;
        const int sub16 = lane & (LPF - 1), fr16 = lane / LPF;
        for (int fl0 = 0; fl0 < fq; fl0 += 4 * (64 / LPF)) {
          const int fl = fl0 + wave * (64 / LPF) + fr16, tf = f0 + fl;
          float acc = 0.f, acc1 = 0.f;
          if (tf < flen) {
            const lds_cu16 krow = (lds_cu16)lk + (size_t)fl * KST + sub16 * 8;
            uint4 kv[NKL];
#pragma unroll
            for (int j = 0; j < NKL; ++j) kv[j] = ld16(krow + j * 8 * LPF);
#pragma unroll
            for (int j = 0; j < NKL; ++j) {
              const uint4 kk = kv[j];
              const int k = sub16 * 8 + j * 8 * LPF;
              const float4 q0 = *reinterpret_cast<const float4*>(pq + k), q1 = *reinterpret_cast<const float4*>(pq + k + 4);
              const float4 v0 = *reinterpret_cast<const float4*>(vq + k), v1 = *reinterpret_cast<const float4*>(vq + k + 4);
              const float a0 = v0.x * las_tanh(__uint_as_float(kk.x << 16) + q0.x) + v0.y * las_tanh(__uint_as_float(kk.x & 0xffff0000u) + q0.y);
              const float a1 = v0.z * las_tanh(__uint_as_float(kk.y << 16) + q0.z) + v0.w * las_tanh(__uint_as_float(kk.y & 0xffff0000u) + q0.w);
              const float a2 = v1.x * las_tanh(__uint_as_float(kk.z << 16) + q1.x) + v1.y * las_tanh(__uint_as_float(kk.z & 0xffff0000u) + q1.y);
              const float a3 = v1.z * las_tanh(__uint_as_float(kk.w << 16) + q1.z) + v1.w * las_tanh(__uint_as_float(kk.w & 0xffff0000u) + q1.w);
              if (j & 1) acc1 += (a0 + a1) + (a2 + a3);
              else acc += (a0 + a1) + (a2 + a3);
            }
          }
          acc += acc1;
#pragma unroll
          for (int o = 1; o < LPF; o <<= 1) acc += __shfl_xor(acc, o, 64);
          if (sub16 == 0 && fl < fq && tf < f1) {
            const float sv = (tf < len) ? acc : -INFINITY;
            pgranule_store(xsc + tf, xtag, sv, local);     // the other three parts are waiting for it
            sc[tf] = sv;
          }
        }
      } else
      {
        uint4 qreg[ATT == LAS_ATT_LUONG ? NK : 1];
        if (ATT == LAS_ATT_LUONG) {
#pragma unroll
          for (int j = 0; j < NK; ++j) qreg[j] = ld16((lds_cu16)hqb + sub * 8 + j * 32);
        }
        for (int fl0 = 0; fl0 < fq; fl0 += 64) {
          const int fl = fl0 + wave * 16 + fr, tf = f0 + fl;
          float acc = 0.f, acc1 = 0.f;
          if (tf < flen) {
            const lds_cu16 krow = (lds_cu16)lk + (size_t)fl * KST + sub * 8;
            uint4 kv[NK];
#pragma unroll
            for (int j = 0; j < NK; ++j) kv[j] = ld16(krow + j * 32);
#pragma unroll
            for (int j = 0; j < NK; ++j)
              {
                if (ATT == LAS_ATT_LUONG) {
                  acc = dot2_bf16(kv[j].x, qreg[j].x, acc);
                  acc = dot2_bf16(kv[j].y, qreg[j].y, acc);
                  acc = dot2_bf16(kv[j].z, qreg[j].z, acc);
                  acc = dot2_bf16(kv[j].w, qreg[j].w, acc);
                } else {
                  // (pq and attention_v as 16-byte LDS reads, the pieces on two running sums: as scalar reads each pair was
                  //  waited for before its two tanh -- see score_additive_rows)
                  const uint4 kk = kv[j];
                  const int k = sub * 8 + j * 32;
                  const float4 q0 = *reinterpret_cast<const float4*>(pq + k), q1 = *reinterpret_cast<const float4*>(pq + k + 4);
                  const float4 v0 = *reinterpret_cast<const float4*>(vq + k), v1 = *reinterpret_cast<const float4*>(vq + k + 4);
                  const float a0 = v0.x * las_tanh(__uint_as_float(kk.x << 16) + q0.x) + v0.y * las_tanh(__uint_as_float(kk.x & 0xffff0000u) + q0.y);
                  const float a1 = v0.z * las_tanh(__uint_as_float(kk.y << 16) + q0.z) + v0.w * las_tanh(__uint_as_float(kk.y & 0xffff0000u) + q0.w);
                  const float a2 = v1.x * las_tanh(__uint_as_float(kk.z << 16) + q1.x) + v1.y * las_tanh(__uint_as_float(kk.z & 0xffff0000u) + q1.y);
                  const float a3 = v1.z * las_tanh(__uint_as_float(kk.w << 16) + q1.z) + v1.w * las_tanh(__uint_as_float(kk.w & 0xffff0000u) + q1.w);
                  if (j & 1) acc1 += (a0 + a1) + (a2 + a3);
                  else acc += (a0 + a1) + (a2 + a3);
                }
              }
          }
          acc = las_quad_sum(acc + acc1);
          if (sub == 0 && fl < fq && tf < f1) {
            const float sv = (tf < len) ? acc : -INFINITY;
            pgranule_store(xsc + tf, xtag, sv, local);     // the other three parts are waiting for it
            sc[tf] = sv;
          }
        }
      }
      LAS_STAMP(t, 5);
      // ---- the other parts' scores: the data is its own flag ----
      for (int t0 = 0; t0 < Tm; t0 += 256) {
        const int tf = t0 + tid;
        const bool need = tf < Tm && (tf < f0 || tf >= f1);
        unsigned spins = 0;
        for (;;) {                                          // wave-uniform, bounded
          const pu64 gq = need ? pgranule_load(xsc + tf) : ((pu64)xtag << 32);
          if (__all((unsigned)(gq >> 32) == xtag)) {
            if (need) sc[tf] = __uint_as_float((unsigned)gq);
            break;
          }
          if (++spins > P_SPIN_LIMIT || *fail) { *fail = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      lds_barrier();
      LAS_STAMP(t, 6);
      // ---- masked softmax over the frames: every wave reduces its own elements (max, then the sum relative to that
      //      max); the four (max, sum) pairs meet in LDS behind ONE barrier and every thread rescales ----
      {
        float mw = -INFINITY;
        for (int tf = tid; tf < Tm; tf += 256) mw = fmaxf(mw, sc[tf]);
        mw = las_wave_max_dpp(mw);
        float sw = 0.f;
        for (int tf = tid; tf < Tm; tf += 256) {
          const float e = (tf < len) ? __expf(sc[tf] - mw) : 0.f;
          sc[tf] = e;
          sw += e;
        }
        sw = las_wave_sum_dpp(sw);
        if (lane == 0) { red[wave] = mw; red[4 + wave] = sw; }
        lds_barrier();
        const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += red[4 + w] > 0.f ? red[4 + w] * __expf(red[w] - mx) : 0.f;
        const float scale = (len > 0 && sw > 0.f) ? __expf(mw - mx) / sum : 0.f;
        float* const arow_out = p.s.align_out + (int64_t)bs * p.s.lda + (int64_t)t * p.inc_align;
        unsigned short* const abf = p.s.align_bf16 ? p.s.align_bf16 + (int64_t)bs * p.s.lda + (int64_t)t * p.inc_align : nullptr;
        for (int tf = tid; tf < Tm; tf += 256) {
          const float pr = sc[tf] * scale;
          const unsigned short hi = las_f2bf(pr);
          psb[tf] = hi;
          psb[KP + tf] = las_f2bf(pr - las_bf2f(hi));
          if (tf >= f0 && tf < f1) {                       // every part saves its own frames
            arow_out[tf] = pr;
            if (abf) abf[tf] = las_f2bf(pr);
          }
        }
        lds_barrier();
      }
      LAS_STAMP(t, 7);
      // ---- context columns [c_begin, c_begin + ncols) on the matrix cores: ctx[c] = sum_t' p[t'] values[t'][c] as 16x16x32
      //      products, A = the alignments (rows 0, 4, 8, 12 their high halves, rows 1, 5, 9, 13 the low halves, the other
      //      rows zeros), B = 16 columns x 32 frames of the transposed values.  Every quarter of the wave ends up with a
      //      tile's sums, so lane l of wave w keeps column 64 w + l of each pass of 256: no partial sums through LDS.
      //      Chunks past the utterance's length are not visited ----
      {
        const int kcn = (len + 31) / 32;
        const lds_cu16 zslot = (lds_cu16)psb + 2 * KP;
        const lds_cu16 abase = (l15 & 3) < 2 ? (lds_cu16)psb + (l15 & 3) * KP + lq * 8 : zslot;
        const int astep = (l15 & 3) < 2 ? 32 : 0;
        for (int c0 = wave * 64; c0 < CP; c0 += 256) {
          const lds_cu16 brow = (lds_cu16)lv + (size_t)(c0 + l15) * TS;
          f32x4 acc[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
          uint4 av[2], bv[2][4];
          auto request = [&](int buf, int kc) {            // (branch-free: a chunk past the end multiplies zeros)
            const bool in = kc < kcn;
            const int kk = kc * 32 + lq * 8;
            av[buf] = ld16(in ? abase + kc * astep : zslot);
            const lds_cu16 bp = brow + ((in && kk < TS) ? kk : 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[buf][g] = ld16(bp + (size_t)g * 16 * TS);
          };
          auto products = [&](int buf) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
              acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[buf]), __builtin_bit_cast(bf16x8, bv[buf][g]), acc[g], 0, 0, 0);
          };
          request(0, 0);
#pragma unroll 1
          for (int kc = 0; kc < kcn; kc += 2) {
            request(1, kc + 1);
            __builtin_amdgcn_sched_barrier(0);
            products(0);
            __builtin_amdgcn_sched_barrier(0);
            request(0, kc + 2);
            __builtin_amdgcn_sched_barrier(0);
            products(1);
            __builtin_amdgcn_sched_barrier(0);
          }
          const float r0 = acc[0][0] + acc[0][1], r1 = acc[1][0] + acc[1][1], r2 = acc[2][0] + acc[2][1], r3 = acc[3][0] + acc[3][1];
          const float cv = lq == 0 ? r0 : (lq == 1 ? r1 : (lq == 2 ? r2 : r3));
          const int j = c0 + lane;                          // = c0 - 64 wave + tid
          if (c0 == wave * 64) LAS_STAMP(t, 8);
          if (j < ncols) {
            const unsigned short o = las_f2bf(cv);
            p.s.ctx_out[(int64_t)bs * p.s.ldc + (int64_t)t * p.inc_ctx + c_begin + j] = o;
            if constexpr (SAMPLING) cred[j] = las_bf2f(o);   // (the sampling phase's operand: the context as the projection sees it)
            if (!last) {
              unsigned short o2 = o;
              if (keep < 1.0f) {   // the copy that feeds step t+1's cell goes through that step's input dropout
                const unsigned long long idx = ((unsigned long long)(t + 1) * B + bs) * p.s.feed_width + (p.s.feed_width - M) + c_begin + j;
                o2 = las_uniform(p.s.drop_seed, p.s.drop_stream, idx) < keep ? las_f2bf(las_bf2f(o) / keep) : (unsigned short)0;
              }
              p.s.ctx_out2[(int64_t)bs * p.s.ldc2 + (int64_t)t * p.inc_ctx2 + c_begin + j] = o2;
            }
          }
        }
      }
      LAS_STAMP(t, 9);
    }
    tok_cur = tok_next;
    bool resampled = false;
    if constexpr (SAMPLING) {
      // ---- scheduled sampling (utils/training_helper.py:48-87), as in dec_persist_fwd_kernel: logits_t = context_t W_proj + b
      //      from the four parts' partial products, then the token fed at step t+1 = Categorical(logits_t) with probability
      //      p, else the teacher's.  The draws are counter-based: every member knows which utterances of its group are
      //      selected at this step; a group without a selection skips the phase and its barrier ----
      const int V = p.V, Vp = p.Vp;
      // (one draw per lane -- lane & 7 = the utterance of the group -- and a ballot: eight draws per thread were 1.3 us of
      //  every step, selection or not)
      const int bj = group * 8 + (lane & 7);
      const bool selj = t + 1 < U && bj < B && las_uniform(p.seed, 0x5e1ec7u, (unsigned long long)t * B + bj) < p.sampling_prob;
      const unsigned selmask = (unsigned)(__ballot(selj) & 0xffull);
      const bool any_sel = selmask != 0;
      const bool my_sel = bs < B && ((selmask >> (bs - group * 8)) & 1u) != 0;
      if (any_sel) {
        float* plog_t = p.plog + ((int64_t)t * B + bsc) * 4 * Vp;
        if (my_sel) {
          const float* cx = cred;                         // [ncols] my context columns (bf16 values), left there by the context phase
          __syncthreads();
          if (V <= 64 && (ncols & 31) == 0) {
            // 64 rows x 4 column quarters: every thread's pieces of W_proj are requested at once (one L2 round trip, not
            // ncols / 8 of them in 64 threads), the quarters meet in LDS
            const int v = tid & 63, q = tid >> 6, qn = ncols >> 2;
            float acc = 0.f;
            if (v < V) {
              const unsigned short* wrow = p.wprojT + (int64_t)v * p.ldw + c_begin + q * qn;
              for (int c0 = 0; c0 < qn; c0 += 64) {
                uint4 w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = c0 + i * 8 < qn ? *reinterpret_cast<const uint4*>(wrow + c0 + i * 8) : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                  if (c0 + i * 8 < qn) acc += dot8(w[i], cx + q * qn + c0 + i * 8);
              }
            }
            cred[1024 + tid] = acc;
            __syncthreads();
            if (tid < Vp) plog_t[part * Vp + tid] = tid < V ? (cred[1024 + tid] + cred[1088 + tid]) + (cred[1152 + tid] + cred[1216 + tid]) : 0.f;
          } else {
            for (int v = tid; v < Vp; v += 256) {
              float acc = 0.f;
              if (v < V) {
                const unsigned short* wrow = p.wprojT + (int64_t)v * p.ldw + c_begin;
                for (int c = 0; c < ncols; c += 8) {
                  const uint4 w = *reinterpret_cast<const uint4*>(wrow + c);
                  acc += dot8(w, cx + c);
                }
              }
              plog_t[part * Vp + v] = acc;
            }
          }
        }
        if (!persist_barrier(flags, member, ++epoch, local, fail)) break;
        if (my_sel && part == 0) {
          float* lg = cred;                               // [Vp]
          for (int v = tid; v < Vp; v += 256)
            lg[v] = v < V ? plog_t[v] + plog_t[Vp + v] + plog_t[2 * Vp + v] + plog_t[3 * Vp + v] + p.bproj[v] : p.bproj[v];
          __syncthreads();
          if (tid < 64) {
            const unsigned long long sidx = (unsigned long long)t * B + bs;
            float best = -INFINITY;
            int arg = 0;
            for (int v = lane; v < V; v += 64) {
              const float u = fmaxf(las_uniform(p.seed, 0x9a3b1eu, sidx * V + v), 1e-12f);
              const float gmb = lg[v] - __logf(-__logf(u));
              if (gmb > best) { best = gmb; arg = v; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
              const float ob = __shfl_xor(best, o, 64);
              const int oa = __shfl_xor(arg, o, 64);
              if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
            }
            if (lane == 0) const_cast<int32_t*>(p.s.tok_ids)[(int64_t)bs * p.s.tok_stride + t + 1] = arg;   // else: the teacher's, already there
          }
        }
        resampled = my_sel;
      }
    }
    if (!persist_barrier(flags, member, ++epoch, local, fail)) break;
    if constexpr (SAMPLING) {
      // the token part 0 has just drawn for step t+1 (behind the barrier: every part of the utterance reads it)
      if (resampled) tok_cur = __hip_atomic_load(p.s.tok_ids + (int64_t)bs * p.s.tok_stride + t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    LAS_STAMP(t, 10);
  }
  if (*fail && tid == 0) atomicOr(status, 8u);
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// Sum over the workgroups of a launch in a FIXED order (d(attention_v), d(attention_score_bias) of the one-launch backward
// decoders: fp32 atomics would add the workgroups' shares in the order they happen to finish, and two runs would differ in
// the last bits).  Every workgroup hands in n values `mine` (LDS); ws = {counter word, 15 pad words, nblk rows of n floats}.  The
// rows are written with write-through stores and acknowledged before the counter moves (no release fence: see l2_norm_kernel);
// the LAST workgroup to arrive adds the rows up in workgroup order with L2-bypassing loads, adds the sums into dst_a[0..na) and
// dst_b[0..n-na) and clears the counter for the next launch.  All 256 threads of every workgroup call it.
__device__ void ordered_accumulate(unsigned* ws, const int blk, const int nblk, const int n, const float* mine, float* dst_a, const int na,
                                   float* dst_b, int* lds_word) {
  const int tid = threadIdx.x;
  float* rows = reinterpret_cast<float*>(ws + 16);
  for (int u = tid; u < n; u += 256) __hip_atomic_store(rows + (int64_t)blk * n + u, mine[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) *lds_word = (__hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nblk - 1)) ? 1 : 0;
  __syncthreads();
  if (!*lds_word) return;
  for (int u = tid; u < n; u += 256) {
    float t = 0.f;
#pragma unroll 8
    for (int b = 0; b < nblk; ++b) t += __hip_atomic_load(rows + (int64_t)b * n + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (u < na) { if (dst_a) dst_a[u] += t; }
    else if (dst_b) dst_b[u - na] += t;
  }
  if (tid == 0) __hip_atomic_store(ws, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS floats of one backward step (dec_step_bwd_body): d(context), dalign -> dscore, per-phase partials, scratch, the
// monotonic normaliser's five arrays, the query-layer scratch
__host__ __device__ inline size_t dec_step_bwd_floats(int M, int Tm, int Hd, int norm) {
  return (size_t)M + Tm + 2048 + 8 + Hd + (norm != LAS_NORM_SOFTMAX ? 5 * (size_t)Tm : 0) + 2048;
}

// One backward step of utterance b by the 256 threads of a workgroup (dec_step_bwd_kernel: one launch per step; dec_seq_bwd_kernel:
// all steps of an utterance in one launch, the struct's pointers advanced by the caller; a pointer may then address LDS).
// vw / datt / A (dec_seq_bwd_kernel with an attention layer): d(alignments) without touching the values.  d(context) is
// d(attention) W_c^T there (W_c: the context rows of the attention layer's kernel), so d(alignments)[t'] = values[t'] . d(context)
// = sum_a d(attention)[a] (values W_c)[t'][a]: VW = values W_c [T', A] does not depend on the decoder step -- one product per
// train step, resident in LDS (row stride A + 1) -- and a step's pass over the utterance's values (400 KB from beyond L2: 21 of
// the sequential backward's 75 us per step at cfg5) becomes T' x A multiply-adds.
// Exchange rows of one utterance (PARTS = 4), in granules: d(scores) [Tm], partial d(processed query) [3][Hd], d(attention_t) as
// bf16 pairs [64], d[query | context] [Hd + M], dz_t as bf16 pairs [2 Hd], d[feed | h] [W0], XCC ids [4]
struct SeqXLayout { int xdq, xda, xqc, xdz, xdf, xcc, total; };
__host__ __device__ inline SeqXLayout seq_xlayout(int Tm, int Hd, int M, int W0) {
  SeqXLayout L;
  L.xdq = Tm;
  L.xda = L.xdq + 3 * Hd;
  L.xqc = L.xda + 64;
  L.xdz = L.xqc + Hd + M;
  L.xdf = L.xdz + 2 * Hd;
  L.xcc = L.xdf + ((W0 + 15) & ~15);                   // [4] {1, XCC id + 1} of the four parts (are they on one XCD?)
  L.total = (L.xcc + 4 + 15) & ~15;
  return L;
}

// out(n, value) for the columns n of the 16-column tiles vwave, vwave + nvw, ... of y = a W^T, W as a LAS_IMAGE_PACK_MFMA_B image
// with KCa 32-deep chunks, a (bf16, row 0 of the A tile) in LDS at a_lds with 8 zeros at zeros: TB tiles per pass, one pass and
// chunk = TB loads in flight, the next (pass, chunk)'s requested before this one's products.  (d[query | context] = d(attention) W_al^T)
template <typename OUT>
__device__ __forceinline__ void seq_matvec_few_chunks(const unsigned short* packed, const int KCa, const int NT, const unsigned short* a_lds,
                                                      const unsigned short* zeros, const int lane, const int vwave, const int nvw, OUT out) {
  constexpr int TB = 10;
  const int l15 = lane & 15, lq = lane >> 4;
  const unsigned short* azp = l15 == 0 ? a_lds + 8 * lq : zeros;
  const int azs = l15 == 0 ? 32 : 0;
  const int mine = vwave < NT ? (NT - vwave + nvw - 1) / nvw : 0;       // tiles of this (virtual) wave
  const int NB = (mine + TB - 1) / TB, Q = NB * KCa;
  if (Q == 0) return;
  uint4 cur[TB], nxt[TB];
  f32x4 acc[TB];
  auto frag = [&](int q, int i) {
    const int bi = q / KCa, kc = q - bi * KCa;
    const int nt = min(vwave + nvw * (bi * TB + i), NT - 1);
    return packed + (((int64_t)nt * KCa + kc) * 64 + lane) * 8;
  };
#pragma unroll
  for (int i = 0; i < TB; ++i) cur[i] = ld16(frag(0, i));
#pragma unroll 1
  for (int q = 0; q < Q; ++q) {
    const int bi = q / KCa, kc = q - bi * KCa;
    if (q + 1 < Q) {
#pragma unroll
      for (int i = 0; i < TB; ++i) nxt[i] = ld16(frag(q + 1, i));
    }
    const uint4 av = *reinterpret_cast<const uint4*>(azp + kc * azs);
    if (kc == 0) {
#pragma unroll
      for (int i = 0; i < TB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < TB; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, cur[i]), acc[i], 0, 0, 0);
    if (kc == KCa - 1) {
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        const int nt = vwave + nvw * (bi * TB + i);
        if (lq == 0 && nt < NT) out(nt * 16 + l15, acc[i][0]);
      }
    }
#pragma unroll
    for (int i = 0; i < TB; ++i) cur[i] = nxt[i];
  }
}

// The same for a deep product (KC chunks, compile-time): a tile's 16 fragments of half its chunks in flight at once, the A
// fragments read from LDS beside them.  (d[feed | h]_t = dz_t K^T)
template <int KC, typename OUT>
__device__ __forceinline__ void seq_matvec_deep(const unsigned short* packed, const int NT, const unsigned short* a_lds, const unsigned short* zeros,
                                                const int lane, const int vwave, const int nvw, OUT out) {
  const int l15 = lane & 15, lq = lane >> 4;
  const unsigned short* azp = l15 == 0 ? a_lds + 8 * lq : zeros;
  const int azs = l15 == 0 ? 32 : 0;
  for (int nt = vwave; nt < NT; nt += nvw) {
    const unsigned short* kfr = packed + ((int64_t)nt * KC * 64 + lane) * 8;     // fragment (nt, kc): + kc * 512
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < KC; k0 += 16) {
      uint4 bv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) bv[i] = ld16(kfr + (k0 + i) * 512);
      uint4 av[16];                              // (the LDS reads overlap the loads' round trip)
#pragma unroll
      for (int i = 0; i < 16; ++i) av[i] = *reinterpret_cast<const uint4*>(azp + (k0 + i) * azs);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[i]), __builtin_bit_cast(bf16x8, bv[i]), acc, 0, 0, 0);
    }
    if (lq == 0) out(nt * 16 + l15, acc[0]);
  }
}

// part that owns column n of such a product split over 4 x 4 virtual waves (tile n / 16 belongs to virtual wave (n / 16) % 16)
__device__ __forceinline__ int seq_owner(int n) { return ((n >> 4) & 15) >> 2; }

// ... and for a square product with ALL of a wave's fragments in flight (KC^2 / 2 of them: 32 at 256 units): y = a W^T with
// 2 KC tiles, wave w takes the tiles w, w + 4, ...  (dh = d(processed query) Wq^T of the Bahdanau / Custom query layer)
template <int KC, typename OUT>
__device__ __forceinline__ void seq_matvec_square(const unsigned short* packed, const unsigned short* a_lds, const unsigned short* zeros,
                                                  const int lane, const int wave, OUT out) {
  constexpr int TPW = KC / 2;
  const int l15 = lane & 15, lq = lane >> 4;
  const unsigned short* azp = l15 == 0 ? a_lds + 8 * lq : zeros;
  const int azs = l15 == 0 ? 32 : 0;
  uint4 bv[TPW][KC], av[KC];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) bv[i][kc] = ld16(packed + (((int64_t)(wave + 4 * i) * KC + kc) * 64 + lane) * 8);
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) av[kc] = *reinterpret_cast<const uint4*>(azp + kc * azs);
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[kc]), __builtin_bit_cast(bf16x8, bv[i][kc]), acc, 0, 0, 0);
    if (lq == 0) out((wave + 4 * i) * 16 + l15, acc[0]);
  }
}

// dec_seq_bwd_kernel with FOUR workgroups per utterance (PARTS = 4): part 0 walks the chain as before, parts 1..3 take three
// quarters of the memory frames of the Bahdanau query path off it -- 200 tanh per thread and step on one CU otherwise.
// Per step part 0 publishes d(scores) of the frames it does not own (xds, {tag, fp32} granules), every part runs the query path
// over its own frames (d(keys) of those frames in ITS registers, d(attention_v) in its own running sum) and parts 1..3 send
// their partial d(processed query) back (xdq); part 0 adds them in part order.  tag = U - t (1, 2, ...: the workspace is zeroed
// at every launch); every wait is bounded (the sticky status word of the workspace reports a timeout).
struct SeqXchg {
  pu64* xds;            // [Tm] d(scores) of this utterance and step
  pu64* xdq;            // [3][Hd] partial d(processed query) of parts 1..3
  unsigned tag;
  int* fail;            // LDS
  int f0, f1;           // this workgroup's frames
  bool local;           // the four parts share an XCD (its L2 is the point of coherence: plain stores)
};
constexpr unsigned SEQ_SPIN_LIMIT = 1u << 22;

// Wave-uniform bounded wait for granule p[i] (threads with have = false ride along); returns its value (0 after a timeout).
__device__ __forceinline__ float seq_wait(const pu64* p, const bool have, const unsigned tag, int* fail) {
  unsigned spins = 0;
  for (;;) {
    const pu64 g = have ? pgranule_load(p) : ((pu64)tag << 32);
    if (__all((unsigned)(g >> 32) == tag)) return __uint_as_float((unsigned)g);
    if (++spins > SEQ_SPIN_LIMIT || *fail) { *fail = 1; return 0.f; }
    __builtin_amdgcn_s_sleep(1);
  }
}

// Bahdanau query path over the frames [f0, flen) of an utterance with d(keys) in registers (see dec_step_bwd_body, NPK > 0):
// thread (phase, u) takes frames f0 + phase + i P, columns u .. u + 7; the key rows of the NEXT FB passes are requested before
// this group's tanh work (two register buffers).  a: d(processed query) partial, dv: d(attention_v) partial.
template <int NPK>
__device__ __forceinline__ void query_path_regk(const unsigned short* keys, const int Hd, const int f0, const int flen, const float* ds,
                                                const float* qq, const float* vv, const int phase, const int P, const int u,
                                                float (&a)[8], float (&dv)[8], float (*dkr)[8]) {
  constexpr int FB = 4, NG = (NPK + FB - 1) / FB;
  uint4 kv[2][FB];
#pragma unroll
  for (int i = 0; i < FB; ++i) {
    const int t = f0 + phase + i * P;
    if (i < NPK && t < flen) kv[0][i] = *reinterpret_cast<const uint4*>(keys + (int64_t)t * Hd + u);
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
#pragma unroll
      for (int i = 0; i < FB; ++i) {
        const int t = f0 + phase + ((g + 1) * FB + i) * P;
        if ((g + 1) * FB + i < NPK && t < flen) kv[(g + 1) & 1][i] = *reinterpret_cast<const uint4*>(keys + (int64_t)t * Hd + u);
      }
    }
#pragma unroll
    for (int i = 0; i < FB; ++i) {
      const int t = f0 + phase + (g * FB + i) * P;
      if (g * FB + i < NPK && t < flen) {
        const unsigned short* e = reinterpret_cast<const unsigned short*>(&kv[g & 1][i]);
        const float d = ds[t];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float th = las_tanh(las_bf2f(e[j]) + qq[j]);
          dv[j] += d * th;
          const float p = d * vv[j] * (1.f - th * th);
          a[j] += p;
          dkr[g * FB + i < NPK ? g * FB + i : 0][j] += p;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);         // keep later groups' loads from being hoisted over this one (registers)
  }
}

// NPK > 0 (dec_seq_bwd_kernel, Bahdanau scores): the utterance's fp32 d(keys) lives in REGISTERS across the steps -- dkr[i][j]
// is frame phase + i P, column u + j of the thread's fixed (phase, u) -- instead of being read and written in memory at
// every step (NPK frame passes x 8 columns; the caller adds them into dkeys_acc once, after the last step).
constexpr int SEQ_NPK = 25;
constexpr int SEQ_NPK4 = 7;        // ... of a quarter of the frames (four workgroups per utterance)
template <int NPK = 0>
__device__ __forceinline__ void dec_step_bwd_body(const las_dec_step_bwd& s, const int b, float* sm, const float* vw = nullptr,
                                                  const float* datt = nullptr, const int A = 0, float (*dkr)[8] = nullptr,
                                                  const int tid_in = -1, float* acc_run = nullptr, const SeqXchg* xc = nullptr,
                                                  const unsigned short* wq_pk = nullptr) {
  float* dctx = sm;               // [M]
  float* ds = dctx + s.M;         // [Tm] dalign -> dscore
  float* dhs = ds + s.Tm;         // [256/L][Hd] = 2048 floats: per-phase partial d h (score path) / dpq
  float* red = dhs + 2048;        // [8] + [Hd] scratch

  // (tid_in: the caller's opaque copy of threadIdx.x -- see dec_seq_bwd_kernel)
  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = tid_in >= 0 ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
  const int Hd = s.Hd, M = s.M, Tm = s.Tm;
  if (s.mode == LAS_DEC_CELL_ONLY) {
    for (int u = tid; u < Hd; u += 256) dhs[u] = 0.f;
    __syncthreads();
  } else {
  const int len = min(s.mem_len[b], Tm);
  // the register-resident monotonic chain's inputs (saved p_choose, the previous alignments, the carry of step t+1's
  // normaliser -- written by this very thread) are requested now and used two phases later
  float pre_p = 0.f, pre_pr = 0.f, pre_carry = 0.f;
  if (s.norm == LAS_NORM_MONOTONIC_PARALLEL && Tm <= 256 && tid < Tm) {
    if (tid < len) pre_p = s.p[(int64_t)b * s.ldp + tid];
    pre_pr = s.prev_align ? s.prev_align[(int64_t)b * s.ldpa + tid] : (tid == 0 ? 1.f : 0.f);
    pre_carry = s.dalign_carry[(int64_t)b * s.ldcarry + tid];
  }

  // total gradient w.r.t. the context of this step; keep a bf16 copy for the d(memory) batched GEMM
  for (int m = tid; m < M; m += 256) {
    float v = s.dctx_a ? s.dctx_a[(int64_t)b * s.ldda + m] : 0.f;
    if (s.dctx_b) {
      float fb = s.dctx_b[(int64_t)b * s.lddb + m];
      if (s.drop_keep < 1.0f) {     // gradient through step t+1's input dropout of the attention feed
        const unsigned long long idx = ((unsigned long long)(s.step + 1) * s.B + b) * s.feed_width + (s.feed_width - M) + m;
        fb = las_uniform(s.drop_seed, s.drop_stream, idx) < s.drop_keep ? fb / s.drop_keep : 0.f;
      }
      v += fb;
    }
    dctx[m] = v;
    if (s.dctx_save) s.dctx_save[(int64_t)b * s.ldds + m] = las_f2bf(v);
  }
  __syncthreads();
  LAS_STAMPB(s.step, 2);

  // dalign[t'] = values[b,t',:] . dctx : 16 lanes per frame (4 frames per wave instruction); the loads of DB frames per
  // lane are all in flight together (cold L2 at every launch: each dependent round trip goes to Infinity Cache)
  const unsigned short* vals = s.values + (int64_t)b * Tm * M;
  if (NPK > 0 || vw) {                                 // (the register-d(keys) kernel is only launched with VW)
    for (int t = tid; t < Tm; t += 256) {
      float acc = 0.f;
      if (t < len) {
        // (A is a multiple of 8; eight LDS reads of the row in flight and two running sums: the rolled loop waited for every
        //  pair of reads -- 3.6 us per step for 80 multiply-adds per thread)
        const float* row = vw + (int64_t)t * (A + 1);
        float acc1 = 0.f;
        for (int a = 0; a < A; a += 8) {
          float r[8], d[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { r[j] = row[a + j]; d[j] = datt[a + j]; }
          acc += r[0] * d[0] + r[2] * d[2] + r[4] * d[4] + r[6] * d[6];
          acc1 += r[1] * d[1] + r[3] * d[3] + r[5] * d[5] + r[7] * d[7];
        }
        acc += acc1;
      }
      ds[t] = acc;
    }
  } else {
    constexpr int DB = 8, NVMAX = 4;               // frames in flight per lane; 16-byte pieces per lane and frame (M <= 512)
    const int sub = lane & 15, grp = lane >> 4;
    const int nv = M / 128;
    if (nv <= NVMAX) {
      for (int t0 = 0; t0 < Tm; t0 += 16 * DB) {
        uint4 vv[DB][NVMAX];
#pragma unroll
        for (int f = 0; f < DB; ++f) {
          const int t = t0 + f * 16 + wave * 4 + grp;
          const unsigned short* r = vals + (int64_t)min(t, Tm - 1) * M;
#pragma unroll
          for (int j = 0; j < NVMAX; ++j)
            if (j < nv && t < len) vv[f][j] = *reinterpret_cast<const uint4*>(r + sub * 8 + j * 128);
        }
#pragma unroll
        for (int f = 0; f < DB; ++f) {
          const int t = t0 + f * 16 + wave * 4 + grp;
          float acc = 0.f;
          if (t < len) {
#pragma unroll
            for (int j = 0; j < NVMAX; ++j)
              if (j < nv) acc += dot8(vv[f][j], dctx + sub * 8 + j * 128);
          }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
          if (sub == 0 && t < Tm) ds[t] = (t < len) ? acc : 0.f;
        }
      }
    } else if (nv <= 8) {
      // M <= 1024 (the usual pyramid output, 4 x 256): four frames per lane at a time with ALL of their pieces in flight (32
      // loads of 16 bytes).  The rolled loop below kept eight in flight and made seven dependent rounds of them per step:
      // 23 us of the sequential backward's 78 us per step at cfg5 (phase stamps).
      constexpr int FR = 4, NV8 = 8;
      for (int t0 = 0; t0 < Tm; t0 += 16 * FR) {
        uint4 vv[FR][NV8];
#pragma unroll
        for (int f = 0; f < FR; ++f) {
          const int t = t0 + f * 16 + wave * 4 + grp;
          const unsigned short* r = vals + (int64_t)min(t, Tm - 1) * M;
#pragma unroll
          for (int j = 0; j < NV8; ++j)
            if (j < nv && t < len) vv[f][j] = *reinterpret_cast<const uint4*>(r + sub * 8 + j * 128);
        }
#pragma unroll
        for (int f = 0; f < FR; ++f) {
          const int t = t0 + f * 16 + wave * 4 + grp;
          float acc = 0.f;
          if (t < len) {
#pragma unroll
            for (int j = 0; j < NV8; ++j)
              if (j < nv) acc += dot8(vv[f][j], dctx + sub * 8 + j * 128);
          }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
          if (sub == 0 && t < Tm) ds[t] = (t < len) ? acc : 0.f;
        }
      }
    } else {
      for (int t0 = 0; t0 < Tm; t0 += 32) {
        const int ta = t0 + wave * 4 + grp, tb = ta + 16;
        float acc_a = 0.f, acc_b = 0.f;
        const bool oa = ta < len, ob = tb < len;
        const unsigned short* ra = vals + (int64_t)(oa ? ta : 0) * M;
        const unsigned short* rb = vals + (int64_t)(ob ? tb : 0) * M;
#pragma unroll 4
        for (int k = sub * 8; k < M; k += 128) {
          const uint4 va = *reinterpret_cast<const uint4*>(ra + k);
          const uint4 vb = *reinterpret_cast<const uint4*>(rb + k);
          acc_a += dot8(va, dctx + k);
          acc_b += dot8(vb, dctx + k);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          acc_a += __shfl_xor(acc_a, o, 64);
          acc_b += __shfl_xor(acc_b, o, 64);
        }
        if (sub == 0) {
          if (ta < Tm) ds[ta] = oa ? acc_a : 0.f;
          if (tb < Tm) ds[tb] = ob ? acc_b : 0.f;
        }
      }
    }
  }
  __syncthreads();

  LAS_STAMPB(s.step, 3);
  if (s.norm == LAS_NORM_MONOTONIC_PARALLEL) {
    // backward of a = p * c * S,  c = exp(excl-cumsum(log clip(1-p, tiny, 1))),  S = cumsum(prev / clip(c, 1e-10, 1));
    // the gradient also flows into prev = align_{t-1} (dalign_carry) and comes back from step t+1 the same way
    float* wp = red + 8 + Hd;       // [Tm] p
    float* wc = wp + Tm;            // [Tm] c
    float* wS = wc + Tm;            // [Tm] S, later dS -> du
    float* wd = wS + Tm;            // [Tm] dc -> dcs -> dlx
    float* wprev = wd + Tm;         // [Tm] prev
    float* tmp = dhs;               // scan scratch (dhs is free until the query-path phase)
    const float* pv = s.p + (int64_t)b * s.ldp;
    const float* prev = s.prev_align ? s.prev_align + (int64_t)b * s.ldpa : nullptr;
    float* carry = s.dalign_carry + (int64_t)b * s.ldcarry;
    if (Tm <= 256) {
      // frame t in the registers of thread t through the whole chain: four scans, four barriers (see scan256)
      const int t = tid;
      const bool in = t < Tm;
      const float p = pre_p, pr = pre_pr;
      const float da = in ? ds[t] + pre_carry : 0.f;          // d(align_t): this step's + step t+1's normaliser's
      const float c = __expf(scan256<false, false, true>(in ? __logf(fminf(fmaxf(1.f - p, 1.17549435e-38f), 1.f)) : 0.f, tmp, lane, wave));
      const float cc = fminf(fmaxf(c, 1e-10f), 1.f);
      const float S = scan256<false, false, false>(in ? pr / cc : 0.f, tmp + 4, lane, wave);
      const float du = scan256<false, true, false>(in ? da * p * c : 0.f, tmp, lane, wave);      // sum_{i >= t} dS[i]
      float dc = da * p * S;
      if (c >= 1e-10f && c <= 1.f) dc -= du * pr / (cc * cc);
      const float dlx = scan256<false, true, true>(in ? dc * c : 0.f, tmp + 4, lane, wave);         // sum_{j > t} dcs[j]
      float v = 0.f;
      if (in) {
        carry[t] = du / cc;                                   // d(align_{t-1})
        const float x = 1.f - p;
        float dp = da * c * S;
        if (x >= 1.17549435e-38f && x <= 1.f) dp -= dlx / x;
        v = (t < len) ? dp * p * (1.f - p) : 0.f;
        ds[t] = v;
        if (s.ds_out) s.ds_out[(int64_t)b * s.ldso + t] = las_f2bf(v);
        if (xc && (t < xc->f0 || t >= xc->f1)) pgranule_store(xc->xds + t, xc->tag, v, xc->local);     // the other parts' frames
      }
      const float dbias = block_reduce(v, red, false);
      if (tid == 0 && s.dbias_acc) { if (acc_run) acc_run[Hd] += dbias; else atomicAdd(s.dbias_acc, dbias); }   // (acc_run: see dec_seq_bwd_kernel)
      __syncthreads();
    } else {
    for (int t = tid; t < Tm; t += 256) {
      const float p = (t < len) ? pv[t] : 0.f;
      wp[t] = p;
      wc[t] = __logf(fminf(fmaxf(1.f - p, 1.17549435e-38f), 1.f));
      wprev[t] = prev ? prev[t] : (t == 0 ? 1.f : 0.f);
      ds[t] += carry[t];                                    // d(align_t) from step t+1's normaliser
    }
    __syncthreads();
    block_scan<false>(wc, Tm, tmp, true, false);
    for (int t = tid; t < Tm; t += 256) {
      const float c = __expf(wc[t]);
      wc[t] = c;
      wS[t] = wprev[t] / fminf(fmaxf(c, 1e-10f), 1.f);
    }
    block_scan<false>(wS, Tm, tmp, false, false);
    for (int t = tid; t < Tm; t += 256) {
      const float da = ds[t], p = wp[t], c = wc[t], S = wS[t];
      ds[t] = da * c * S;            // dp (direct part)
      wd[t] = da * p * S;            // dc (direct part)
      wS[t] = da * p * c;            // dS
    }
    block_scan<false>(wS, Tm, tmp, false, true);            // du[t] = sum_{i >= t} dS[i]
    for (int t = tid; t < Tm; t += 256) {
      const float c = wc[t], cc = fminf(fmaxf(c, 1e-10f), 1.f), du = wS[t];
      carry[t] = du / cc;                                   // d(align_{t-1})
      float dc = wd[t];
      if (c >= 1e-10f && c <= 1.f) dc -= du * wprev[t] / (cc * cc);
      wd[t] = dc * c;                                       // d(cs)
    }
    block_scan<false>(wd, Tm, tmp, true, true);             // dlx[k] = sum_{j > k} dcs[j]
    float dbias = 0.f;
    for (int t = tid; t < Tm; t += 256) {
      const float p = wp[t], x = 1.f - p;
      float dp = ds[t];
      if (x >= 1.17549435e-38f && x <= 1.f) dp -= wd[t] / x;
      const float v = (t < len) ? dp * p * (1.f - p) : 0.f;
      ds[t] = v;
      dbias += v;
      if (s.ds_out) s.ds_out[(int64_t)b * s.ldso + t] = las_f2bf(v);
      if (xc && (t < xc->f0 || t >= xc->f1)) pgranule_store(xc->xds + t, xc->tag, v, xc->local);
    }
    dbias = block_reduce(dbias, red, false);
    if (tid == 0 && s.dbias_acc) { if (acc_run) acc_run[Hd] += dbias; else atomicAdd(s.dbias_acc, dbias); }
    __syncthreads();
    }
  } else {
  // softmax backward: ds = p * (dalign - sum p*dalign)
  const float* align = s.align + (int64_t)b * s.lda;
  float dot = 0.f;
  for (int t = tid; t < len; t += 256) dot += align[t] * ds[t];
  dot = block_reduce(dot, red, false);
  for (int t = tid; t < Tm; t += 256) {
    const float v = (t < len) ? align[t] * (ds[t] - dot) : 0.f;
    ds[t] = v;
    if (s.ds_out) s.ds_out[(int64_t)b * s.ldso + t] = las_f2bf(v);
    if (xc && (t < xc->f0 || t >= xc->f1)) pgranule_store(xc->xds + t, xc->tag, v, xc->local);
  }
  __syncthreads();
  }

  LAS_STAMPB(s.step, 4);
  // gradient into the query path: L = Hd/8 lanes cover one frame (16-byte loads), 256/L frame phases
  const unsigned short* keys = s.keys + (int64_t)b * Tm * Hd;
  const int L = Hd / 8, P = 256 / L;            // Hd in {64,...,2048} and a power of two
  {
    const int phase = tid / L, u = (tid % L) * 8;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    if (NPK == 0 && !att_additive(s.attention)) {      // (NPK > 0: launched for Bahdanau scores only)
      constexpr int KB = 16;                 // key loads in flight per thread
      for (int tb = phase; tb < len; tb += P * KB) {
        uint4 kk[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const int t = tb + i * P;
          if (t < len) kk[i] = *reinterpret_cast<const uint4*>(keys + (int64_t)t * Hd + u);
        }
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const int t = tb + i * P;
          if (t < len) {
            const unsigned short* e = reinterpret_cast<const unsigned short*>(&kk[i]);
            const float d = ds[t];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += d * las_bf2f(e[j]);
          }
        }
      }
    } else {
      // Bahdanau: score = sum_a v[a] tanh(keys[t',a] + pq[a]); d_pre = ds * v * (1 - tanh^2)
      const float* pqv = s.pq + (int64_t)b * s.ldpq;
      float dv[8], vv[8], qq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { dv[j] = 0.f; vv[j] = s.att_v[u + j]; qq[j] = pqv[u + j]; }
      // FB frames per thread at a time: their key rows and their d(keys) rows (read-modify-write, 2 x 16 bytes) are all
      // requested before the first is used (one frame at a time made every frame a dependent L2 round trip: 25 per thread
      // and step at 512 units)
      constexpr int FB = 4;
      if constexpr (NPK > 0) {
        // (with four workgroups per utterance -- xc -- this one's frames only)
        const int qf0 = xc ? xc->f0 : 0, qflen = xc ? min(len, xc->f1) : len;
        query_path_regk<NPK>(keys, Hd, qf0, qflen, ds, qq, vv, phase, P, u, a, dv, dkr);
      } else
      for (int tb = phase; tb < len; tb += P * FB) {
        uint4 kv[FB];
        float4 d0[FB], d1[FB];
#pragma unroll
        for (int i = 0; i < FB; ++i) {
          const int t = tb + i * P;
          if (t < len) {
            kv[i] = *reinterpret_cast<const uint4*>(keys + (int64_t)t * Hd + u);
            const float4* dk = reinterpret_cast<const float4*>(s.dkeys_acc + ((int64_t)b * Tm + t) * Hd + u);
            d0[i] = dk[0];
            d1[i] = dk[1];
          }
        }
#pragma unroll
        for (int i = 0; i < FB; ++i) {
        const int t = tb + i * P;
        if (t >= len) continue;
        const unsigned short* e = reinterpret_cast<const unsigned short*>(&kv[i]);
        const float d = ds[t];
        float dkv[8] = {d0[i].x, d0[i].y, d0[i].z, d0[i].w, d1[i].x, d1[i].y, d1[i].z, d1[i].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float th = las_tanh(las_bf2f(e[j]) + qq[j]);
          dv[j] += d * th;
          const float p = d * vv[j] * (1.f - th * th);
          a[j] += p;
          dkv[j] += p;
        }
        // (streaming hints on this read-modify-write were tried -- the 8 utterances of an XCD move 3.2 MB of fp32 d(keys) through
        //  its 4 MB L2 at every step -- and made the phase slower, 11.5 -> 17.3 us, without helping anything else)
        float4* dk = reinterpret_cast<float4*>(s.dkeys_acc + ((int64_t)b * Tm + t) * Hd + u);   // this workgroup owns utterance b
        dk[0] = make_float4(dkv[0], dkv[1], dkv[2], dkv[3]);
        dk[1] = make_float4(dkv[4], dkv[5], dkv[6], dkv[7]);
        }
      }
      // d(attention_v): the P frame phases of a column meet in LDS first (the scratch behind the kernel's other arrays),
      // so a workgroup sends one atomic per column instead of P -- all 64 workgroups of a step add into the same Hd words
      float* dvs = red + 8 + Hd + (s.norm != LAS_NORM_SOFTMAX ? 5 * Tm : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) dvs[phase * Hd + u + j] = dv[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) dhs[phase * Hd + u + j] = a[j];
  }
  __syncthreads();
  if (att_additive(s.attention)) {
    const float* dvs = red + 8 + Hd + (s.norm != LAS_NORM_SOFTMAX ? 5 * Tm : 0);
    for (int u = tid; u < Hd; u += 256) {
      float acc = 0.f;
      for (int ph = 0; ph < P; ++ph) acc += dvs[ph * Hd + u];
      if (acc_run) acc_run[u] += acc;          // (the launch's running sum of this utterance: thread u owns column u)
      else atomicAdd(s.dv_acc + u, acc);
    }
  }
  LAS_STAMPB(s.step, 5);
  // reduce the P per-phase partials (dhs is [P][Hd] = 2048 floats)
  for (int u0 = 0; u0 < Hd; u0 += 256) {                // (whole waves: Hd is a multiple of 64)
    const int u = u0 + tid;
    float acc = 0.f;
    if (u < Hd)
      for (int ph = 0; ph < P; ++ph) acc += dhs[ph * Hd + u];
    if (xc)                                              // + the other three parts' frames, in part order
      for (int q = 0; q < 3; ++q) acc += seq_wait(xc->xdq + q * Hd + min(u, Hd - 1), u < Hd, xc->tag, xc->fail);
    if (u < Hd) red[8 + u] = acc;
  }
  __syncthreads();
  for (int u = tid; u < Hd; u += 256) dhs[u] = red[8 + u];
  __syncthreads();
  if (att_uses_wq(s.attention)) {
    // dhs holds d(processed query); save it (bf16) for d(query_layer) and map back: dh[u] = sum_a dpq[a] Wq[u][a]
    float* tmp = dhs + Hd;
    if (s.attention == LAS_ATT_CUSTOM) {          // through relu: the saved processed query is the relu output
      const float* pqv = s.pq + (int64_t)b * s.ldpq;
      for (int u = tid; u < Hd; u += 256) if (!(pqv[u] > 0.f)) dhs[u] = 0.f;
      __syncthreads();
    }
    for (int u = tid; u < Hd; u += 256)
      if (s.dpq_out) s.dpq_out[(int64_t)b * s.lddpq + u] = las_f2bf(dhs[u]);
    bool done = false;
    if constexpr (NPK == SEQ_NPK4) {
      // (the four-workgroup kernel has the registers: the product on the matrix cores from Wq's B-fragment image, d(processed
      //  query) rounded to bf16 -- the operand the d(query_layer) product sees too -- as row 0 of the A tile)
      if (wq_pk && (Hd == 256 || Hd == 128)) {
        unsigned short* opq = reinterpret_cast<unsigned short*>(tmp);
        for (int u = tid; u < Hd; u += 256) opq[u] = las_f2bf(dhs[u]);
        if (tid < 4) reinterpret_cast<unsigned*>(opq + Hd)[tid] = 0u;
        __syncthreads();
        if (Hd == 256) seq_matvec_square<8>(wq_pk, opq, opq + Hd, lane, wave, [&](int n, float v) { dhs[n] = v; });
        else seq_matvec_square<4>(wq_pk, opq, opq + Hd, lane, wave, [&](int n, float v) { dhs[n] = v; });
        __syncthreads();
        done = true;
      }
    }
    if (!done) {
      // (scratch: 2048 floats behind everything else of this kernel's LDS, see las_decoder_step_bwd)
      square_matvec_bf16(s.wq_t, dhs, tmp, red + 8 + Hd + (s.norm != LAS_NORM_SOFTMAX ? 5 * Tm : 0), Hd);
      for (int u = tid; u < Hd; u += 256) dhs[u] = tmp[u];
      __syncthreads();
    }
  }

  if (s.mode == LAS_DEC_ATTENTION_ONLY) {      // hand d(query) to the caller; the query's cell is differentiated elsewhere
    for (int u = tid; u < Hd; u += 256) s.dq_out[(int64_t)b * s.lddq + u] = dhs[u];
    return;
  }
  }
  LAS_STAMPB(s.step, 6);
  // ---- LSTM cell backward (Appendix F) ----
  for (int u = tid; u < Hd; u += 256) {
    const float* gp = s.gates + (int64_t)b * s.ldg + u;
    const float gi = gp[0], gj = gp[Hd], gf = gp[2 * Hd], go = gp[3 * Hd];
    const float ct = s.c_new[(int64_t)b * s.ldcn + u];
    const float cp = s.c_prev[(int64_t)b * s.ldcp + u];
    float dht = dhs[u];
    if (s.dh_rec) dht += s.dh_rec[(int64_t)b * s.ldr + u];
    if (s.dh_b) dht += s.dh_b[(int64_t)b * s.ldhb + u];
    if (s.dh_c) dht += s.dh_c[(int64_t)b * s.ldhc + u];
    const float tc = las_tanh(ct);
    const float dov = dht * tc * go * (1.f - go);
    const float dct = s.dc[(int64_t)b * Hd + u] + dht * go * (1.f - tc * tc);
    const float di = dct * gj * gi * (1.f - gi);
    const float dj = dct * gi * (1.f - gj * gj);
    const float df = dct * cp * gf * (1.f - gf);
    s.dc[(int64_t)b * Hd + u] = dct * gf;
    unsigned short* zp = s.dz + (int64_t)b * s.ldz + u;
    zp[0] = las_f2bf(di); zp[Hd] = las_f2bf(dj); zp[2 * Hd] = las_f2bf(df); zp[3 * Hd] = las_f2bf(dov);
  }
}

__global__ __launch_bounds__(256) void dec_step_bwd_kernel(las_dec_step_bwd s) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  dec_step_bwd_body(s, blockIdx.x, sm);
}

// ------------------------------------------------------------------------------------------------
// SEQUENTIAL backward decoder (round 3): all U steps of the single-cell decoders that the grouped one-launch kernel below does
// not cover -- an attention layer (attention_layer_size / --binf_projection) and / or a monotonic normaliser -- in ONE launch,
// one workgroup per utterance, NO exchange between workgroups: an utterance's backward chain only ever needs the shared
// weights.  Per step, last to first:
//   d(attention_t) = d(outputs)_t + d(feed)_{t+1}[:A]  (bf16, as the attention layer's products see it; saved for d(W_al))
//   d[query | context] = d(attention_t) W_al^T          (A x (Hd + M) from L2: 205 KB at cfg5)
//   the step body: d(context) -> d(alignments) -> normaliser backward -> d(scores) -> d(keys), d(query); LSTM cell backward
//   d(feed)_t = dz_t K^T                                 (4 Hd x (A + Hd) from L2: 688 KB at cfg5; a wave per output row)
// The per-step launches this replaces made five launches per step (818 per train step at cfg5, each a cold-L2 start); here the
// weights and the utterance's keys / values stay warm in the XCD's L2.  Results: the per-step path's, bit for bit (same body,
// same operand roundings, same summation order of the two products' K loops is NOT guaranteed -- compared at 1e-3).
// LDS: the step body's floats, then d(attention) [A], d[query | context] [Hd + M], d(feed) [W0].
// ------------------------------------------------------------------------------------------------
// KC = 4 Hd / 32: the 32-deep chunks of the gate columns (Hd 128: 16, 256: 32).  PARTS = 4 (Bahdanau scores, NPK > 0): four
// workgroups per utterance, blocks in chunks of 8 utterances (block = chunk * 32 + part * 8 + utterance % 8: the four parts of
// an utterance are 8 blocks apart, i.e. on one XCD under round-robin dispatch); see SeqXchg.
template <int NPK, int KC, int PARTS = 1>
__global__ __launch_bounds__(256) void dec_seq_bwd_kernel(las_dec_seq_bwd p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const las_dec_step_bwd& s0 = p.s;
  const int b = PARTS == 1 ? (int)blockIdx.x : (int)(blockIdx.x / 32) * 8 + (int)(blockIdx.x & 7);
  const int part = PARTS == 1 ? 0 : (int)(blockIdx.x % 32) / 8;
  const int tid0 = threadIdx.x;
  const int Hd = s0.Hd, M = s0.M, A = p.A, W0 = p.W0, feed = p.A > 0 ? p.A : M;
  // frames of this part, exchange granules of this utterance (PARTS = 4)
  const int fq = (s0.Tm + PARTS - 1) / PARTS, pf0 = part * fq, pf1 = min(s0.Tm, pf0 + fq);
  const SeqXLayout XL = seq_xlayout(s0.Tm, Hd, M, W0);
  pu64* const xbase = PARTS > 1 ? reinterpret_cast<pu64*>(static_cast<char*>(p.xchg_workspace) + 64) + (int64_t)min(b, s0.B - 1) * XL.total : nullptr;
  bool xlocal = false;
  if constexpr (PARTS > 1) {
    if (b < s0.B) {
      // are the four parts on one XCD?  (decides the flavour of the granule stores only)
      int* lw = reinterpret_cast<int*>(sm);
      if (tid0 == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        __hip_atomic_store(xbase + XL.xcc + part, ((pu64)1 << 32) | (xcc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        for (int m = 0; m < 4; ++m) {
          pu64 v = 0;
          unsigned spins = 0;
          do {
            v = __hip_atomic_load(xbase + XL.xcc + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 32) == 1) break;
            __builtin_amdgcn_s_sleep(2);
          } while (++spins < SEQ_SPIN_LIMIT);
          same = same && ((v >> 32) == 1) && ((unsigned)v == xcc + 1);
        }
        *lw = same ? 1 : 0;
      }
      __syncthreads();
      xlocal = *lw != 0;
      __syncthreads();
    }
    if (b >= s0.B || part > 0) {
      // ---- an absent utterance's workgroups only take their place in the fixed-order sums; parts 1..3 run the query path of
      //      their frames: wait for d(scores), tanh work with d(keys) in registers, partial d(processed query) back ----
      float* ds_l = sm;                                   // [Tm] (indexed by frame)
      float* dhs = sm + ((s0.Tm + 3) & ~3);               // [P][Hd] partial d(processed query), then [P][Hd] partial d(attention_v)
      float* run = dhs + 4096;                            // [Hd + 8]: running d(attention_v) of this part; [Hd + 4]: flag word
      int* fail = reinterpret_cast<int*>(run + Hd + 5);
      unsigned short* opl = reinterpret_cast<unsigned short*>(run + Hd + 8);      // [4 Hd] bf16 operand row (d(attention_t), then dz_t) + 8 zeros
      for (int n = tid0; n < Hd + 8; n += 256) run[n] = 0.f;
      if (tid0 < 4) reinterpret_cast<unsigned*>(opl + 4 * Hd)[tid0] = 0u;
      __syncthreads();
      if (b < s0.B) {
        float dkr[NPK > 0 ? NPK : 1][8];
#pragma unroll
        for (int i = 0; i < (NPK > 0 ? NPK : 1); ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) dkr[i][j] = 0.f;
        const int L = Hd / 8, P = 256 / L, phase = tid0 / L, u = (tid0 % L) * 8;
        const int len = min(s0.mem_len[b], s0.Tm), flen = min(len, pf1);
        const unsigned short* keys = s0.keys + (int64_t)b * s0.Tm * Hd;
        float vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] = s0.att_v[u + j];
        const int hl = tid0 & 63, hvw = part * 4 + __builtin_amdgcn_readfirstlane(tid0 >> 6);      // lane, virtual wave of the products
        for (int t = p.U - 1; t >= 0; --t) {
          const unsigned tag = (unsigned)(p.U - t);
          {
            // this part's quarter of d[query | context] = d(attention_t) W_al^T
            const float v = seq_wait(xbase + XL.xda + min(tid0, A / 2 - 1), tid0 < A / 2, tag, fail);
            if (tid0 < A / 2) reinterpret_cast<unsigned*>(opl)[tid0] = __float_as_uint(v);
            for (int a = A + tid0; a < ((A + 31) & ~31); a += 256) opl[a] = 0;
            __syncthreads();
            seq_matvec_few_chunks(p.waln_packed, (A + 31) / 32, (Hd + M) / 16, opl, opl + 4 * Hd, hl, hvw, 16,
                                  [&](int n, float v2) { pgranule_store(xbase + XL.xqc + n, tag, v2, xlocal); });
          }
          const float* pqv = s0.pq + (int64_t)b * s0.ldpq + (int64_t)t * p.inc_pq;
          float qq[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) qq[j] = pqv[u + j];
          for (int f = pf0; f < pf1; f += 256) {
            const float v = seq_wait(xbase + min(f + tid0, pf1 - 1), f + tid0 < pf1, tag, fail);
            if (f + tid0 < pf1) ds_l[f + tid0] = v;
          }
          __syncthreads();
          float a[8], dv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] = dv[j] = 0.f;
          if constexpr (NPK > 0) query_path_regk<NPK>(keys, Hd, pf0, flen, ds_l, qq, vv, phase, P, u, a, dv, dkr);
#pragma unroll
          for (int j = 0; j < 8; ++j) { dhs[phase * Hd + u + j] = a[j]; dhs[2048 + phase * Hd + u + j] = dv[j]; }
          __syncthreads();
          for (int c = tid0; c < Hd; c += 256) {
            float sa = 0.f, sv = 0.f;
            for (int ph = 0; ph < P; ++ph) { sa += dhs[ph * Hd + c]; sv += dhs[2048 + ph * Hd + c]; }
            pgranule_store(xbase + XL.xdq + (part - 1) * Hd + c, tag, sa, xlocal);
            run[c] += sv;
          }
          __syncthreads();
          if (t > 0 || p.dfeed_out) {
            // this part's quarter of d[feed | h]_t = dz_t K^T
            for (int i = tid0; i < 2 * Hd; i += 256)
              reinterpret_cast<unsigned*>(opl)[i] = __float_as_uint(seq_wait(xbase + XL.xdz + i, true, tag, fail));
            __syncthreads();
            seq_matvec_deep<KC>(p.kn_packed, (W0 + 15) / 16, opl, opl + 4 * Hd, hl, hvw, 16,
                                [&](int n, float v2) { if (n < W0) pgranule_store(xbase + XL.xdf + n, tag, v2, xlocal); });
            __syncthreads();
          }
        }
        if constexpr (NPK > 0) {
#pragma unroll
          for (int i = 0; i < NPK; ++i) {
            const int tt = pf0 + phase + i * P;
            if (tt < pf1) {
              float4* dk = reinterpret_cast<float4*>(s0.dkeys_acc + ((int64_t)b * s0.Tm + tt) * Hd + u);
              float4 a0 = dk[0], a1 = dk[1];
              a0.x += dkr[i][0]; a0.y += dkr[i][1]; a0.z += dkr[i][2]; a0.w += dkr[i][3];
              a1.x += dkr[i][4]; a1.y += dkr[i][5]; a1.z += dkr[i][6]; a1.w += dkr[i][7];
              dk[0] = a0;
              dk[1] = a1;
            }
          }
        }
        if (*fail && tid0 == 0) atomicOr(static_cast<unsigned*>(p.xchg_workspace), 32u);
      }
      if (p.sum_workspace && (s0.dv_acc || s0.dbias_acc)) {
        __syncthreads();
        ordered_accumulate(static_cast<unsigned*>(p.sum_workspace), blockIdx.x, gridDim.x, Hd + 1, run, s0.dv_acc, Hd, s0.dbias_acc,
                           reinterpret_cast<int*>(run + Hd + 4));
      }
      return;
    }
  }
  float* datt = sm + dec_step_bwd_floats(M, s0.Tm, Hd, s0.norm);      // [A] (bf16-rounded values)
  float* dqc = datt + (A > 0 ? A : 0);                                 // [Hd + M]
  float* dfeed = dqc + (A > 0 ? Hd + M : 0);                           // [W0] d[feed | h]_{t} from step t+1
  // [4 Hd] bf16 dz_t and 8 zeros behind it, on a 16-byte boundary (sm is): the A operand of the d(feed) product
  unsigned short* dzl = reinterpret_cast<unsigned short*>(sm + (((dfeed + W0 - sm) + 3) & ~(ptrdiff_t)3));
  float* acc_run = dfeed + W0 + 2 * Hd + 8;                            // [Hd + 1]: this utterance's d(attention_v), d(score_bias) over the steps
  float* vwl = acc_run + Hd + 8;                                       // [Tm][A + 1]: values W_c of this utterance (p.vw given)
  for (int n = tid0; n < W0; n += 256) dfeed[n] = 0.f;
  for (int n = tid0; n < Hd + 8; n += 256) acc_run[n] = 0.f;
  if (tid0 < 4) reinterpret_cast<unsigned*>(dzl + 4 * Hd)[tid0] = 0u;       // the zeros behind the A operands' row 0
  float dkr[NPK > 0 ? NPK : 1][8];                                    // d(keys) of this utterance (Bahdanau scores; see the body)
#pragma unroll
  for (int i = 0; i < (NPK > 0 ? NPK : 1); ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) dkr[i][j] = 0.f;
  const bool use_vw = A > 0 && p.vw != nullptr;
  if (use_vw) {
    const float* src = p.vw + (int64_t)b * p.ld_vw;
    for (int e = tid0; e < s0.Tm * A; e += 256) vwl[(e / A) * (A + 1) + e % A] = src[e];
  }
  __syncthreads();
  float dout_pre = (A > 0 && tid0 < A) ? (p.d_out + (int64_t)b * p.ld_dout + (int64_t)(p.U - 1) * p.inc_dout)[tid0] : 0.f;
  for (int t = p.U - 1; t >= 0; --t) {
    // The thread index is made opaque to the compiler once per step: everything that depends on it alone (row addresses of
    // the weights' fragments, of the keys' frames, bounds tests ...) is invariant across the steps and was hoisted out of
    // this loop -- some 200 registers held for the whole launch, more than the loop's own working set.
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool has_next = t + 1 < p.U;
    las_dec_step_bwd st = s0;
    st.mode = 0;
    st.gates = s0.gates + (int64_t)t * p.inc_gates;
    st.c_new = s0.c_new + (int64_t)t * p.inc_c;
    st.c_prev = s0.c_prev + (int64_t)t * p.inc_c;
    st.align = s0.align + (int64_t)t * p.inc_align;
    st.dz = s0.dz + (int64_t)t * p.inc_dz;
    st.ds_out = s0.ds_out + (int64_t)t * p.inc_ds;
    st.dctx_save = s0.dctx_save + (int64_t)t * p.inc_save;
    if (s0.pq) st.pq = s0.pq + (int64_t)t * p.inc_pq;
    if (s0.dpq_out) st.dpq_out = s0.dpq_out + (int64_t)t * p.inc_pq;
    if (s0.norm != LAS_NORM_SOFTMAX) {
      st.p = s0.p + (int64_t)t * p.inc_align;
      st.prev_align = t > 0 ? s0.align + (int64_t)(t - 1) * p.inc_align : nullptr;
      st.ldpa = s0.lda;
    }
    st.step = t;
    const float* dout = p.d_out + (int64_t)b * p.ld_dout + (int64_t)t * p.inc_dout;
    // d(outputs)_t[tid] was requested a step ago; the next step's goes out now (an input: no dependence on the chain)
    const float dout_cur = dout_pre;
    if (t > 0 && tid0 < A) dout_pre = (dout - p.inc_dout)[tid0];
    LAS_STAMPB(t, 0);
    if (A > 0) {
      // d(attention_t), rounded to bf16 once (the operand of both of the attention layer's backward products)
      for (int a = tid; a < A; a += 256) {
        const unsigned short v = las_f2bf((a < 256 ? dout_cur : dout[a]) + (has_next ? dfeed[a] : 0.f));
        datt[a] = las_bf2f(v);
        dzl[a] = v;
        p.datt_out[(int64_t)b * p.ld_datt + (int64_t)t * A + a] = v;
      }
      for (int a = A + tid; a < ((A + 31) & ~31); a += 256) dzl[a] = 0;
      __syncthreads();
      // d[query | context][n] = sum_a d(attention)[a] W_al[n][a] on the matrix cores (seq_matvec_few_chunks); with four workgroups
      // per utterance d(attention_t) goes out to the other three first, this one takes the tiles of virtual waves 0..3 and
      // collects the rest
      if constexpr (PARTS > 1) {
        if (tid < A / 2) pgranule_store(xbase + XL.xda + tid, (unsigned)(p.U - t), __uint_as_float(reinterpret_cast<const unsigned*>(dzl)[tid]), xlocal);
      }
      seq_matvec_few_chunks(p.waln_packed, (A + 31) / 32, (Hd + M) / 16, dzl, dzl + 4 * Hd, lane, wave, 4 * PARTS,
                            [&](int n, float v) { dqc[n] = v; });
      if constexpr (PARTS > 1) {
        int* fl = reinterpret_cast<int*>(acc_run + Hd + 5);
        for (int n0 = 0; n0 < Hd + M; n0 += 256) {
          const int n = n0 + tid;
          const bool have = n < Hd + M && seq_owner(n) != 0;
          const float v = seq_wait(xbase + XL.xqc + min(n, Hd + M - 1), have, (unsigned)(p.U - t), fl);
          if (have) dqc[n] = v;
        }
      }
      __syncthreads();
      LAS_STAMPB(t, 1);
      st.dctx_a = dqc + Hd;  st.ldda = 0;          // (LDS: the row stride is not used)
      st.dctx_b = nullptr;
      st.dh_b = dqc;         st.ldhb = 0;          // the attention layer's query input
      st.dh_c = nullptr;
      st.dh_rec = has_next ? dfeed + feed : nullptr;   st.ldr = 0;
    } else {
      // no attention layer: the context is the output and the feed
      st.dctx_a = dout - (int64_t)b * p.ld_dout;   st.ldda = p.ld_dout;       // (the body adds b * ldda back)
      st.dctx_b = has_next ? dfeed : nullptr;      st.lddb = 0;
      st.dh_b = st.dh_c = nullptr;
      st.dh_rec = has_next ? dfeed + feed : nullptr;   st.ldr = 0;
    }
    if constexpr (PARTS > 1) {
      const SeqXchg xc{xbase, xbase + XL.xdq, (unsigned)(p.U - t), reinterpret_cast<int*>(acc_run + Hd + 5), pf0, pf1, xlocal};
      dec_step_bwd_body<NPK>(st, b, sm, use_vw ? vwl : nullptr, datt, A, dkr, tid, p.sum_workspace ? acc_run : nullptr, &xc, p.wq_packed);
    } else {
      dec_step_bwd_body<NPK>(st, b, sm, use_vw ? vwl : nullptr, datt, A, dkr, tid, p.sum_workspace ? acc_run : nullptr);
    }
    __syncthreads();                               // dz_t of this utterance is in memory (same workgroup: visible behind the barrier)
    LAS_STAMPB(t, 8);
    if (t > 0 || p.dfeed_out) {
      // d[feed | h]_t[n] = sum_k dz_t[k] K[n][k]   (kn: [W0, 4 Hd] bf16, row n contiguous) on the matrix cores: dz_t is row 0 of
      // the A tile (the other 15 rows zero), a wave takes the 16-column tiles wave, wave + 4, ... and reads its B fragments
      // straight from kn (16 bytes per lane: 8 consecutive k of row n).  As multiply-adds on bf16 -> fp32 conversions (a wave
      // per output row, all pieces in flight) this product was ALU-bound: 23 of the 55 us of a step.
      // dz_t goes through LDS (2 KB): the A fragments are read from there at every product -- held in registers across the
      // tiles they cost 128 VGPRs, which the register-resident d(keys) needs.  Lanes of rows 1..15 read the zeros behind it.
      const unsigned short* dzr = st.dz + (int64_t)b * st.ldz;
      if (tid < Hd / 2) *reinterpret_cast<uint4*>(dzl + tid * 8) = ld16(dzr + tid * 8);
      __syncthreads();
      LAS_STAMPB(t, 10);
      if constexpr (PARTS > 1) {
        for (int i = tid; i < 2 * Hd; i += 256)
          pgranule_store(xbase + XL.xdz + i, (unsigned)(p.U - t), __uint_as_float(reinterpret_cast<const unsigned*>(dzl)[i]), xlocal);
      }
      const int NTL = (W0 + 15) / 16;
      seq_matvec_deep<KC>(p.kn_packed, NTL, dzl, dzl + 4 * Hd, lane, wave, 4 * PARTS, [&](int n, float v) { if (n < W0) dfeed[n] = v; });
      if constexpr (PARTS > 1) {
        int* fl = reinterpret_cast<int*>(acc_run + Hd + 5);
        for (int n0 = 0; n0 < NTL * 16; n0 += 256) {
          const int n = n0 + tid;
          const bool have = n < W0 && seq_owner(n) != 0;
          const float v = seq_wait(xbase + XL.xdf + min(n, W0 - 1), have, (unsigned)(p.U - t), fl);
          if (have) dfeed[n] = v;
        }
      }
    }
    __syncthreads();
    LAS_STAMPB(t, 9);
  }
  if (p.dfeed_out)
    for (int n = tid0; n < W0; n += 256) p.dfeed_out[(int64_t)b * W0 + n] = dfeed[n];
  if (p.sum_workspace && (s0.dv_acc || s0.dbias_acc)) {
    // d(attention_v) / d(score_bias): the utterances' sums over the steps meet in utterance order (ordered_accumulate)
    __syncthreads();
    ordered_accumulate(static_cast<unsigned*>(p.sum_workspace), blockIdx.x, gridDim.x, Hd + 1, acc_run, s0.dv_acc, Hd, s0.dbias_acc,
                       reinterpret_cast<int*>(acc_run + Hd + 4));
  }
  if (PARTS > 1 && tid0 == 0 && *reinterpret_cast<int*>(acc_run + Hd + 5)) atomicOr(static_cast<unsigned*>(p.xchg_workspace), 32u);
  if constexpr (NPK > 0) {
    // the register-resident d(keys) joins the accumulator once (same thread layout as the body's query path)
    const int L = Hd / 8, P = 256 / L, phase = tid0 / L, u = (tid0 % L) * 8;
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
      const int tt = pf0 + phase + i * P;
      if (tt < pf1) {
        float4* dk = reinterpret_cast<float4*>(s0.dkeys_acc + ((int64_t)b * s0.Tm + tt) * Hd + u);
        float4 a0 = dk[0], a1 = dk[1];
        a0.x += dkr[i][0]; a0.y += dkr[i][1]; a0.z += dkr[i][2]; a0.w += dkr[i][3];
        a1.x += dkr[i][4]; a1.y += dkr[i][5]; a1.z += dkr[i][6]; a1.w += dkr[i][7];
        dk[0] = a0;
        dk[1] = a1;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// PERSISTENT backward decoder (softmax attentions): all U steps, last to first, in one launch.  Same grouping as the
// forward (8 utterances per group of 32 workgroups on one XCD); the four workgroups of an utterance split its FRAMES:
//   S1  d(context) total; dalign over own frames (their quarter of the values); partial sum p.dalign      | barrier
//   S2  softmax backward on own frames (ds, saved as bf16 for the d(keys) GEMM); partial dh = sum ds*keys  | barrier
//   S3  part 0: dh total (+ recurrent gradient), LSTM cell backward -> dz_t, dc_{t-1}                      | barrier
//   G   d[attention_{t-1}, h_{t-1}] = dz_t K^T for the group's 8 utterances, member j owns the 16-column
//       tiles j, j+32, j+64 (its rows of K register-resident as MFMA B fragments)                          | barrier
// Every exchanged tensor has its own rows per step (no address is re-read after being rewritten).
// ------------------------------------------------------------------------------------------------
// NPQ = M / 128 when the written-out d(alignments) pass applies (LDS-resident frames, M = 512, 1024 or 2048), else 0.
// NT_MAX 16-column tiles of d[attention, h] per member (W / 16 / 32), KCW_MAX 32-deep chunks of the 4 Hd gate columns per wave
// of which KRES stay in registers (512 units: 5 tiles x 6 of 16 chunks resident, the rest streamed from L2 every step, three in
// flight); PD d(context) columns per thread (M <= 256 PD); UPT hidden units per thread in the cell backward (512 units: 2).
// TWO (round 4): a second decoder cell (decoder_layers = 2) inside the launch -- `s` describes cell 0 and the attention, the
// fields behind `workspace` cell 1.  Per step, cell 1 first: (cell backward) -> barrier -> P1 = dz1_t K1^T -> (cell backward) ->
// barrier -> P0 = dz0_t K0^T; the attention phases S1 / S2 / S3 belong to the cell that queries the attention: cell 1 in
// wiring 0 (MultiRNNCell inside the AttentionWrapper), cell 0 in wiring 1 (--bottom_only), the other cell's backward is the
// plain one (part 0 of each utterance, unit = thread).  Both products travel as granules; P1's weights are streamed from L2.
// Input dropout: the masks of the step-by-step path (draw (seed, in_stream_l + t, b * win_l + c)) where a product's columns
// are consumed.
template <bool WQ, int NPQ = 0, int NT_MAX = 3, int KRES = 8, int KCW_MAX = 8, int PD = 6, int UPT = 1, bool TWO = false>
__global__ __launch_bounds__(256) void dec_persist_bwd_kernel(las_dec_persist_bwd p) {
  static_assert(!TWO || UPT == 1, "second cell: decoder_units <= 256");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int RS2 = NT_MAX * 16 + 1;              // row stride of the G role's partial tiles
  const las_dec_step_bwd& s0 = p.s;
  const int B = s0.B, Hd = s0.Hd, M = s0.M, Tm = s0.Tm, W = p.W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  // blocks in chunks of 8 groups (block = chunk * 256 + member * 8 + group % 8): a group's 32 members are 8 blocks apart (one
  // XCD under round-robin dispatch), and the in-order dispatcher completes the 8 groups of a chunk -- one workgroup per CU of
  // a 256-CU device -- before it starts the next chunk: a batch of more than 64 utterances runs chunk after chunk
  const int groups = (B + 7) / 8;
  const int group = (blockIdx.x / (8 * P_MEMBERS)) * 8 + (blockIdx.x & 7), member = (blockIdx.x % (8 * P_MEMBERS)) >> 3;
  if (group >= groups) {
    // (a workgroup without utterances still takes its place in the fixed-order sum of d(attention_v): a row of zeros)
    if (WQ && att_additive(s0.attention) && p.sum_workspace) {
      for (int c = tid; c < Hd; c += 256) sm[c] = 0.f;
      __syncthreads();
      ordered_accumulate(static_cast<unsigned*>(p.sum_workspace), blockIdx.x, gridDim.x, Hd, sm, s0.dv_acc, Hd, nullptr, reinterpret_cast<int*>(sm + Hd));
    }
    return;
  }
  float* dctx = sm;                       // [M]
  unsigned short* dcb = reinterpret_cast<unsigned short*>(dctx);       // NPQ > 0: [2][M] bf16 instead, d(context) = high + low; then 16 B of zeros
  unsigned short* dsb = reinterpret_cast<unsigned short*>(dctx + M + 4);    // [2][ds_pad] bf16: ds of the own frames = high + low
  float* dal = dctx + M + 4 + persist_bwd_ds_pad(Tm);   // [Tm] dalign -> ds (own frames)
  float* alg = dal + Tm;                  // [Tm] alignments (own frames)
  float* dhs = alg + Tm;                  // [2048] per-phase partial dh
  float* red = dhs + 2048;                // [16 + Hd]
  float* red2 = red + 16 + Hd;            // [4 waves][8 utterances][RS2] partial output tiles of the G role
  int* fail = reinterpret_cast<int*>(red2 + persist_bwd_red_area(M, Hd, TWO));
  int* colo = fail + 1;
  unsigned* status = reinterpret_cast<unsigned*>(p.workspace);
  pu64* flags = reinterpret_cast<pu64*>(reinterpret_cast<char*>(p.workspace) + 64) + (size_t)group * 2 * P_MEMBERS;
  pu64* xcc_tab = flags + P_MEMBERS;
  pu64* xdot = reinterpret_cast<pu64*>(reinterpret_cast<char*>(p.workspace) + 64) + persist_flag_words(B);   // [2][B][4]
  pu64* xdh = xdot + 2 * (size_t)B * 4;                                                                        // [2][B][4][Hd]
  pu64* xdq = xdh + 2 * (size_t)B * 4 * Hd;                                                                    // [2][B][Hd]
  pu64* xdf = xdq + 2 * (size_t)B * Hd;                                                                        // [2][B][W]
  pu64* xdf1 = xdf + 2 * (size_t)B * W;                                                                        // [2][B][W1] (TWO)
  const int W1 = TWO ? p.W1 : 0;
  const bool w1 = TWO && p.wiring == 1;
  // the cell the attention phases belong to (A) and the plain one (P): saved tensors, d(c), dz rows
  const bool a_is_1 = TWO && !w1;
  const float* gatesA = a_is_1 ? p.gates1 : s0.gates;
  const float* cnewA = a_is_1 ? p.c1 + Hd : s0.c_new;
  const float* cprevA = a_is_1 ? p.c1 : s0.c_prev;
  unsigned short* dzA = a_is_1 ? p.dz1 : s0.dz;
  float* dcA = a_is_1 ? p.dc1 : s0.dc;
  const float* gatesP = a_is_1 ? s0.gates : p.gates1;
  const float* cnewP = a_is_1 ? s0.c_new : p.c1 + Hd;
  const float* cprevP = a_is_1 ? s0.c_prev : p.c1;
  unsigned short* dzP = a_is_1 ? s0.dz : p.dz1;
  float* dcP = a_is_1 ? s0.dc : p.dc1;
  const float keep = s0.drop_keep, inv_keep = 1.0f / s0.drop_keep;
  if (tid == 0) { *fail = 0; *colo = 0; }
  __syncthreads();
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    __hip_atomic_store(xcc_tab + member, ((pu64)1 << 32) | (xcc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool same = true;
    for (int m = 0; m < P_MEMBERS; ++m) {
      pu64 v = 0;
      unsigned spins = 0;
      do {
        v = __hip_atomic_load(xcc_tab + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 32) == 1) break;
        __builtin_amdgcn_s_sleep(2);
      } while (++spins < P_SPIN_LIMIT);
      same = same && ((v >> 32) == 1) && ((unsigned)v == xcc + 1);
    }
    *colo = same ? 1 : 0;
  }
  __syncthreads();
  const bool local = *colo != 0;

  // G role: rows of kc ([W, 4Hd] bf16: row n = output column n) for this member's tiles, register-resident
  const int NTW = W / 16, KC = 4 * Hd / 32;
  bf16x8 wf[NT_MAX][KRES];
  const unsigned short* wrow[NT_MAX];          // this lane's row of kc per tile (absent tiles: the last one's, results dropped)
#pragma unroll
  for (int j = 0; j < NT_MAX; ++j) {
    const int tile = member + 32 * j;
    wrow[j] = p.kc + (int64_t)(min(tile, NTW - 1) * 16 + l15) * p.ldk + 8 * lq;
#pragma unroll
    for (int i = 0; i < KRES; ++i) {
      const int kc = wave + 4 * i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (tile < NTW && kc < KC) v = *reinterpret_cast<const uint4*>(wrow[j] + kc * 32);
      wf[j][i] = __builtin_bit_cast(bf16x8, v);
    }
  }
  const int bg = group * 8 + (l15 & 7);
  const int b = group * 8 + member / 4, part = member & 3;
  const bool active = b < B;
  const int len = active ? min(s0.mem_len[b], Tm) : 0;
  const int fq = (Tm + 3) / 4, f0 = part * fq, f1 = min(Tm, f0 + fq), flen = min(len, f1);
  const unsigned short* vals = s0.values + (int64_t)(active ? b : 0) * Tm * M;
  const unsigned short* keys = s0.keys + (int64_t)(active ? b : 0) * Tm * Hd;
  // the frames of this workgroup do not change over the U steps: keep them in LDS when they fit (every step would
  // otherwise stream them from L2 / Infinity Cache again, four dependent round trips in S1 alone)
  constexpr bool KT = !WQ && NPQ > 0;     // Luong with the matrix-core passes: the resident keys are transposed
  // value frames [f0, f0 + vfr) of this workgroup are in LDS (vfr = fq unless the memory is long: then the rest is streamed)
  const int vfr = (KT || NPQ > 0) ? (persist_bwd_resident(M, Hd, Tm, KT, TWO) ? fq : 0) : persist_bwd_resident_rows(M, Hd, Tm, TWO);
  const bool resident = vfr > 0;
  const int MS = M + P_VPAD, FS = persist_bwd_kt_stride(Tm), DSP = persist_bwd_ds_pad(Tm);
  unsigned short* lvals = reinterpret_cast<unsigned short*>(sm + persist_bwd_scratch_floats(M, Hd, Tm, TWO));   // [vfr][MS]
  unsigned short* lkeys = lvals + (size_t)vfr * MS;                                                          // [fq][Hd]
  if (tid < 8) dcb[2 * M + tid] = 0;
  for (int e = tid; e < 2 * DSP; e += 256) dsb[e] = 0;
  if (resident && active) {
    const int nrow = max(f1 - f0, 0);
    for (int e = tid; e < min(nrow, vfr) * (M / 8); e += 256) {
      const int r = e / (M / 8), c = e % (M / 8);
      *reinterpret_cast<uint4*>(lvals + (size_t)r * MS + c * 8) = *reinterpret_cast<const uint4*>(vals + (int64_t)(f0 + r) * M + c * 8);
    }
    if constexpr (KT) {
      // (consecutive lanes take consecutive frames of the same 8 units: the 2-byte LDS stores of a wave fall on 32 banks;
      // frames past the utterance are zeros)
      for (int e = tid; e < FS * (Hd / 8); e += 256) {
        const int r = e % FS, c = e / FS;
        const uint4 v = f0 + r < flen ? *reinterpret_cast<const uint4*>(keys + (int64_t)(f0 + r) * Hd + c * 8) : make_uint4(0, 0, 0, 0);
        const unsigned short* e8 = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) lkeys[(size_t)(c * 8 + j) * FS + r] = e8[j];
      }
    } else {
      for (int e = tid; e < nrow * (Hd / 8); e += 256) {
        const int r = e / (Hd / 8), c = e % (Hd / 8);
        *reinterpret_cast<uint4*>(lkeys + (size_t)r * Hd + c * 8) = *reinterpret_cast<const uint4*>(keys + (int64_t)(f0 + r) * Hd + c * 8);
      }
    }
    __syncthreads();
  }
  unsigned epoch = 0;
  // S3 runs on part 0 with unit = threadIdx.x (Hd <= 256): d(c) stays in a register over the steps, and the saved
  // values of a step are requested at its top, two barriers before the cell backward uses them
  const bool cellw = active && part == 0 && tid < Hd;
  float dcr[UPT];
#pragma unroll
  for (int q = 0; q < UPT; ++q) dcr[q] = cellw ? dcA[(int64_t)b * Hd + tid + q * 256] : 0.f;
  float dcp = (TWO && cellw) ? dcP[(int64_t)b * Hd + tid] : 0.f;          // d(c) of the plain cell
  // input-dropout factor of element (b, col) of cell l's input row at step tt (1 without dropout)
  auto in_mask = [&](int l, int tt, int col) -> float {
    if (!(keep < 1.0f)) return 1.0f;
    const int win = l == 0 ? p.win0 : p.win1;
    if (l == 0) col += p.win0 - M;       // (a dense token vector in front of cell 0's attention feed)
    return las_uniform(s0.drop_seed, (l == 0 ? p.in_stream0 : p.in_stream1) + (unsigned)tt, (unsigned long long)b * win + col) < keep ? inv_keep : 0.f;
  };
  // the plain cell's backward at step t (part 0, unit = thread): dh from the sources the wiring names, dz_t (bf16) out
  auto plain_cell = [&](int t, unsigned xtag, bool first) {
    if (!cellw) return;
    float dht = 0.f;
    const pu64* xf0 = xdf + ((size_t)((xtag - 1) & 1) * B + b) * W;
    const pu64* xf1p = xdf1 + ((size_t)((xtag - 1) & 1) * B + b) * W1;
    const pu64* xf1c = xdf1 + ((size_t)(xtag & 1) * B + b) * W1;
    unsigned spins = 0;
    for (;;) {
      // wiring 0 (cell 0): d(h0_t) = mask . P1_t[tid] + P0_{t+1}[M + tid];  wiring 1 (cell 1): d_out_t[tid] + P1_{t+1}[2 M + tid]
      const pu64 ga = w1 ? ((pu64)xtag << 32) : pgranule_load(xf1c + tid);
      const pu64 gb = first ? ((pu64)(xtag - 1) << 32) : pgranule_load(w1 ? xf1p + 2 * M + tid : xf0 + M + tid);
      if (__all((unsigned)(ga >> 32) == xtag && (unsigned)(gb >> 32) == xtag - 1)) {
        dht = __uint_as_float((unsigned)gb);
        if (w1) dht += p.d_out1[(int64_t)b * p.ld_dout1 + (int64_t)t * p.inc_dout1 + tid];
        else dht += __uint_as_float((unsigned)ga) * in_mask(1, t, tid);
        break;
      }
      if (++spins > P_SPIN_LIMIT || *fail) { if (lane == 0) *fail = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    const float* gp = gatesP + (int64_t)b * s0.ldg + (int64_t)t * p.inc_gates + tid;
    const float gi = gp[0], gj = gp[Hd], gf = gp[2 * Hd], go = gp[3 * Hd];
    const float ct = cnewP[(int64_t)b * s0.ldcn + (int64_t)t * p.inc_c + tid], cp = cprevP[(int64_t)b * s0.ldcp + (int64_t)t * p.inc_c + tid];
    const float tc = las_tanh(ct);
    const float dov = dht * tc * go * (1.f - go);
    const float dct = dcp + dht * go * (1.f - tc * tc);
    const float di = dct * gj * gi * (1.f - gi);
    const float dj = dct * gi * (1.f - gj * gj);
    const float df = dct * cp * gf * (1.f - gf);
    dcp = dct * gf;
    unsigned short* zp = dzP + (int64_t)b * s0.ldz + (int64_t)t * p.inc_dz + tid;
    zp[0] = las_f2bf(di); zp[Hd] = las_f2bf(dj); zp[2 * Hd] = las_f2bf(df); zp[3 * Hd] = las_f2bf(dov);
  };
  // P1 = dz1_t K1^T for the group's 8 utterances: member j owns the 16-column tiles j, j + 32, ... of its W1 columns; K1's
  // rows are streamed from L2 (four 32-deep chunks in flight per wave)
  auto product1 = [&](int t, unsigned xtag) {
    constexpr int NT1 = TWO ? 5 : 1, RS1 = NT1 * 16 + 1;       // up to 5 column tiles per member (2 M + Hd <= 2560 columns)
    const int NTW1 = W1 / 16, KC1 = 4 * Hd / 32;
    f32x4 acc[NT1];
#pragma unroll
    for (int j = 0; j < NT1; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int bgc = min(group * 8 + (l15 & 7), B - 1);
    const unsigned short* zrow = (l15 < 8) ? p.dz1 + (int64_t)bgc * s0.ldz + (int64_t)t * p.inc_dz + 8 * lq
                                           : reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(p.workspace) + 16);
    const int zstep = (l15 < 8) ? 32 : 0;
    const unsigned short* w1row[NT1];
#pragma unroll
    for (int j = 0; j < NT1; ++j) w1row[j] = p.k1c + (int64_t)(min(member + 32 * j, NTW1 - 1) * 16 + l15) * p.ldk1 + 8 * lq;
    constexpr int SB = 4;
#pragma unroll 1
    for (int i0 = 0; wave + 4 * i0 < KC1; i0 += SB) {
      uint4 a[SB], w[SB][NT1];
#pragma unroll
      for (int q = 0; q < SB; ++q) {
        const int kc = wave + 4 * (i0 + q), kcc = min(kc, KC1 - 1);
        a[q] = *reinterpret_cast<const uint4*>(zrow + kcc * zstep);
        if (kc >= KC1) a[q] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT1; ++j) w[q][j] = *reinterpret_cast<const uint4*>(w1row[j] + kcc * 32);
      }
#pragma unroll
      for (int q = 0; q < SB; ++q)
#pragma unroll
        for (int j = 0; j < NT1; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[q]), __builtin_bit_cast(bf16x8, w[q][j]), acc[j], 0, 0, 0);
    }
    if (lq < 2) {
#pragma unroll
      for (int j = 0; j < NT1; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red2[(wave * 8 + lq * 4 + r) * RS1 + j * 16 + l15] = acc[j][r];
    }
    lds_barrier();
    for (int e = tid; e < 8 * NT1 * 16; e += 256) {
      const int row = e / (NT1 * 16), c = e % (NT1 * 16), j = c / 16, col = c % 16;
      const int tile = member + 32 * j, bb = group * 8 + row;
      if (tile < NTW1 && bb < B) {
        const float v = red2[(0 * 8 + row) * RS1 + c] + red2[(1 * 8 + row) * RS1 + c] +
                        red2[(2 * 8 + row) * RS1 + c] + red2[(3 * 8 + row) * RS1 + c];
        if (t == 0 && p.dfeed1_all) p.dfeed1_all[(int64_t)bb * W1 + tile * 16 + col] = v;      // the caller's d(initial state of cell 1)
        pgranule_store(xdf1 + ((size_t)(xtag & 1) * B + bb) * W1 + tile * 16 + col, xtag, v, local);
      }
    }
    lds_barrier();                          // (red2 is written again by the next product)
  };
  // likewise the step's row of d(context) from the projection layer and its alignments: step t-1's are requested
  // at the top of step t (PD values per thread cover M <= 256 * PD)
  float cur_dc[PD], cur_al = 0.f;
  auto fetch_step = [&](int tt_, float (&dcv)[PD], float& alv) {
#pragma unroll
    for (int i = 0; i < PD; ++i) {
      const int m = tid + i * 256;
      dcv[i] = (active && m < M && (!TWO || s0.dctx_a)) ? s0.dctx_a[(int64_t)b * s0.ldda + (int64_t)tt_ * p.inc_a + m] : 0.f;
    }
    const int fr = f0 + tid;
    alv = (active && fr < f1 && fr < len) ? (s0.align + (int64_t)b * s0.lda + (int64_t)tt_ * p.inc_align)[fr] : 0.f;
  };
  fetch_step(p.U - 1, cur_dc, cur_al);

  // d(attention_v) of this thread's 8 columns over ALL steps (Bahdanau): one round of atomics at the end of the launch
  // instead of 2 048 per workgroup and step into the same Hd words (that contention was most of the Bahdanau backward)
  float dv_tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int t = p.U - 1; t >= 0; --t) {
    const bool first = (t == p.U - 1);
    LAS_STAMPB(p.U - 1 - t, 0);
    float sg[UPT][4], sct[UPT], scp[UPT], sdf[UPT];
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
      sg[q][0] = sg[q][1] = sg[q][2] = sg[q][3] = 0.f;
      sct[q] = scp[q] = sdf[q] = 0.f;
      if (cellw) {
        const float* gp = gatesA + (int64_t)b * s0.ldg + (int64_t)t * p.inc_gates + tid + q * 256;
        sg[q][0] = gp[0]; sg[q][1] = gp[Hd]; sg[q][2] = gp[2 * Hd]; sg[q][3] = gp[3 * Hd];
        sct[q] = cnewA[(int64_t)b * s0.ldcn + (int64_t)t * p.inc_c + tid + q * 256];
        scp[q] = cprevA[(int64_t)b * s0.ldcp + (int64_t)t * p.inc_c + tid + q * 256];
      }
    }
    float nxt_dc[PD], nxt_al = 0.f;
    if (t > 0) fetch_step(t - 1, nxt_dc, nxt_al);
    const unsigned xtag = (unsigned)(p.U - t);       // 1, 2, ...: tag of this step's granules, parity slot xtag & 1
    if constexpr (TWO) {
      if (w1) {
        // ---- cell 1 (plain; its h1_t is the decoder's output) -> barrier -> P1_t ----
        if (active && part == 0) plain_cell(t, xtag, first);
        if (!persist_barrier(flags, member, ++epoch, local, fail)) break;
        product1(t, xtag);
      }
    }
    // ---- S1 ----
    if (active) {
      // the feed gradient of step t+1 arrives as granules from that step's 32 product slices (tag xtag - 1): this
      // thread's columns of the attention part, and on part 0 its unit of the h part (wave-uniform, bounded polling)
      float fbv[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) fbv[i] = 0.f;
      if constexpr (TWO) {
        // sources of d(context)_t and of the attention cell's recurrent gradient: step t+1's products (tag xtag - 1; none at
        // the first step) and, in wiring 1, this step's P1; the masks of the cell inputs they went through
        const pu64* xf0 = xdf + ((size_t)((xtag - 1) & 1) * B + b) * W;
        const pu64* xf1p = xdf1 + ((size_t)((xtag - 1) & 1) * B + b) * W1;
        const pu64* xf1c = xdf1 + ((size_t)(xtag & 1) * B + b) * W1;
        unsigned spins = 0;
        for (;;) {
          bool got = true;
          pu64 g0[PD], g1[PD], g2[PD];
#pragma unroll
          for (int i = 0; i < PD; ++i) {
            const int m = tid + i * 256;
            g0[i] = (!first && m < M) ? pgranule_load(xf0 + m) : ((pu64)(xtag - 1) << 32);
            g1[i] = (w1 && !first && m < M) ? pgranule_load(xf1p + M + m) : ((pu64)(xtag - 1) << 32);
            g2[i] = (w1 && m < M) ? pgranule_load(xf1c + m) : ((pu64)xtag << 32);
            got = got && ((unsigned)(g0[i] >> 32) == xtag - 1) && ((unsigned)(g1[i] >> 32) == xtag - 1) && ((unsigned)(g2[i] >> 32) == xtag);
          }
          const pu64 gr = (cellw && !first) ? pgranule_load(w1 ? xf0 + M + tid : xf1p + Hd + tid) : ((pu64)(xtag - 1) << 32);
          got = got && ((unsigned)(gr >> 32) == xtag - 1);
          if (__all(got)) {
#pragma unroll
            for (int i = 0; i < PD; ++i) {
              const int m = tid + i * 256;
              if (m < M) {
                fbv[i] = __uint_as_float((unsigned)g0[i]) * in_mask(0, t + 1, m);
                if (w1) fbv[i] += __uint_as_float((unsigned)g1[i]) * in_mask(1, t + 1, M + m) + __uint_as_float((unsigned)g2[i]) * in_mask(1, t, m);
              }
            }
            if (cellw) sdf[0] = __uint_as_float((unsigned)gr);
            break;
          }
          if (++spins > P_SPIN_LIMIT || *fail) { if (lane == 0) *fail = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      } else if (!first) {
        const pu64* xf = xdf + ((size_t)((xtag - 1) & 1) * B + b) * W;
        unsigned spins = 0;
        for (;;) {
          bool got = true;
          pu64 gq[PD + UPT];
#pragma unroll
          for (int i = 0; i < PD; ++i) {
            const int m = tid + i * 256;
            gq[i] = m < M ? pgranule_load(xf + m) : ((pu64)(xtag - 1) << 32);
            got = got && ((unsigned)(gq[i] >> 32) == xtag - 1);
          }
#pragma unroll
          for (int q = 0; q < UPT; ++q) {
            gq[PD + q] = cellw ? pgranule_load(xf + M + tid + q * 256) : ((pu64)(xtag - 1) << 32);
            got = got && ((unsigned)(gq[PD + q] >> 32) == xtag - 1);
          }
          if (__all(got)) {
#pragma unroll
            for (int i = 0; i < PD; ++i) fbv[i] = __uint_as_float((unsigned)gq[i]);
#pragma unroll
            for (int q = 0; q < UPT; ++q)
              if (cellw) sdf[q] = __uint_as_float((unsigned)gq[PD + q]);
            break;
          }
          if (++spins > P_SPIN_LIMIT || *fail) { if (lane == 0) *fail = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      LAS_STAMPB(p.U - 1 - t, 1);
#pragma unroll
      for (int i = 0; i < PD; ++i) {
        const int m = tid + i * 256;
        if (m >= M) break;
        float v = cur_dc[i];
        if constexpr (TWO) {
          v += fbv[i];                   // (already through the masks)
        } else if (!first) {
          float fb = fbv[i];
          if (s0.drop_keep < 1.0f) {     // gradient through step t+1's input dropout of the attention feed
            const unsigned long long idx = ((unsigned long long)(t + 1) * B + b) * s0.feed_width + (s0.feed_width - M) + m;
            fb = las_uniform(s0.drop_seed, s0.drop_stream, idx) < s0.drop_keep ? fb / s0.drop_keep : 0.f;
          }
          v += fb;
        }
        if constexpr (NPQ > 0) {
          const unsigned short hi = las_f2bf(v);
          dcb[m] = hi;
          dcb[M + m] = las_f2bf(v - las_bf2f(hi));
        } else {
          dctx[m] = v;
        }
        if (part == 0 && s0.dctx_save) s0.dctx_save[(int64_t)b * s0.ldds + (int64_t)t * p.inc_save + m] = las_f2bf(v);
      }
      if (f0 + tid < f1) alg[f0 + tid] = cur_al;
      {                                                      // frame shares beyond 256 frames (Tm > 1024): the rest from memory
        const float* arow = s0.align + (int64_t)b * s0.lda + (int64_t)t * p.inc_align;
        for (int tt = f0 + 256 + tid; tt < f1; tt += 256) alg[tt] = tt < len ? arow[tt] : 0.f;
      }
      lds_barrier();
      LAS_STAMPB(p.U - 1 - t, 2);
      {
        const int sub = lane & 15, grp = lane >> 4;
        // rows: the frames' value rows, from LDS (row 0 = frame f0) or from memory (row 0 = frame 0)
        // the fast form (LDS-resident frames, M a multiple of 128): on the matrix cores.  One 16x16x32 product per 16
        // frames and 32 columns: A = the frames' value rows as they lie in LDS (bf16, no conversion), B = d(context) as
        // two bf16 columns (high half, low half -- together 16 mantissa bits) and 14 zero columns.  Wave w takes the
        // frame tiles w, w+4, ...
        constexpr bool fast = NPQ > 0;
        if constexpr (fast) {
          const lds_cu16 bsrc = (lds_cu16)dcb + (sub < 2 ? sub * M + grp * 8 : 2 * M);
          const int bstep = sub < 2 ? 32 : 0;
          for (int t0 = f0 + wave * 16; t0 < f1; t0 += 64) {
            const int ta = t0 + sub;
            const lds_cu16 ra = (lds_cu16)lvals + (size_t)(ta < flen ? ta - f0 : 0) * MS + grp * 8;
            // (M / 32 = 4 NPQ chunks, written out four at a time: the next four chunks' fragments are requested before
            // this four's products, so that the LDS latency is paid once, not per chunk)
            constexpr int KC = 4 * NPQ, CH = NPQ > 8 ? 2 : 4;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            uint4 av[2][CH], bv[2][CH];
            auto request = [&](int buf, int c) {
#pragma unroll
              for (int j = 0; j < CH; ++j) { av[buf][j] = ld16(ra + (c * CH + j) * 32); bv[buf][j] = ld16(bsrc + (c * CH + j) * bstep); }
            };
            auto products = [&](int buf) {
#pragma unroll
              for (int j = 0; j < CH; j += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[buf][j]), __builtin_bit_cast(bf16x8, bv[buf][j]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[buf][j + 1]), __builtin_bit_cast(bf16x8, bv[buf][j + 1]), acc1, 0, 0, 0);
              }
            };
            request(0, 0);
            // (the 512-unit variants have no registers to spare: their loop stays rolled, two chunks of four per turn)
#pragma unroll(NPQ > 8 ? 1 : KC / CH / 2)
            for (int c = 0; c < KC / CH; c += 2) {
              request(1, c + 1);
              __builtin_amdgcn_sched_barrier(0);
              products(0);
              __builtin_amdgcn_sched_barrier(0);
              if (c + 2 < KC / CH) request(0, c + 2);
              __builtin_amdgcn_sched_barrier(0);
              products(1);
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = acc0[r] + acc1[r];
              v += las_dpp<0xB1>(v);                       // column 0 (high half) + column 1 (low half)
              const int tr = t0 + grp * 4 + r;
              if (sub == 0 && tr < f1) dal[tr] = tr < flen ? v : 0.f;
            }
          }
        }
        // frames [fa, fe) of the own quarter
        auto dalign_pass = [&](const unsigned short* rows, int row0, int stride, int fa, int fe) {
          const int fle = min(flen, fe);
          for (int t0 = fa; t0 < fe; t0 += 32) {
            const int ta = t0 + wave * 4 + grp, tb = ta + 16;
            float acc_a = 0.f, acc_b = 0.f;
            const bool oa = ta < fle, ob = tb < fle;
            const unsigned short* ra = rows + (int64_t)(oa ? ta - row0 : 0) * stride;
            const unsigned short* rb = rows + (int64_t)(ob ? tb - row0 : 0) * stride;
#pragma unroll 4
            for (int k = sub * 8; k < M; k += 128) {
              const uint4 va = *reinterpret_cast<const uint4*>(ra + k);
              const uint4 vb = *reinterpret_cast<const uint4*>(rb + k);
              acc_a += dot8(va, dctx + k);
              acc_b += dot8(vb, dctx + k);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
              acc_a += __shfl_xor(acc_a, o, 64);
              acc_b += __shfl_xor(acc_b, o, 64);
            }
            if (sub == 0) {
              if (ta < fe) dal[ta] = oa ? acc_a : 0.f;
              if (tb < fe) dal[tb] = ob ? acc_b : 0.f;
            }
          }
        };
        if constexpr (fast) { }
        else if (resident) {
          dalign_pass(lvals, f0, MS, f0, min(f1, f0 + vfr));
          if (f0 + vfr < f1) dalign_pass(vals, 0, M, f0 + vfr, f1);        // long memories: the frames that did not fit the LDS
        } else dalign_pass(vals, 0, M, f0, f1);
      }
      lds_barrier();
      LAS_STAMPB(p.U - 1 - t, 3);
      float dot = 0.f;
      for (int tt = f0 + tid; tt < flen; tt += 256) dot += alg[tt] * dal[tt];
      dot = las_wave_sum_dpp(dot);                          // (DPP row sums + one barrier instead of 12 LDS permutes + two)
      if (lane == 0) red[8 + wave] = dot;
      lds_barrier();
      if (tid == 0) pgranule_store(xdot + ((size_t)(xtag & 1) * B + b) * 4 + part, xtag, red[8] + red[9] + red[10] + red[11], local);
    }
    LAS_STAMPB(p.U - 1 - t, 4);
    // ---- S2 ----  (the four partial dots meet as granules: no group barrier)
    if (active) {
      if (tid < 64) {
        const pu64* xd = xdot + ((size_t)(xtag & 1) * B + b) * 4;
        float v = 0.f;
        unsigned spins = 0;
        for (;;) {
          const pu64 gq = lane < 4 ? pgranule_load(xd + lane) : ((pu64)xtag << 32);
          if (__all((unsigned)(gq >> 32) == xtag)) { v = __uint_as_float((unsigned)gq); break; }
          if (++spins > P_SPIN_LIMIT || *fail) { if (lane == 0) *fail = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        const float d0 = __shfl(v, 0, 64), d1 = __shfl(v, 1, 64), d2 = __shfl(v, 2, 64), d3 = __shfl(v, 3, 64);
        if (lane == 0) red[0] = d0 + d1 + d2 + d3;
      }
      lds_barrier();
      LAS_STAMPB(p.U - 1 - t, 5);
      const float dot = red[0];
      unsigned short* dso = s0.ds_out + (int64_t)b * s0.ldso + (int64_t)t * p.inc_ds;
      for (int tt = f0 + tid; tt < f1; tt += 256) {
        const float v = (tt < len) ? alg[tt] * (dal[tt] - dot) : 0.f;
        const unsigned short hi = las_f2bf(v);
        dal[tt] = v;
        dso[tt] = hi;
        if constexpr (KT) { dsb[tt - f0] = hi; dsb[DSP + tt - f0] = las_f2bf(v - las_bf2f(hi)); }
      }
      lds_barrier();
      LAS_STAMPB(p.U - 1 - t, 6);
      float dh_own[UPT];
      const int L = Hd / 8, P = 256 / L;
      if constexpr (KT) {
        // dh[u] = sum over the own frames of ds[t'] keys[t'][u] on the matrix cores: A = ds (rows 0, 4, 8, 12 the high
        // halves, rows 1, 5, 9, 13 the low halves, the rest zeros), B = the transposed keys of 16 units.  Every quarter
        // of the wave ends up with the tile's sums, so lane l of wave w keeps unit 64 w + l (+ 256 q): the per-phase
        // partial sums in LDS, their barrier and their 8 reads per unit are gone
        const int sub = lane & 15, grp = lane >> 4;
        const lds_cu16 asrc = (lds_cu16)((sub & 3) < 2 ? dsb + (sub & 3) * DSP + grp * 8 : dcb + 2 * M);
        const int astep = (sub & 3) < 2 ? 32 : 0;
        f32x4 acc[UPT][4];
#pragma unroll
        for (int q = 0; q < UPT; ++q)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[q][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc * 32 < f1 - f0; ++kc) {
          const int kk = kc * 32 + grp * 8;
          const uint4 av = ld16(asrc + kc * astep);
          uint4 bv[UPT][4];
#pragma unroll
          for (int q = 0; q < UPT; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g)          // (waves beyond Hd / 64 reread wave 0's units: their sums are not used)
              bv[q][g] = ld16((lds_cu16)lkeys + (size_t)((q * 256 + wave * 64) % Hd + g * 16 + sub) * FS + (kk < FS ? kk : 0));
#pragma unroll
          for (int q = 0; q < UPT; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              acc[q][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv[q][g]), acc[q][g], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
          const float r0 = acc[q][0][0] + acc[q][0][1], r1 = acc[q][1][0] + acc[q][1][1];
          const float r2 = acc[q][2][0] + acc[q][2][1], r3 = acc[q][3][0] + acc[q][3][1];
          dh_own[q] = grp == 0 ? r0 : (grp == 1 ? r1 : (grp == 2 ? r2 : r3));
        }
        LAS_STAMPB(p.U - 1 - t, 7);
      } else {
      const int phase = tid / L, u = (tid % L) * 8;
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = 0.f;
      if (!WQ || !att_additive(s0.attention)) {
        auto dh_pass = [&](const unsigned short* rows, int row0) {
#pragma unroll 4
          for (int tt = f0 + phase; tt < flen; tt += P) {
            const uint4 v = *reinterpret_cast<const uint4*>(rows + (int64_t)(tt - row0) * Hd + u);
            const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
            const float d = dal[tt];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += d * las_bf2f(e[j]);
          }
        };
        if (resident) dh_pass(lkeys, f0);
        else dh_pass(keys, 0);
      } else {
        // Bahdanau: score = sum_a v[a] tanh(keys[t',a] + pq[a]); d_pre = ds * v * (1 - tanh^2).  This workgroup owns its
        // frames of the utterance: d(keys) is accumulated in place, d(attention_v) with atomics.
        const float* pqv = s0.pq + (int64_t)b * s0.ldpq + (int64_t)t * p.inc_pq;
        float dv[8], vv[8], qq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { dv[j] = 0.f; vv[j] = s0.att_v[u + j]; qq[j] = pqv[u + j]; }
        // four frames per thread at a time: their d(keys) rows (read-modify-write, 2 x 16 bytes) are requested together
        constexpr int FB = 4;
        for (int tb = f0 + phase; tb < flen; tb += P * FB) {
          float4 d0[FB], d1[FB];
#pragma unroll
          for (int i = 0; i < FB; ++i) {
            const int tt = tb + i * P;
            if (tt < flen) {
              const float4* dk = reinterpret_cast<const float4*>(s0.dkeys_acc + ((int64_t)b * Tm + tt) * Hd + u);
              d0[i] = dk[0];
              d1[i] = dk[1];
            }
          }
#pragma unroll
          for (int i = 0; i < FB; ++i) {
            const int tt = tb + i * P;
            if (tt >= flen) continue;
            // (an if, and the LDS side through an address-space pointer: as `resident ? *lds : *global` the two met in a generic
            // pointer and the load was a FLAT one, which waits on both the LDS and the vector-memory counters; round 6)
            uint4 v;
            if (resident) v = ld16((lds_cu16)lkeys + (size_t)(tt - f0) * Hd + u);
            else v = ld16(keys + (int64_t)tt * Hd + u);
            const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
            const float d = dal[tt];
            float dkv[8] = {d0[i].x, d0[i].y, d0[i].z, d0[i].w, d1[i].x, d1[i].y, d1[i].z, d1[i].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float th = las_tanh(las_bf2f(e[j]) + qq[j]);
              dv[j] += d * th;
              const float pp = d * vv[j] * (1.f - th * th);
              a[j] += pp;
              dkv[j] += pp;
            }
            float4* dk = reinterpret_cast<float4*>(s0.dkeys_acc + ((int64_t)b * Tm + tt) * Hd + u);
            dk[0] = make_float4(dkv[0], dkv[1], dkv[2], dkv[3]);
            dk[1] = make_float4(dkv[4], dkv[5], dkv[6], dkv[7]);
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) dv_tot[j] += dv[j];
      }
      if ((Tm & 1) == 0) {                       // (dhs sits 2 Tm floats behind a 16-byte boundary)
        *reinterpret_cast<float4*>(dhs + phase * Hd + u) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(dhs + phase * Hd + u + 4) = make_float4(a[4], a[5], a[6], a[7]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) dhs[phase * Hd + u + j] = a[j];
      }
      lds_barrier();
      LAS_STAMPB(p.U - 1 - t, 7);
      // partial dh of this workgroup's frames (unit = threadIdx.x; Hd <= 256): parts 1..3 send theirs to part 0 as granules
#pragma unroll
      for (int q = 0; q < UPT; ++q) {
        dh_own[q] = 0.f;
        if (tid < Hd)
          for (int ph = 0; ph < P; ++ph) dh_own[q] += dhs[ph * Hd + tid + q * 256];
      }
      }
      pu64* xh = xdh + ((size_t)(xtag & 1) * B + b) * 4 * Hd;
      constexpr bool SPLITQ = WQ;            // query-layer attentions: the dh = d(pq) Wq^T product is split over the four parts
      const bool splitq = SPLITQ && att_uses_wq(s0.attention);
      // parts 1..3 send their partial sums to part 0 (slots 0..2); with the split product every part sends to every other
      if (part != 0 || splitq) {
#pragma unroll
        for (int q = 0; q < UPT; ++q)
          if (tid < Hd) pgranule_store(xh + (size_t)(splitq ? part : part - 1) * Hd + tid + q * 256, xtag, dh_own[q], local);
      }
      if (part == 0 || splitq) {
        // ---- S3: total dh (part 0; every part when the product is split) ----
        float tot[UPT];
#pragma unroll
        for (int q = 0; q < UPT; ++q) tot[q] = dh_own[q];
        unsigned spins = 0;
        for (;;) {                                  // wave-uniform polling, bounded
          bool got = true;
          pu64 g3[UPT][3];
#pragma unroll
          for (int q = 0; q < UPT; ++q)
#pragma unroll
            for (int w = 0; w < 3; ++w) {
              const int slot = splitq ? (w + (w >= part ? 1 : 0)) : w;          // the other three parts
              g3[q][w] = tid < Hd ? pgranule_load(xh + (size_t)slot * Hd + tid + q * 256) : ((pu64)xtag << 32);
              got = got && ((unsigned)(g3[q][w] >> 32) == xtag);
            }
          if (__all(got)) {
#pragma unroll
            for (int q = 0; q < UPT; ++q)
#pragma unroll
              for (int w = 0; w < 3; ++w) tot[q] += __uint_as_float((unsigned)g3[q][w]);
            break;
          }
          if (++spins > P_SPIN_LIMIT || *fail) { if (lane == 0) *fail = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        lds_barrier();                            // dhs (the per-phase partials) has been read by everybody
#pragma unroll
        for (int q = 0; q < UPT; ++q)
          if (tid < Hd) dhs[tid + q * 256] = tot[q];
      }
      if constexpr (SPLITQ) if (splitq) {
        // dhs holds d(processed query) in every part.  Part 0 saves it (bf16) for d(query_layer); each part maps ITS
        // quarter of the units back -- dh[u] = sum_a dpq[a] Wq[u][a] over a quarter of Wq^T's columns (a quarter of the
        // matrix from L2 instead of all of it on one workgroup: 11 of the 32 us of a 512-unit step) -- and parts 1..3 hand
        // their quarters to part 0 as granules
        lds_barrier();
        if (s0.attention == LAS_ATT_CUSTOM) {
          const float* pqv = s0.pq + (int64_t)b * s0.ldpq + (int64_t)t * p.inc_pq;
          for (int u = tid; u < Hd; u += 256) if (!(pqv[u] > 0.f)) dhs[u] = 0.f;
          lds_barrier();
        }
        if (part == 0)
          for (int u = tid; u < Hd; u += 256)
            if (s0.dpq_out) s0.dpq_out[(int64_t)b * s0.lddpq + (int64_t)t * p.inc_pq + u] = las_f2bf(dhs[u]);
        const int qc = Hd / 4, c0 = part * qc;
        const int AG = qc / 8, UG = 256 / AG;
        {
          const int ag = tid % AG, ug = tid / AG;
          float acc[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = 0.f;
          const unsigned short* wp = s0.wq_t + c0 + ag * 8;
          int r = ug;
          for (; r + 7 * UG < Hd; r += 8 * UG) {
            uint4 w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = ld16(wp + (int64_t)(r + i * UG) * Hd);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const unsigned short* e = reinterpret_cast<const unsigned short*>(&w[i]);
              const float xv = dhs[r + i * UG];
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[j] += xv * las_bf2f(e[j]);
            }
          }
          for (; r < Hd; r += UG) {
            const uint4 w = ld16(wp + (int64_t)r * Hd);
            const unsigned short* e = reinterpret_cast<const unsigned short*>(&w);
            const float xv = dhs[r];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += xv * las_bf2f(e[j]);
          }
          // (scratch: the G role's partial-tile area, idle between two products: at least 2048 floats)
#pragma unroll
          for (int j = 0; j < 8; ++j) red2[(ug * AG + ag) * 8 + j] = acc[j];
        }
        lds_barrier();
        pu64* xq = xdq + ((size_t)(xtag & 1) * B + b) * Hd;
        float* tmp = dhs + Hd;                      // (dhs: 2048 floats, Hd <= 512)
        for (int c = tid; c < qc; c += 256) {
          float v = 0.f;
          for (int g = 0; g < UG; ++g) v += red2[(g * AG + (c >> 3)) * 8 + (c & 7)];
          if (part == 0) tmp[c0 + c] = v;
          else pgranule_store(xq + c0 + c, xtag, v, local);
        }
        if (part == 0) {
          for (int a0 = 0; a0 < Hd; a0 += 256) {
            const int aa = a0 + tid;
            const bool need = aa < Hd && aa >= qc;
            unsigned spins = 0;
            for (;;) {
              const pu64 gq = need ? pgranule_load(xq + aa) : ((pu64)xtag << 32);
              if (__all((unsigned)(gq >> 32) == xtag)) {
                if (need) tmp[aa] = __uint_as_float((unsigned)gq);
                break;
              }
              if (++spins > P_SPIN_LIMIT || *fail) { if (lane == 0) *fail = 1; break; }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          lds_barrier();
          for (int u = tid; u < Hd; u += 256) dhs[u] = tmp[u];
        }
      }
    }
    if (active && part == 0) {
      lds_barrier();
      if (cellw) {
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
          const int u = tid + q * 256;
          const float gi = sg[q][0], gj = sg[q][1], gf = sg[q][2], go = sg[q][3], ct = sct[q], cp = scp[q];
          const float dht = dhs[u] + sdf[q];
          const float tc = las_tanh(ct);
          const float dov = dht * tc * go * (1.f - go);
          const float dct = dcr[q] + dht * go * (1.f - tc * tc);
          const float di = dct * gj * gi * (1.f - gi);
          const float dj = dct * gi * (1.f - gj * gj);
          const float df = dct * cp * gf * (1.f - gf);
          dcr[q] = dct * gf;
          unsigned short* zp = dzA + (int64_t)b * s0.ldz + (int64_t)t * p.inc_dz + u;
          zp[0] = las_f2bf(di); zp[Hd] = las_f2bf(dj); zp[2 * Hd] = las_f2bf(df); zp[3 * Hd] = las_f2bf(dov);
        }
      }
    }
    LAS_STAMPB(p.U - 1 - t, 8);
    if (!persist_barrier(flags, member, ++epoch, local, fail)) break;
    LAS_STAMPB(p.U - 1 - t, 9);
    if constexpr (TWO) {
      if (!w1) {
        // ---- wiring 0: P1_t -> cell 0 (plain; polls its columns of P1_t) -> barrier -> P0_t below ----
        product1(t, xtag);
        if (active && part == 0) plain_cell(t, xtag, first);
        if (!persist_barrier(flags, member, ++epoch, local, fail)) break;
      }
    }
    // ---- G: dfeed_t[group's utterances, my column tiles] = dz_t K^T ----
    {
      f32x4 acc[NT_MAX];
#pragma unroll
      for (int j = 0; j < NT_MAX; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // no branches around the loads or the products (a branch costs a vmcnt(0) at its join and the scheduler then
      // re-uses a few registers for the operand pieces: dependent round trips to L2): chunks past the end of K re-read the
      // last one against zero weights, absent tiles multiply zero weights, absent utterances' rows are dropped below,
      // and rows 8..15 of the MFMA tile (never read) all load one piece
      // (those lanes read 16 bytes of the workspace header: words 4..7 are never written, so they are zeros.  Round 3 noted
      //  "wrong gradients with uninitialised memory there" without finding a consumer that mixes rows.  Round 4 ran the
      //  experiment: NaN / Inf / max-finite / 1.0 patterns in exactly these 16 bytes, forward and backward workspace, 128 and
      //  256 units, Luong and Bahdanau, one and two groups -- logits and gradients bit-identical to the clean run in all 64
      //  combinations (tests/test_gpu_model.py::test_padding_rows_of_the_one_launch_decoders_do_not_leak keeps it so).  The
      //  rows do not leak; what round 3 saw was not this read.  Zeros stay: one cache line every lane group shares.)
      const unsigned short* zrow = (l15 < 8) ? s0.dz + (int64_t)min(bg, B - 1) * s0.ldz + (int64_t)t * p.inc_dz + 8 * lq
                                             : reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(p.workspace) + 16);
      const int zstep = (l15 < 8) ? 32 : 0;
      uint4 av[KRES];
#pragma unroll
      for (int i = 0; i < KRES; ++i) av[i] = *reinterpret_cast<const uint4*>(zrow + min(wave + 4 * i, KC - 1) * zstep);
      constexpr int NS = KCW_MAX - KRES, SD = 3;           // streamed chunks: operand piece and NT_MAX weight pieces each
      uint4 sa[NS > 0 ? SD : 1], sw[NS > 0 ? SD : 1][NT_MAX];
      auto stream_issue = [&](int slot, int i) {
        const int kcc = min(wave + 4 * i, KC - 1);
        sa[slot] = *reinterpret_cast<const uint4*>(zrow + kcc * zstep);
#pragma unroll
        for (int j = 0; j < NT_MAX; ++j) sw[slot][j] = *reinterpret_cast<const uint4*>(wrow[j] + kcc * 32);
      };
      if constexpr (NS > 0) {
#pragma unroll
        for (int q = 0; q < SD; ++q) stream_issue(q, KRES + q);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < KRES; ++i)
#pragma unroll
        for (int j = 0; j < NT_MAX; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[i]), wf[j][i], acc[j], 0, 0, 0);
      if constexpr (NS > 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int slot = i % SD;
          const bool in = wave + 4 * (KRES + i) < KC;
          const uint4 a4 = in ? sa[slot] : make_uint4(0, 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NT_MAX; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, sw[slot][j]), acc[j], 0, 0, 0);
          if (i + SD < NS) stream_issue(slot, KRES + i + SD);
        }
      }
      if (lq < 2) {
#pragma unroll
        for (int j = 0; j < NT_MAX; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) red2[(wave * 8 + lq * 4 + r) * RS2 + j * 16 + l15] = acc[j][r];
      }
      LAS_STAMPB(p.U - 1 - t, 10);
      lds_barrier();
      float* df = p.dfeed_all + (int64_t)t * B * W;
      for (int e = tid; e < 8 * NT_MAX * 16; e += 256) {
        const int row = e / (NT_MAX * 16), c = e % (NT_MAX * 16), j = c / 16, col = c % 16;
        const int tile = member + 32 * j, bb = group * 8 + row;
        if (tile < NTW && bb < B) {
          const float v = red2[(0 * 8 + row) * RS2 + c] + red2[(1 * 8 + row) * RS2 + c] +
                          red2[(2 * 8 + row) * RS2 + c] + red2[(3 * 8 + row) * RS2 + c];
          if (t == 0) df[(int64_t)bb * W + tile * 16 + col] = v;        // the caller's d(initial attention, h)
          else pgranule_store(xdf + ((size_t)(xtag & 1) * B + bb) * W + tile * 16 + col, xtag, v, local);
        }
      }
    }
    // (no group barrier: step t-1 polls the granules of its utterance)
#pragma unroll
    for (int i = 0; i < PD; ++i) cur_dc[i] = nxt_dc[i];
    cur_al = nxt_al;
  }
#pragma unroll
  for (int q = 0; q < UPT; ++q)
    if (cellw) dcA[(int64_t)b * Hd + tid + q * 256] = dcr[q];            // d(c) before the first step: the caller's d(initial state)
  if (TWO && cellw) dcP[(int64_t)b * Hd + tid] = dcp;
  if (WQ && att_additive(s0.attention)) {
    const int L8 = Hd / 8, P8 = 256 / L8, u = (tid % L8) * 8, phs = tid / L8;
    if (p.sum_workspace) {
      // the frame phases of a column meet in LDS, the workgroups in workgroup order (ordered_accumulate); absent utterances add zeros
      __syncthreads();
      float* scr = sm;                                   // [P8][Hd] + [Hd] + a flag word: the step arrays are free now
#pragma unroll
      for (int j = 0; j < 8; ++j) scr[phs * Hd + u + j] = active ? dv_tot[j] : 0.f;
      __syncthreads();
      for (int c = tid; c < Hd; c += 256) {
        float a = 0.f;
        for (int ph = 0; ph < P8; ++ph) a += scr[ph * Hd + c];
        scr[P8 * Hd + c] = a;
      }
      __syncthreads();
      ordered_accumulate(static_cast<unsigned*>(p.sum_workspace), blockIdx.x, gridDim.x, Hd, scr + P8 * Hd, s0.dv_acc, Hd, nullptr,
                         reinterpret_cast<int*>(scr + P8 * Hd + Hd));
    } else if (active) {
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(s0.dv_acc + u + j, dv_tot[j]);
    }
  }
  if (*fail && tid == 0) atomicOr(status, 16u);
}

// ------------------------------------------------------------------------------------------------
// sequence cross-entropy (model_helper.py:24-30 -> tf.contrib.seq2seq.sequence_loss)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seq_ce_kernel(const float* logits, int64_t ldl, const int32_t* targets,
                                                     const int32_t* target_len, int B, int U, int V, float grad_scale,
                                                     float* loss_out, unsigned short* dlogits, int64_t ldd) {
  // one wave per (b, t) row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float total = 0.f;
  for (int i = lane; i < B; i += 64) total += (float)min(target_len[i], U);
  total = las_wave_sum(total) + 1e-12f;
  const float inv_total = 1.0f / total;
  float local_loss = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < B * U; row += gridDim.x * 4) {
    const int b = row / U, t = row % U;
    const bool on = t < target_len[b];
    const float* lg = logits + (int64_t)row * ldl;
    unsigned short* dl = dlogits ? dlogits + (int64_t)row * ldd : nullptr;
    if (!on) {
      if (dl) for (int v = lane; v < V; v += 64) dl[v] = 0;
      continue;
    }
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, lg[v]);
    mx = las_wave_max(mx);
    float sum = 0.f;
    for (int v = lane; v < V; v += 64) sum += __expf(lg[v] - mx);
    sum = las_wave_sum(sum);
    const float lse = mx + __logf(sum);
    const int tgt = targets[row];
    if (lane == 0) local_loss += (lse - lg[tgt]) * inv_total;
    if (dl) {
      const float sc = grad_scale * inv_total;
      for (int v = lane; v < V; v += 64) {
        const float p = __expf(lg[v] - lse);
        dl[v] = las_f2bf((p - (v == tgt ? 1.f : 0.f)) * sc);
      }
    }
  }
  // one atomic per workgroup: thousands of same-address atomics cost more than the rows themselves
  __shared__ float wl[4];
  if (lane == 0) wl[wave] = local_loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = wl[0] + wl[1] + wl[2] + wl[3];
    if (v != 0.f) atomicAdd(loss_out, v);
  }
}

// ------------------------------------------------------------------------------------------------
// sequence_loss_sigmoid (model_helper.py:81-95) as compute_loss_sigmoid uses it (:98-130): per decoder step the MEAN over
// the nf features of sigmoid_cross_entropy_with_logits(labels, logits) = max(x, 0) - x z + log(1 + exp(-|x|)), weighted
// by the sequence mask, divided by (sum of the weights + 1e-12).  targets: bf16 0/1 rows [B*U, ldt].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seq_sigmoid_kernel(const float* logits, int64_t ldl, const unsigned short* targets,
                                                          int64_t ldt, const int32_t* seq_len, int B, int U, int nf,
                                                          float grad_scale, float* loss_out, unsigned short* dlogits, int64_t ldd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float total = 0.f;
  for (int i = lane; i < B; i += 64) total += (float)min(seq_len[i], U);
  total = las_wave_sum(total) + 1e-12f;
  const float inv_total = 1.0f / total, inv_nf = 1.0f / (float)nf;
  float local_loss = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < B * U; row += gridDim.x * 4) {
    const int b = row / U, t = row % U;
    const bool on = t < seq_len[b];
    const float* lg = logits + (int64_t)row * ldl;
    unsigned short* dl = dlogits ? dlogits + (int64_t)row * ldd : nullptr;
    if (!on) {
      if (dl) for (int f = lane; f < nf; f += 64) dl[f] = 0;
      continue;
    }
    const unsigned short* tg = targets + (int64_t)row * ldt;
    float ce = 0.f;
    for (int f = lane; f < nf; f += 64) {
      const float x = lg[f], z = las_bf2f(tg[f]);
      ce += fmaxf(x, 0.f) - x * z + log1pf(__expf(-fabsf(x)));
      if (dl) dl[f] = las_f2bf((las_sigmoid(x) - z) * inv_nf * inv_total * grad_scale);
    }
    ce = las_wave_sum(ce);
    if (lane == 0) local_loss += ce * inv_nf * inv_total;
  }
  __shared__ float wl[4];
  if (lane == 0) wl[wave] = local_loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = wl[0] + wl[1] + wl[2] + wl[3];
    if (v != 0.f) atomicAdd(loss_out, v);
  }
}

// ScheduledSigmoidHelper (utils/training_helper.py:89-119, binf_to_ipa None): with probability prob per utterance the next
// decoder input is a Bernoulli(sigmoid(logits)) draw per feature, else the teacher's feature vector.  One thread per (b, f).
__global__ void sample_features_kernel(const float* logits, int64_t ldl, int nf, const unsigned short* teacher, int64_t ldt,
                                       unsigned short* next, int64_t ldn, int B, float prob, unsigned seed, unsigned step) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nf) return;
  const int b = i / nf, f = i % nf;
  const bool select = las_uniform(seed, 0x5e1ec7u, (unsigned long long)step * B + b) < prob;
  unsigned short out = teacher ? teacher[(int64_t)b * ldt + f] : 0;
  if (select) {
    const float u = las_uniform(seed, 0xb17f00du, ((unsigned long long)step * B + b) * nf + f);
    out = u < las_sigmoid(logits[(int64_t)b * ldl + f]) ? 0x3F80u : 0u;       // bf16 1.0 / 0.0
  }
  next[(int64_t)b * ldn + f] = out;
}

// ------------------------------------------------------------------------------------------------
// compute_log_probs_loss (model_helper.py:132-146): mean over all rows x nf of
//   |e^a + e^b - 1| + relu(a) + relu(b),  a = x[:, f] (log p(feature = 1)), b = x[:, nf + f]
// (the reference multiplies and divides by a gradient-free constant for stability; e^a + e^b is the same value)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void log_probs_loss_kernel(const unsigned short* x, int64_t ldx, int rows, int nf, float weight,
                                                             float grad_scale, float* loss_out, float* dx, int64_t ldd) {
  const int64_t total = (int64_t)rows * nf;
  const float inv = 1.0f / (float)total;
  float local = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / nf), f = (int)(i % nf);
    const float a = las_bf2f(x[(int64_t)r * ldx + f]), b = las_bf2f(x[(int64_t)r * ldx + nf + f]);
    const float ea = __expf(a), eb = __expf(b);
    const float v = ea + eb - 1.f;
    local += (fabsf(v) + fmaxf(a, 0.f) + fmaxf(b, 0.f)) * inv;
    if (dx) {
      const float sg = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
      const float k = weight * grad_scale * inv;
      dx[(int64_t)r * ldd + f] = k * (sg * ea + (a > 0.f ? 1.f : 0.f));
      dx[(int64_t)r * ldd + nf + f] = k * (sg * eb + (b > 0.f ? 1.f : 0.f));
    }
  }
  local = las_wave_sum(local);
  if ((threadIdx.x & 63) == 0 && local != 0.f) atomicAdd(loss_out, local * weight);
}

// ------------------------------------------------------------------------------------------------
// one step of tf.contrib.seq2seq.BeamSearchDecoder (las/model.py:312-319; length_penalty_weight 0): log-softmax per
// beam, finished beams keep their mass on end_token (_mask_probs), total = previous log-prob + step log-prob, top-K
// of the K*V candidates of an utterance (ties: lower index first, as tf.nn.top_k).  One workgroup per utterance.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void beam_step_kernel(const float* logits, int64_t ldl, float* log_probs, int32_t* finished,
                                                        int32_t* lengths, int32_t* word_ids, int32_t* parent_ids, int K, int V,
                                                        int eos) {
  extern __shared__ __attribute__((aligned(16))) float bsm[];
  float* cand = bsm;                                   // [K*V]
  float* lse = cand + K * V;                           // [K]
  float* oldlp = lse + K;                              // [K]
  int* oldfin = reinterpret_cast<int*>(oldlp + K);     // [K]
  int* oldlen = oldfin + K;                            // [K]
  float* rv = reinterpret_cast<float*>(oldlen + K);    // [4] wave maxima
  int* ri = reinterpret_cast<int*>(rv + 4);            // [4] their indices
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float NEG = -3.402823466e38f;
  for (int k = wave; k < K; k += 4) {
    const float* lg = logits + (int64_t)(b * K + k) * ldl;
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, lg[v]);
    mx = las_wave_max(mx);
    float sum = 0.f;
    for (int v = lane; v < V; v += 64) sum += __expf(lg[v] - mx);
    sum = las_wave_sum(sum);
    if (lane == 0) {
      lse[k] = mx + __logf(sum);
      oldlp[k] = log_probs[b * K + k];
      oldfin[k] = finished[b * K + k];
      oldlen[k] = lengths[b * K + k];
    }
  }
  __syncthreads();
  for (int i = tid; i < K * V; i += 256) {
    const int k = i / V, v = i % V;
    const float step = oldfin[k] ? (v == eos ? 0.f : NEG) : logits[(int64_t)(b * K + k) * ldl + v] - lse[k];
    cand[i] = oldlp[k] + step;
  }
  __syncthreads();
  for (int j = 0; j < K; ++j) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < K * V; i += 256) {
      const float c = cand[i];
      if (!(c != c) && (bi == 0x7fffffff || c > best)) { best = c; bi = i; }      // first (lowest) index wins ties; NaN = taken
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
    }
    if (lane == 0) { rv[wave] = best; ri[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (ri[w] != 0x7fffffff && (bi == 0x7fffffff || rv[w] > best || (rv[w] == best && ri[w] < bi))) { best = rv[w]; bi = ri[w]; }
      const int k = bi / V, v = bi % V;
      const int pf = oldfin[k];
      word_ids[b * K + j] = v;
      parent_ids[b * K + j] = k;
      log_probs[b * K + j] = best;
      finished[b * K + j] = (pf || v == eos) ? 1 : 0;
      lengths[b * K + j] = oldlen[k] + (pf ? 0 : 1);
      cand[bi] = __builtin_nanf("");          // taken
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int las_beam_step(const float* logits, int64_t ldl, float* log_probs, int32_t* finished, int32_t* lengths,
                             int32_t* word_ids, int32_t* parent_ids, int B, int K, int V, int eos, void* stream) {
  LAS_REQUIRE(logits && log_probs && finished && lengths && word_ids && parent_ids, "las_beam_step: null argument");
  LAS_REQUIRE(B > 0 && K > 0 && V > 0 && eos >= 0 && eos < V, "las_beam_step: bad shape B=%d K=%d V=%d eos=%d", B, K, V, eos);
  const size_t lds = ((size_t)K * V + 4 * (size_t)K + 8) * sizeof(float);
  LAS_REQUIRE(lds <= 64 * 1024 && K * V >= K, "las_beam_step: beam_width * vocabulary = %d exceeds the LDS candidate buffer", K * V);
  hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, logits, ldl, log_probs, finished, lengths,
                     word_ids, parent_ids, K, V, eos);
  LAS_LAUNCH_CHECK("beam step launch");
  return LAS_OK;
}

extern "C" int las_log_probs_loss(const las_bf16* x, int64_t ldx, int rows, int nf, float weight, float grad_scale,
                                  float* loss_out, float* dx, int64_t ldd, void* stream) {
  LAS_REQUIRE(x && loss_out && rows > 0 && nf > 0, "las_log_probs_loss: bad arguments");
  int blocks = (int)(((int64_t)rows * nf + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(log_probs_loss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, nf, weight, grad_scale,
                     loss_out, dx, ldd);
  LAS_LAUNCH_CHECK("log-probs loss launch");
  return LAS_OK;
}

extern "C" int las_decoder_step_fwd(const las_dec_step* s, int parts, void* stream) {
  LAS_REQUIRE(s->B > 0 && s->Hd % 8 == 0 && s->M % 8 == 0 && s->Tm > 0, "las_decoder_step_fwd: bad shape");
  const int att = s->attention;
  const bool additive = att == LAS_ATT_BAHDANAU || att == LAS_ATT_BAHDANAU_MONOTONIC;
  LAS_REQUIRE(att >= LAS_ATT_LUONG && att <= LAS_ATT_BAHDANAU_MONOTONIC, "las_decoder_step_fwd: unknown attention %d", att);
  LAS_REQUIRE(s->mode == LAS_DEC_CELL_ONLY || !(additive || att == LAS_ATT_CUSTOM) || s->wq,
              "las_decoder_step_fwd: this attention type needs the query_layer kernel wq");
  LAS_REQUIRE(s->mode == LAS_DEC_CELL_ONLY || !additive || s->att_v, "las_decoder_step_fwd: Bahdanau scores need att_v");
  LAS_REQUIRE(s->norm >= LAS_NORM_SOFTMAX && s->norm <= LAS_NORM_MONOTONIC_HARD, "las_decoder_step_fwd: unknown normaliser %d", s->norm);
  LAS_REQUIRE(s->mode != LAS_DEC_ATTENTION_ONLY || s->query, "las_decoder_step_fwd: attention-only mode needs a query");
  if (s->mode == LAS_DEC_CELL_ONLY) parts = 1;
  if (parts < 1) parts = 1;
  const size_t lds = (size_t)(2 * s->Hd + s->Tm + 8 + 8 + 2048 + (s->norm != LAS_NORM_SOFTMAX ? 2 * s->Tm : 0)) * sizeof(float);
  LAS_REQUIRE(lds <= 64 * 1024, "las_decoder_step_fwd: memory length %d too long for the LDS score buffer", s->Tm);
  hipLaunchKernelGGL(dec_step_fwd_kernel, dim3(s->B, parts), dim3(256), lds, (hipStream_t)stream, *s);
  LAS_LAUNCH_CHECK("decoder step fwd launch");
  return LAS_OK;
}

extern "C" int las_decoder_persist_al_supported(int Hd, int M, int K_in, int A, int attention, int norm) {
  // the one-launch forward with an attention layer and / or a monotonic normaliser ('parallel' mode: TRAIN): general body
  if (norm != LAS_NORM_SOFTMAX && norm != LAS_NORM_MONOTONIC_PARALLEL) return 0;
  if (attention < LAS_ATT_LUONG || attention > LAS_ATT_BAHDANAU_MONOTONIC) return 0;
  if ((norm == LAS_NORM_SOFTMAX) != (attention == LAS_ATT_LUONG || attention == LAS_ATT_BAHDANAU || attention == LAS_ATT_CUSTOM)) return 0;
  if (Hd != 128 && Hd != 256) return 0;
  if (K_in % 64 != 0 || K_in / 32 > 48 || M % 32 != 0) return 0;
  if (A < 0 || A % 16 != 0 || A / 16 > P_MEMBERS || (A > 0 && (Hd + M) / 32 > 40)) return 0;      // <= 10 K chunks of W_al per wave
  return 1;
}

extern "C" int las_decoder_persist2_supported(int Hd, int M, int K_in, int K1_in, int attention, int wiring) {
  // the one-launch forward with a second cell (general body): decoder_units 128 / 256, softmax attentions
  if (attention != LAS_ATT_LUONG && attention != LAS_ATT_BAHDANAU) return 0;
  if (Hd != 128 && Hd != 256) return 0;
  if (K_in % 64 != 0 || K_in / 32 > 48 || M % 32 != 0) return 0;
  if (wiring == 0) return K1_in == 2 * Hd ? 1 : 0;
  if (wiring == 1) return K1_in == 2 * M + Hd ? 1 : 0;
  return 0;
}

extern "C" int las_decoder_persist_supported(int Hd, int M, int K_in, int attention, int norm) {
  if (norm != LAS_NORM_SOFTMAX) return 0;
  if (attention != LAS_ATT_LUONG && attention != LAS_ATT_BAHDANAU && attention != LAS_ATT_CUSTOM) return 0;
  if (Hd != 128 && Hd != 256 && Hd != 512) return 0;      // 4Hd/32 columns per member: 16, 32 or 64
  if (K_in % 64 != 0 || K_in / 32 > (Hd == 512 ? 80 : 48)) return 0;   // register-resident K slice (<= 12 / 20 chunks per wave);
                                                           // operand rows of whole 128-byte lines (no line shared by two steps)
  if (M % 32 != 0) return 0;
  return 1;
}

// Every workgroup of a persistent launch needs a CU of its own (its registers and LDS fill one) and the 32 of a group must
// be resident together: 8 utterances per 32 CUs, 64 at once on a 256-CU MI355X.  Blocks are laid out in chunks of 8 groups,
// so a larger batch (up to four chunks per launch) runs chunk after chunk as CUs become free.
static int persist_max_batch() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return (cus / (8 * P_MEMBERS)) * 8 * 8;
}
extern "C" int las_decoder_persist_max_batch(void) { return persist_max_batch(); }

extern "C" size_t las_decoder_persist_workspace_bytes(int B, int Tm, int Hd, int M) {
  return 64 + (persist_flag_words(B) + persist_exchange_words(B, Tm, Hd, M)) * sizeof(pu64);
}

extern "C" int las_decoder_persist_fwd(const las_dec_persist* p, void* stream) {
  const las_dec_step* s = &p->s;
  LAS_REQUIRE(s->B > 0 && p->U > 0 && s->mode == LAS_DEC_FUSED, "las_decoder_persist_fwd: bad shape / mode");
  LAS_REQUIRE(s->B <= 4 * persist_max_batch(), "las_decoder_persist_fwd: at most %d utterances per launch (got %d)", 4 * persist_max_batch(), s->B);
  const bool two = p->k1T != nullptr;
  const bool al_path = p->walT != nullptr || s->norm != LAS_NORM_SOFTMAX || two;
  LAS_REQUIRE(!two || (las_decoder_persist2_supported(s->Hd, s->M, p->K_in, p->K1_in, s->attention, p->wiring) && !p->walT &&
                       s->norm == LAS_NORM_SOFTMAX && p->bias1 && p->c1 && p->gates1 && p->h1 && p->ldk1 >= p->K1_in && p->ldk1 % 8 == 0),
              "las_decoder_persist_fwd: second cell: unsupported configuration or missing buffers (Hd=%d M=%d K_in=%d K1_in=%d wiring=%d)",
              s->Hd, s->M, p->K_in, p->K1_in, p->wiring);
  LAS_REQUIRE(two || (al_path ? las_decoder_persist_al_supported(s->Hd, s->M, p->K_in, p->walT ? p->A : 0, s->attention, s->norm)
                              : las_decoder_persist_supported(s->Hd, s->M, p->K_in, s->attention, s->norm)),
              "las_decoder_persist_fwd: configuration not supported (Hd=%d M=%d K_in=%d attention=%d norm=%d)", s->Hd, s->M, p->K_in,
              s->attention, s->norm);
  LAS_REQUIRE(!al_path || two || (p->sampling_prob <= 0.f && s->drop_keep >= 1.0f), "las_decoder_persist_fwd: attention layer / monotonic "
              "normalisers inside the launch: without scheduled sampling and input dropout");
  LAS_REQUIRE(!two || s->drop_keep >= 1.0f || (p->win0 >= s->M && p->win0 % 8 == 0 && p->win0 <= p->K_in && p->win1 > 0 && p->win1 % 8 == 0 && p->win1 <= p->K1_in),
              "las_decoder_persist_fwd: second cell with input dropout: win0 / win1 (masked columns of the two operand rows, multiples of 8)");
  LAS_REQUIRE(!two || p->sampling_prob <= 0.f || (p->wiring == 0 ? p->ldw >= s->M : (p->ldw >= s->Hd && s->Hd % 32 == 0)),
              "las_decoder_persist_fwd: second cell with scheduled sampling: projection rows of the output width");
  LAS_REQUIRE(!p->walT || (p->att_out && p->ld_wal >= s->Hd + s->M && p->x_att_off >= 0 && p->x_att_off + p->A <= p->K_in && !s->ctx_out2),
              "las_decoder_persist_fwd: attention layer needs att_out, a W_al^T of Hd + M columns, a place in the operand row (and no "
              "context copy there)");
  LAS_REQUIRE(s->norm == LAS_NORM_SOFTMAX || (s->p_out && s->align_out), "las_decoder_persist_fwd: monotonic normalisers save p_choose");
  LAS_REQUIRE(p->x && p->kT && p->workspace && ((uintptr_t)p->workspace % 128 == 0) && ((uintptr_t)p->x % 128 == 0),
              "las_decoder_persist_fwd: null argument, or exchanged rows that are not whole 128-byte lines");
  LAS_REQUIRE(p->sampling_prob <= 0.f || (p->wprojT && p->bproj && p->plog && p->V > 0 && p->Vp >= p->V &&
                                          p->Vp <= 1024 && s->M % 32 == 0 && p->inc_tok == 1),
              "las_decoder_persist_fwd: scheduled sampling needs wprojT, bproj, plog (and fed ids with unit step)");
  hipStream_t st = (hipStream_t)stream;
  const int groups = (s->B + 7) / 8;
  // flags and granules start from zero; the 64-byte status header is sticky (cleared by the host after it has read it)
  int rc = las_check_hip(hipMemsetAsync(reinterpret_cast<char*>(p->workspace) + 64, 0,
                                        las_decoder_persist_workspace_bytes(s->B, s->Tm, s->Hd, s->M) - 64, st), "memset workspace");
  if (rc) return rc;
  size_t lds = persist_fwd_scratch_floats(s->Hd, s->Tm) * sizeof(float);
  LAS_REQUIRE(lds <= 64 * 1024, "las_decoder_persist_fwd: memory length %d too long for the LDS score buffer", s->Tm);
  const bool res = persist_fwd_resident(s->M, s->Hd, s->Tm);
  const dim3 grid(((groups + 7) & ~7) * P_MEMBERS);
  if (al_path) {
    // attention layer and / or monotonic normaliser: the general body with the A role (AL) or without
    // (long memories: keys + as many value frames as fit stay in LDS, the rest of the values is streamed)
    const int vres = persist_fwd_resident_frames(s->M, s->Hd, s->Tm);
    const bool res = vres > 0;
    const size_t lds_al = lds + (res ? persist_fwd_resident_bytes_partial(s->M, s->Hd, s->Tm, vres) : 0);
#define LAS_AL_LAUNCH(...)                                                                                                      \
  do {                                                                                                                          \
    static bool attr = false;                                                                                                   \
    if (!attr) {                                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_persist_fwd_kernel<__VA_ARGS__>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                        \
      attr = true;                                                                                                              \
    }                                                                                                                           \
    hipLaunchKernelGGL((dec_persist_fwd_kernel<__VA_ARGS__>), grid, dim3(256), lds_al, st, *p);                                 \
  } while (0)
    if (two && p->sampling_prob > 0.f) {
      if (res) LAS_AL_LAUNCH(true, true, 2, 12, 12, false, 10, true);
      else LAS_AL_LAUNCH(true, false, 2, 12, 12, false, 10, true);
    } else if (two) {
      if (res) LAS_AL_LAUNCH(false, true, 2, 12, 12, false, 10, true);
      else LAS_AL_LAUNCH(false, false, 2, 12, 12, false, 10, true);
    } else if (p->walT) {
      if (res) LAS_AL_LAUNCH(false, true, 2, 12, 12, true, 10);
      else LAS_AL_LAUNCH(false, false, 2, 12, 12, true, 10);
    } else {
      if (res) LAS_AL_LAUNCH(false, true, 2, 12, 12, false, 1);
      else LAS_AL_LAUNCH(false, false, 2, 12, 12, false, 1);
    }
#undef LAS_AL_LAUNCH
    LAS_LAUNCH_CHECK("persistent decoder fwd (attention layer / monotonic) launch");
    return LAS_OK;
  }
  const int lean_mode = las_knob("LAS_DEC_LEAN", 1);     // 0: the general body also where the written-out one applies (diagnostics)
  if (lean_mode && (p->sampling_prob <= 0.f || s->Hd <= 256) && s->tok_rows &&
      persist_fwd_lean_ok(s->Hd, s->M, s->Tm, p->U, s->attention, s->norm) && p->K_in / 32 <= (s->Hd == 512 ? 80 : 48)) {
    const size_t lbytes = lean_layout(s->Hd, s->Tm, s->M, p->U).total_bytes;
#define LAS_LEAN_LAUNCH(...)                                                                                                    \
  do {                                                                                                                          \
    static bool attr = false;                                                                                                   \
    if (!attr) {                                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_persist_fwd_lean_kernel<__VA_ARGS__>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                        \
      attr = true;                                                                                                              \
    }                                                                                                                           \
    hipLaunchKernelGGL((dec_persist_fwd_lean_kernel<__VA_ARGS__>), grid, dim3(256), lbytes, st, *p);                            \
  } while (0)
    // 4 Hd / 32 columns per member = one, two or four 16-column tiles; 512 units: 8 of the 20 K chunks per wave resident
    if (p->sampling_prob > 0.f) {           // (scheduled sampling inside the launch: decoder_units <= 256)
      if (s->attention == LAS_ATT_LUONG) {
        if (s->Hd == 256) LAS_LEAN_LAUNCH(LAS_ATT_LUONG, 2, 12, 12, true);
        else LAS_LEAN_LAUNCH(LAS_ATT_LUONG, 1, 12, 12, true);
      } else {
        if (s->Hd == 256) LAS_LEAN_LAUNCH(LAS_ATT_BAHDANAU, 2, 12, 12, true);
        else LAS_LEAN_LAUNCH(LAS_ATT_BAHDANAU, 1, 12, 12, true);
      }
    } else if (s->attention == LAS_ATT_LUONG) {
      if (s->Hd == 512) LAS_LEAN_LAUNCH(LAS_ATT_LUONG, 4, 8, 20);
      else if (s->Hd == 256) LAS_LEAN_LAUNCH(LAS_ATT_LUONG, 2);
      else LAS_LEAN_LAUNCH(LAS_ATT_LUONG, 1);
    } else {
      if (s->Hd == 512) LAS_LEAN_LAUNCH(LAS_ATT_BAHDANAU, 4, 8, 20);
      else if (s->Hd == 256) LAS_LEAN_LAUNCH(LAS_ATT_BAHDANAU, 2);
      else LAS_LEAN_LAUNCH(LAS_ATT_BAHDANAU, 1);
    }
#undef LAS_LEAN_LAUNCH
    LAS_LAUNCH_CHECK("persistent decoder fwd launch");
    return LAS_OK;
  }
  if (s->Hd == 512) {          // the wide flavour: 4 column tiles x 20 K chunks of the cell kernel per wave in registers
    LAS_REQUIRE(p->sampling_prob <= 0.f, "las_decoder_persist_fwd: scheduled sampling inside the launch is built for decoder_units <= 256");
    static bool attr512 = false;
    if (!attr512) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_persist_fwd_kernel<false, true, 4, 20, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_persist_fwd_kernel<false, false, 4, 20, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr512 = true;
    }
    if (res) {
      lds += persist_fwd_resident_bytes(s->M, s->Hd, s->Tm);
      hipLaunchKernelGGL((dec_persist_fwd_kernel<false, true, 4, 20, 8>), grid, dim3(256), lds, st, *p);
    } else {
      hipLaunchKernelGGL((dec_persist_fwd_kernel<false, false, 4, 20, 8>), grid, dim3(256), lds, st, *p);
    }
    LAS_LAUNCH_CHECK("persistent decoder fwd launch");
    return LAS_OK;
  }
  if (res) {
    lds += persist_fwd_resident_bytes(s->M, s->Hd, s->Tm);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_persist_fwd_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_persist_fwd_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    if (p->sampling_prob > 0.f) hipLaunchKernelGGL((dec_persist_fwd_kernel<true, true>), grid, dim3(256), lds, st, *p);
    else hipLaunchKernelGGL((dec_persist_fwd_kernel<false, true>), grid, dim3(256), lds, st, *p);
  } else {
    if (p->sampling_prob > 0.f) hipLaunchKernelGGL((dec_persist_fwd_kernel<true, false>), grid, dim3(256), lds, st, *p);
    else hipLaunchKernelGGL((dec_persist_fwd_kernel<false, false>), grid, dim3(256), lds, st, *p);
  }
  LAS_LAUNCH_CHECK("persistent decoder fwd launch");
  return LAS_OK;
}

extern "C" int las_decoder_persist_bwd_supported(int Hd, int M, int W, int attention, int norm) {
  if (norm != LAS_NORM_SOFTMAX) return 0;
  if (attention != LAS_ATT_LUONG && attention != LAS_ATT_BAHDANAU && attention != LAS_ATT_CUSTOM) return 0;
  if (Hd != 128 && Hd != 256 && Hd != 512) return 0;       // 4Hd/32 K chunks: <= 8 per wave (512 units: 16, ten of them streamed)
  if (W % 32 != 0 || M % 128 != 0) return 0;               // dfeed rows of whole lines
  if (Hd == 512) return (W / 16 <= 160 && M <= 2048) ? 1 : 0;   // <= 5 column tiles per member, 8 d(context) columns per thread
  if (W / 16 > 96 || M > 1536) return 0;                   // <= 3 column tiles per member, 6 d(context) columns per thread
  return 1;
}

extern "C" int las_decoder_persist2_bwd_supported(int Hd, int M, int W, int W1, int attention, int wiring) {
  // the one-launch backward with a second cell: decoder_units 128 / 256 (one unit per thread), softmax attentions
  if (attention != LAS_ATT_LUONG && attention != LAS_ATT_BAHDANAU) return 0;
  if (Hd != 128 && Hd != 256) return 0;
  if (!las_decoder_persist_bwd_supported(Hd, M, W, attention, LAS_NORM_SOFTMAX)) return 0;
  if (wiring == 0) return W1 == 2 * Hd ? 1 : 0;
  if (wiring == 1) return (W1 == 2 * M + Hd && W1 / 16 <= 160) ? 1 : 0;
  return 0;
}

extern "C" int las_decoder_persist_bwd(const las_dec_persist_bwd* p, void* stream) {
  const las_dec_step_bwd* s = &p->s;
  LAS_REQUIRE(s->B > 0 && p->U > 0, "las_decoder_persist_bwd: bad shape");
  LAS_REQUIRE(s->B <= 4 * persist_max_batch(), "las_decoder_persist_bwd: at most %d utterances per launch (got %d)", 4 * persist_max_batch(), s->B);
  LAS_REQUIRE(las_decoder_persist_bwd_supported(s->Hd, s->M, p->W, s->attention, s->norm),
              "las_decoder_persist_bwd: configuration not supported (Hd=%d M=%d W=%d attention=%d norm=%d)", s->Hd, s->M, p->W,
              s->attention, s->norm);
  const bool two = p->k1c != nullptr;
  LAS_REQUIRE(!two || (las_decoder_persist2_bwd_supported(s->Hd, s->M, p->W, p->W1, s->attention, p->wiring) && p->gates1 && p->c1 && p->dz1 &&
                       p->dc1 && p->ldk1 >= 4 * s->Hd && p->ldk1 % 8 == 0 && (p->wiring == 0 ? s->dctx_a != nullptr : (p->d_out1 != nullptr && !s->dctx_a))),
              "las_decoder_persist_bwd: second cell: unsupported configuration or missing buffers (Hd=%d M=%d W=%d W1=%d wiring=%d)", s->Hd, s->M,
              p->W, p->W1, p->wiring);
  LAS_REQUIRE(!two || s->drop_keep >= 1.0f || (p->win0 >= s->M && p->win1 > 0 && p->win1 <= p->W1),
              "las_decoder_persist_bwd: second cell with input dropout: win0 (>= M) / win1");
  LAS_REQUIRE(two || s->drop_keep >= 1.0f || s->feed_width >= s->M, "las_decoder_persist_bwd: feed_width");
  LAS_REQUIRE(p->kc && p->dfeed_all && p->workspace && ((uintptr_t)p->workspace % 128 == 0) && (s->dctx_a || two) && s->dc && s->dz && s->ds_out &&
                  s->align && s->gates && s->c_new && s->c_prev && s->keys && s->values && s->mem_len,
              "las_decoder_persist_bwd: null argument");
  LAS_REQUIRE(s->attention == LAS_ATT_LUONG || (s->pq && s->wq_t), "las_decoder_persist_bwd: this attention needs pq and wq_t");
  LAS_REQUIRE(s->attention != LAS_ATT_BAHDANAU || (s->att_v && s->dkeys_acc && s->dv_acc),
              "las_decoder_persist_bwd: Bahdanau scores need att_v, dkeys_acc, dv_acc");
  hipStream_t st = (hipStream_t)stream;
  const int groups = (s->B + 7) / 8;
  // flags and granules start from zero; the 64-byte status header is sticky (cleared by the host after it has read it)
  int rc = las_check_hip(hipMemsetAsync(reinterpret_cast<char*>(p->workspace) + 64, 0,
                                        las_decoder_persist_workspace_bytes(s->B, s->Tm, s->Hd, s->M) - 64, st), "memset workspace");
  if (rc) return rc;
  size_t lds = persist_bwd_scratch_floats(s->M, s->Hd, s->Tm, two) * sizeof(float);
  LAS_REQUIRE(lds <= 64 * 1024, "las_decoder_persist_bwd: shapes exceed the LDS budget");
  // the matrix-core passes need the frames in LDS and M = 512, 1024 or 2048; with Luong scores they keep the keys transposed
  const bool mshape = s->M == 512 || s->M == 1024 || (s->M == 2048 && s->Hd == 512);
  const bool keys_t = s->attention == LAS_ATT_LUONG && mshape && persist_bwd_resident(s->M, s->Hd, s->Tm, true, two);
  const bool res = persist_bwd_resident(s->M, s->Hd, s->Tm, keys_t, two);
  if (res) lds += persist_bwd_resident_bytes(s->M, s->Hd, s->Tm, keys_t);
  else {       // long memories: the key frames and the first rows of the value frames (persist_bwd_resident_rows)
    const int vfr = persist_bwd_resident_rows(s->M, s->Hd, s->Tm, two);
    if (vfr > 0) lds += ((size_t)vfr * (s->M + P_VPAD) + (size_t)((s->Tm + 3) / 4) * s->Hd) * 2;
  }
  const int npq = (res && mshape && (keys_t || s->attention != LAS_ATT_LUONG)) ? s->M / 128 : 0;
  const dim3 grid(((groups + 7) & ~7) * P_MEMBERS);
  const bool wq = s->attention != LAS_ATT_LUONG;
#define LAS_BWD_LAUNCH(...)                                                                                                     \
  do {                                                                                                                          \
    static bool attr = false;                                                                                                   \
    if (!attr) {                                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_persist_bwd_kernel<__VA_ARGS__>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                        \
      attr = true;                                                                                                              \
    }                                                                                                                           \
    hipLaunchKernelGGL((dec_persist_bwd_kernel<__VA_ARGS__>), grid, dim3(256), lds, st, *p);                                    \
  } while (0)
  if (s->Hd == 512) {            // 5 column tiles per member, 6 of 16 K chunks per wave resident, 8 d(context) columns and 2 units per thread
    if (wq) {
      // (two resident chunks fewer than without the query layer: that part's registers.  With 5 of them the kernel spilled 72-76
      // bytes per lane; 4 = no scratch at the same time per step -- round 5, metric-L: 2.185 against 2.189 ms per launch)
      if (npq == 16) LAS_BWD_LAUNCH(true, 16, 5, 4, 16, 8, 2);
      else LAS_BWD_LAUNCH(true, 0, 5, 4, 16, 8, 2);
    } else {
      if (npq == 16) LAS_BWD_LAUNCH(false, 16, 5, 6, 16, 8, 2);
      else LAS_BWD_LAUNCH(false, 0, 5, 6, 16, 8, 2);
    }
  } else if (two) {              // second cell: the same passes, + its plain backward and its streamed product
    if (wq) {
      if (npq == 8) LAS_BWD_LAUNCH(true, 8, 3, 8, 8, 6, 1, true);
      else if (npq == 4) LAS_BWD_LAUNCH(true, 4, 3, 8, 8, 6, 1, true);
      else LAS_BWD_LAUNCH(true, 0, 3, 8, 8, 6, 1, true);
    } else {
      if (npq == 8) LAS_BWD_LAUNCH(false, 8, 3, 8, 8, 6, 1, true);
      else if (npq == 4) LAS_BWD_LAUNCH(false, 4, 3, 8, 8, 6, 1, true);
      else LAS_BWD_LAUNCH(false, 0, 3, 8, 8, 6, 1, true);
    }
  } else if (wq) {
    if (npq == 8) LAS_BWD_LAUNCH(true, 8);
    else if (npq == 4) LAS_BWD_LAUNCH(true, 4);
    else LAS_BWD_LAUNCH(true, 0);
  } else {
    if (npq == 8) LAS_BWD_LAUNCH(false, 8);
    else if (npq == 4) LAS_BWD_LAUNCH(false, 4);
    else LAS_BWD_LAUNCH(false, 0);
  }
#undef LAS_BWD_LAUNCH
  LAS_LAUNCH_CHECK("persistent decoder bwd launch");
  return LAS_OK;
}

extern "C" size_t las_decoder_seq_xchg_bytes(int B, int Tm, int Hd, int M, int W0) {
  return B > 0 && Tm > 0 && Hd > 0 && M > 0 && W0 > 0 ? 64 + (size_t)B * seq_xlayout(Tm, Hd, M, W0).total * sizeof(pu64) : 0;
}

extern "C" size_t las_decoder_sum_workspace_bytes(int blocks, int n) {
  return blocks > 0 && n > 0 ? 64 + (size_t)blocks * n * sizeof(float) : 0;
}

extern "C" int las_decoder_seq_bwd_supported(int Hd, int M, int A, int W0, int Tm, int attention, int norm) {
  if (Hd != 128 && Hd != 256) return 0;                       // dz_t as one or two 16-byte pieces per lane
  if (M % 128 != 0 || A < 0 || A % 8 != 0 || W0 <= 0) return 0;
  if (attention < LAS_ATT_LUONG || attention > LAS_ATT_BAHDANAU_MONOTONIC) return 0;
  if (norm != LAS_NORM_SOFTMAX && norm != LAS_NORM_MONOTONIC_PARALLEL) return 0;
  const size_t lds = (dec_step_bwd_floats(M, Tm, Hd, norm) + (size_t)(A > 0 ? A + Hd + M : 0) + W0 + 2 * Hd + 8 + Hd + 8) * sizeof(float);
  return lds <= 64 * 1024 ? 1 : 0;      // (+ Tm * (A + 1) floats when the caller hands in VW: checked at the launch against 160 KiB)
}

extern "C" int las_decoder_seq_bwd(const las_dec_seq_bwd* p, void* stream) {
  const las_dec_step_bwd* s = &p->s;
  LAS_REQUIRE(s->B > 0 && p->U > 0 && las_decoder_seq_bwd_supported(s->Hd, s->M, p->A, p->W0, s->Tm, s->attention, s->norm),
              "las_decoder_seq_bwd: configuration not supported (Hd=%d M=%d A=%d W0=%d Tm=%d attention=%d norm=%d)", s->Hd, s->M, p->A,
              p->W0, s->Tm, s->attention, s->norm);
  const bool additive = s->attention == LAS_ATT_BAHDANAU || s->attention == LAS_ATT_BAHDANAU_MONOTONIC;
  LAS_REQUIRE(p->d_out && p->kn_packed && s->dc && s->dz && s->ds_out && s->dctx_save && s->align && s->gates && s->c_new && s->c_prev && s->keys &&
                  s->values && s->mem_len && (p->A == 0 || (p->waln_packed && p->datt_out)) && p->W0 == (p->A > 0 ? p->A : s->M) + s->Hd,
              "las_decoder_seq_bwd: null argument or inconsistent widths");
  LAS_REQUIRE(!additive || (s->wq_t && s->att_v && s->pq && s->dkeys_acc && s->dv_acc), "las_decoder_seq_bwd: Bahdanau scores need wq_t, att_v, pq, dkeys_acc, dv_acc");
  LAS_REQUIRE(s->attention != LAS_ATT_CUSTOM || (s->wq_t && s->pq), "las_decoder_seq_bwd: CustomAttention needs wq_t and the saved processed query");
  LAS_REQUIRE(s->norm == LAS_NORM_SOFTMAX || (s->p && s->dalign_carry), "las_decoder_seq_bwd: monotonic attention needs p_choose and the carry buffer");
  LAS_REQUIRE(s->drop_keep >= 1.0f, "las_decoder_seq_bwd: without input dropout");
  size_t lds = (dec_step_bwd_floats(s->M, s->Tm, s->Hd, s->norm) + (size_t)(p->A > 0 ? p->A + s->Hd + s->M : 0) + p->W0 + 2 * s->Hd + 8 + s->Hd + 8) * sizeof(float);
  las_dec_seq_bwd q = *p;
  if (q.A > 0 && q.vw) {
    const size_t with_vw = lds + (size_t)s->Tm * (q.A + 1) * sizeof(float);
    if (with_vw <= 158 * 1024) lds = with_vw;
    else q.vw = nullptr;                             // does not fit the CU's LDS: d(alignments) from the values, as without VW
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_seq_bwd_kernel<0, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_seq_bwd_kernel<0, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_seq_bwd_kernel<SEQ_NPK, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_seq_bwd_kernel<SEQ_NPK, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_seq_bwd_kernel<SEQ_NPK4, 16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_seq_bwd_kernel<SEQ_NPK4, 32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  // Bahdanau scores: d(keys) in registers when the utterance's frames fit SEQ_NPK passes of the 256 / (Hd / 8) frame phases
  // (Tm <= 200 at 256 units, 400 at 128); LAS_DEC_SEQ_REGK=0 keeps the read-modify-write in memory (A/B measurements)
  const bool regk_on = las_knob("LAS_DEC_SEQ_REGK", 1) != 0;
  const int P = 256 / (s->Hd / 8);
  const bool regk = additive && regk_on && q.A > 0 && q.vw && (s->Tm + P - 1) / P <= SEQ_NPK;
  // ... and four workgroups per utterance (the caller hands in the exchange workspace) when every one of them finds a CU of its
  // own -- they wait for one another -- and a quarter of the frames fits SEQ_NPK4 passes
  static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int blocks4 = (s->B + 7) / 8 * 32, fq = (s->Tm + 3) / 4;
  const bool parts4 = regk && q.xchg_workspace && blocks4 <= cus && (fq + P - 1) / P <= SEQ_NPK4 && s->dv_acc && q.sum_workspace && q.A <= 128;
#define LAS_SEQ_LAUNCH(NPK_, KC_) hipLaunchKernelGGL((dec_seq_bwd_kernel<NPK_, KC_>), dim3(s->B), dim3(256), lds, (hipStream_t)stream, q)
#define LAS_SEQ_LAUNCH4(KC_) hipLaunchKernelGGL((dec_seq_bwd_kernel<SEQ_NPK4, KC_, 4>), dim3(blocks4), dim3(256), lds, (hipStream_t)stream, q)
  if (parts4) {
    int rc = las_check_hip(hipMemsetAsync(static_cast<char*>(q.xchg_workspace) + 64, 0, las_decoder_seq_xchg_bytes(s->B, s->Tm, s->Hd, s->M, q.W0) - 64, (hipStream_t)stream),
                           "memset exchange workspace");
    if (rc) return rc;
    if (s->Hd == 128) LAS_SEQ_LAUNCH4(16); else LAS_SEQ_LAUNCH4(32);
  } else {
    q.xchg_workspace = nullptr;
    if (s->Hd == 128) { if (regk) LAS_SEQ_LAUNCH(SEQ_NPK, 16); else LAS_SEQ_LAUNCH(0, 16); }
    else              { if (regk) LAS_SEQ_LAUNCH(SEQ_NPK, 32); else LAS_SEQ_LAUNCH(0, 32); }
  }
#undef LAS_SEQ_LAUNCH
#undef LAS_SEQ_LAUNCH4
  LAS_LAUNCH_CHECK("sequential decoder bwd launch");
  return LAS_OK;
}

extern "C" int las_decoder_step_bwd(const las_dec_step_bwd* s, void* stream) {
  LAS_REQUIRE(s->B > 0 && s->Hd >= 64 && s->Hd <= 1024 && (s->Hd & (s->Hd - 1)) == 0 && s->M % 128 == 0 && s->Tm > 0,
              "las_decoder_step_bwd: decoder_units must be a power of two in [64,1024], memory depth a multiple of 128");
  const int att = s->attention;
  const bool additive = att == LAS_ATT_BAHDANAU || att == LAS_ATT_BAHDANAU_MONOTONIC;
  LAS_REQUIRE(att >= LAS_ATT_LUONG && att <= LAS_ATT_BAHDANAU_MONOTONIC, "las_decoder_step_bwd: unknown attention %d", att);
  LAS_REQUIRE(s->mode == LAS_DEC_CELL_ONLY || !additive || (s->wq_t && s->att_v && s->pq && s->dkeys_acc && s->dv_acc),
              "las_decoder_step_bwd: Bahdanau scores need wq_t, att_v, pq, dkeys_acc, dv_acc");
  LAS_REQUIRE(s->mode == LAS_DEC_CELL_ONLY || att != LAS_ATT_CUSTOM || (s->wq_t && s->pq),
              "las_decoder_step_bwd: CustomAttention needs wq_t and the saved processed query");
  LAS_REQUIRE(s->norm == LAS_NORM_SOFTMAX || s->norm == LAS_NORM_MONOTONIC_PARALLEL,
              "las_decoder_step_bwd: normaliser %d has no backward ('hard' monotonic attention is inference only)", s->norm);
  LAS_REQUIRE(s->mode == LAS_DEC_CELL_ONLY || s->norm == LAS_NORM_SOFTMAX || (s->p && s->dalign_carry),
              "las_decoder_step_bwd: monotonic attention needs the saved p_choose and the dalign_carry buffer");
  LAS_REQUIRE(s->mode != LAS_DEC_ATTENTION_ONLY || s->dq_out, "las_decoder_step_bwd: attention-only mode needs dq_out");
  const size_t lds = (size_t)(s->M + s->Tm + 2048 + 8 + s->Hd + (s->norm != LAS_NORM_SOFTMAX ? 5 * s->Tm : 0) + 2048) * sizeof(float);
  LAS_REQUIRE(lds <= 64 * 1024, "las_decoder_step_bwd: shapes exceed the LDS budget");
  hipLaunchKernelGGL(dec_step_bwd_kernel, dim3(s->B), dim3(256), lds, (hipStream_t)stream, *s);
  LAS_LAUNCH_CHECK("decoder step bwd launch");
  return LAS_OK;
}

extern "C" int las_seq_sigmoid_loss(const float* logits, int64_t ldl, const las_bf16* targets, int64_t ldt,
                                    const int32_t* seq_len, int B, int U, int nf, float grad_scale, float* loss_out,
                                    las_bf16* dlogits, int64_t ldd, void* stream) {
  LAS_REQUIRE(B > 0 && U > 0 && nf > 0 && logits && targets && seq_len && loss_out, "las_seq_sigmoid_loss: bad arguments");
  int blocks = (B * U + 3) / 4;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(seq_sigmoid_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, ldl, targets, ldt, seq_len, B,
                     U, nf, grad_scale, loss_out, dlogits, ldd);
  LAS_LAUNCH_CHECK("seq sigmoid launch");
  return LAS_OK;
}

extern "C" int las_sample_features(const float* logits, int64_t ldl, int nf, const las_bf16* teacher, int64_t ldt,
                                   las_bf16* next, int64_t ldn, int B, float prob, uint32_t seed, uint32_t step, void* stream) {
  LAS_REQUIRE(B > 0 && nf > 0 && logits && next && prob >= 0.f && prob <= 1.f, "las_sample_features: bad arguments");
  hipLaunchKernelGGL(sample_features_kernel, dim3((B * nf + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, ldl, nf,
                     teacher, ldt, next, ldn, B, prob, seed, step);
  LAS_LAUNCH_CHECK("sample features launch");
  return LAS_OK;
}

extern "C" int las_seq_ce_loss(const float* logits, int64_t ldl, const int32_t* targets, const int32_t* target_len,
                               int B, int U, int V, float grad_scale, float* loss_out, las_bf16* dlogits, int64_t ldd,
                               void* stream) {
  LAS_REQUIRE(B > 0 && U > 0 && V > 0, "las_seq_ce_loss: bad shape");
  int blocks = (B * U + 3) / 4;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(seq_ce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, ldl, targets, target_len, B,
                     U, V, grad_scale, loss_out, dlogits, ldd);
  LAS_LAUNCH_CHECK("seq ce launch");
  return LAS_OK;
}

#ifdef LAS_STAMPS
extern "C" int las_debug_read_stamps(unsigned long long* out_host, int n) {
  return las_check_hip(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(las_stamps), sizeof(unsigned long long) * (size_t)n), "read stamps");
}
#endif
