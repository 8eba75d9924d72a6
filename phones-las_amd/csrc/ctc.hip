// CTC head of model_helper.py:347-367: tf.nn.ctc_loss_v2 on dense labels (blank index 0 — the reference's quirk
// B7 — labels include the trailing </s>, label_length = target_sequence_length, logit_length = the reduced
// encoder length), mean over the batch, times ctc_weight.  One workgroup per utterance: log-softmax rows, the
// alpha and beta recursions in log space over the blank-extended label sequence (serial in time, parallel over the
// 2L+1 states, previous column in LDS), then the closed-form gradient w.r.t. the logits.
// Round 6 (the kernel was 1.0 ms of a cfg4 step, on the critical path between the two decoders): the same arithmetic in the same
// order, moved differently -- log-softmax a row per wave (coalesced, lane-parallel reductions in a fixed tree) instead of a row per
// thread; in the recursions a state's emission of the NEXT frame is requested before the step's barrier (the step no longer waits
// for a global gather); the gradient walks each class's own list of states (ascending, as before) instead of all 2L+1 per element.
#include "las_common.h"

namespace {

constexpr float NEG = -1e30f;
constexpr int LONG_LIST = 8;      // states of one class beyond which the gradient walks the class's list a frame per lane

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m <= NEG) return NEG;
  return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) { return lse2(lse2(a, b), c); }
// a barrier that orders LDS accesses only (this wave's LDS operations done, then everybody's): global loads and stores stay in flight
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int NT = 512;               // threads per workgroup (one workgroup per utterance)
__global__ __launch_bounds__(NT) void ctc_kernel(const float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ labels,
                                                  int64_t ldlab, const int32_t* __restrict__ label_len,
                                                  const int32_t* __restrict__ logit_len, int T, int C, int U, int blank,
                                                  float loss_scale, float grad_scale, float* __restrict__ lp_ws,
                                                  float* __restrict__ ab_ws, int Sp, float* __restrict__ loss_out,
                                                  float* __restrict__ per_example, unsigned short* __restrict__ dlogits, int g_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* col = reinterpret_cast<float*>(smem);          // [2][Sp] previous / current column
  int* ext = reinterpret_cast<int*>(col + 2 * Sp);      // [Sp]
  float* G = reinterpret_cast<float*>(ext + Sp);        // [g_rows][Sp]: alpha + beta of a block of frames (the gradient's staging area)
  int* cnt = reinterpret_cast<int*>(G + (size_t)g_rows * Sp);   // [C] states per class, [1] how many are long, [C] the classes with long lists
  float* colb = reinterpret_cast<float*>(cnt + 2 * C + 2);      // [2][Sp] beta's columns (alpha's: col)
  __shared__ float total_lp;

  const int b = blockIdx.x, tid = threadIdx.x;
  const int L = min(label_len[b], U);
  const int Tb = min(logit_len[b], T);
  const int S = 2 * L + 1;
  const float* lg = logits + (int64_t)b * T * ldl;
  float* lp = lp_ws + (int64_t)b * T * ldl;
  float* A = ab_ws + (int64_t)b * 2 * T * Sp;
  float* Bt = A + (int64_t)T * Sp;
  unsigned short* dl = dlogits ? dlogits + (int64_t)b * T * ldl : nullptr;

  for (int s = tid; s < S; s += NT) ext[s] = (s & 1) ? labels[(int64_t)b * ldlab + (s >> 1)] : blank;
  // log-softmax rows: a row per wave at a time, each element loaded ONCE (a lane holds classes lane and lane + 64; more than 128
  // classes: the loop form), the next row requested before this one is reduced; the sum over classes in the order c = 0, 1, ...
  // (as a thread per row formed it): the lanes hold the exponentials and every lane adds them up one after the other
  {
    const int lane = tid & 63, wave = tid >> 6;
    if (C <= 128) {
      const bool h0 = lane < C, h1 = lane + 64 < C;
      float n0 = 0.f, n1 = 0.f;
      if (wave < Tb) { if (h0) n0 = lg[(int64_t)wave * ldl + lane]; if (h1) n1 = lg[(int64_t)wave * ldl + lane + 64]; }
      for (int t = wave; t < Tb; t += NT / 64) {
        const float x0 = n0, x1 = n1;
        if (t + NT / 64 < Tb) { if (h0) n0 = lg[(int64_t)(t + NT / 64) * ldl + lane]; if (h1) n1 = lg[(int64_t)(t + NT / 64) * ldl + lane + 64]; }
        const float mx = las_wave_max(fmaxf(h0 ? x0 : -INFINITY, h1 ? x1 : -INFINITY));
        const float e0 = h0 ? __expf(x0 - mx) : 0.f, e1 = h1 ? __expf(x1 - mx) : 0.f;
        float sum = 0.f;
        const int na = min(64, C), nb = C - na;
        for (int i = 0; i < na; ++i) sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e0), i));
        for (int i = 0; i < nb; ++i) sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e1), i));
        const float lse = mx + __logf(sum);
        if (h0) lp[(int64_t)t * ldl + lane] = x0 - lse;
        if (h1) lp[(int64_t)t * ldl + lane + 64] = x1 - lse;
      }
    } else {
      for (int t = wave; t < Tb; t += NT / 64) {
        const float* row = lg + (int64_t)t * ldl;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
        mx = las_wave_max(mx);
        float sum = 0.f;
        for (int c0 = 0; c0 < C; c0 += 64) {
          const int c = c0 + lane;
          const float e = c < C ? __expf(row[c] - mx) : 0.f;
          const int n = min(64, C - c0);
          for (int i = 0; i < n; ++i) sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), i));
        }
        const float lse = mx + __logf(sum);
        for (int c = lane; c < C; c += 64) lp[(int64_t)t * ldl + c] = row[c] - lse;
      }
    }
  }
  __syncthreads();
  if (Tb <= 0) {
    if (tid == 0) { if (per_example) per_example[b] = 0.f; }
    if (dl) for (int i = tid; i < T * (int)ldl; i += NT) dl[i] = 0;
    return;
  }

  // alpha and beta (each includes the emission at its frame).  At most 256 states: the two recursions run SIDE BY SIDE -- threads
  // 0..255 a state of alpha each, threads 256..511 a state of beta, one barrier per frame for both -- with the state's class, its
  // skip transition and the NEXT frame's emission held in registers (requested a step ahead: the step never waits for a gather).
  int cur = 0;
  const int half = tid >> 8, ht = tid & 255;
  if (S <= 256) {
    const int s = ht;
    const bool on = s < S;
    const int es = on ? ext[s] : 0;
    if (on) {
      if (half == 0) { const float v = (s < 2) ? lp[es] : NEG; col[s] = v; A[s] = v; }
      else { const float v = (s >= S - 2) ? lp[(int64_t)(Tb - 1) * ldl + es] : NEG; colb[s] = v; Bt[(int64_t)(Tb - 1) * Sp + s] = v; }
    }
    __syncthreads();
    const bool skip = on && (half == 0 ? (s >= 2 && es != blank && es != ext[s - 2]) : (s + 2 < S && ext[s + 2] != blank && ext[s + 2] != es));
    float e_next = (on && Tb > 1) ? lp[(int64_t)(half == 0 ? 1 : Tb - 2) * ldl + es] : 0.f;
    for (int k = 1; k < Tb; ++k) {
      const int t = half == 0 ? k : Tb - 1 - k;
      const float e = e_next;
      if (on && k + 1 < Tb) e_next = lp[(int64_t)(half == 0 ? t + 1 : t - 1) * ldl + es];
      if (on) {
        if (half == 0) {
          const float* prev = col + cur * Sp;
          const float a0 = prev[s];
          const float a1 = s >= 1 ? prev[s - 1] : NEG;
          const float a2 = skip ? prev[s - 2] : NEG;
          const float v = lse3(a0, a1, a2) + e;
          col[(cur ^ 1) * Sp + s] = v;
          A[(int64_t)t * Sp + s] = v;
        } else {
          const float* prev = colb + cur * Sp;
          const float b0 = prev[s];
          const float b1 = s + 1 < S ? prev[s + 1] : NEG;
          const float b2 = skip ? prev[s + 2] : NEG;
          const float v = lse3(b0, b1, b2) + e;
          colb[(cur ^ 1) * Sp + s] = v;
          Bt[(int64_t)t * Sp + s] = v;
        }
      }
      lds_barrier();          // (not __syncthreads: that would wait for the emission just requested and for the stores of A / B)
      cur ^= 1;
    }
  } else {
    // more states than threads of a half: one recursion after the other, the states dealt to all threads
    for (int s = tid; s < S; s += NT) {
      const float v = (s < 2) ? lp[ext[s]] : NEG;
      col[s] = v;
      A[s] = v;
    }
    __syncthreads();
    for (int t = 1; t < Tb; ++t) {
      const float* prev = col + cur * Sp;
      float* nxt = col + (cur ^ 1) * Sp;
      for (int s = tid; s < S; s += NT) {
        const float a0 = prev[s];
        const float a1 = s >= 1 ? prev[s - 1] : NEG;
        const float a2 = (s >= 2 && ext[s] != blank && ext[s] != ext[s - 2]) ? prev[s - 2] : NEG;
        const float v = lse3(a0, a1, a2) + lp[(int64_t)t * ldl + ext[s]];
        nxt[s] = v;
        A[(int64_t)t * Sp + s] = v;
      }
      __syncthreads();
      cur ^= 1;
    }
    int curb = 0;
    for (int s = tid; s < S; s += NT) {
      const float v = (s >= S - 2) ? lp[(int64_t)(Tb - 1) * ldl + ext[s]] : NEG;
      colb[s] = v;
      Bt[(int64_t)(Tb - 1) * Sp + s] = v;
    }
    __syncthreads();
    for (int t = Tb - 2; t >= 0; --t) {
      const float* prev = colb + curb * Sp;
      float* nxt = colb + (curb ^ 1) * Sp;
      for (int s = tid; s < S; s += NT) {
        const float b0 = prev[s];
        const float b1 = s + 1 < S ? prev[s + 1] : NEG;
        const float b2 = (s + 2 < S && ext[s + 2] != blank && ext[s + 2] != ext[s]) ? prev[s + 2] : NEG;
        const float v = lse3(b0, b1, b2) + lp[(int64_t)t * ldl + ext[s]];
        nxt[s] = v;
        Bt[(int64_t)t * Sp + s] = v;
      }
      __syncthreads();
      curb ^= 1;
    }
  }
  if (tid == 0) {
    const float* last = col + cur * Sp;
    total_lp = S > 1 ? lse2(last[S - 1], last[S - 2]) : last[S - 1];
  }
  __syncthreads();
  const float lp_total = total_lp;

  if (tid == 0) {
    const float loss = -lp_total;
    if (per_example) per_example[b] = loss;
    if (loss_out) atomicAdd(loss_out, loss * loss_scale);
  }
  if (!dl) return;
  // gradient: y - (1/p) sum_{s: ext[s]=c} alpha_t(s) beta_t(s) / y.  Every class's states as a list (ascending s: the order the
  // sum has always been formed in), built once per utterance in the column buffer, which the recursions have left: next[s] = the
  // next state of the same class (or -1), first[c] = its first -- where 2 Sp floats hold S + C ints (else: the scan over all states)
  __syncthreads();
  int* nexts = reinterpret_cast<int*>(col);
  int* first = nexts + S;
  const bool lists = S + C <= 2 * Sp;
  if (lists) {
    for (int c = tid; c < C; c += NT) first[c] = -1;
    __syncthreads();
    if (g_rows > 0) for (int c = tid; c < C; c += NT) cnt[c] = 0;
    __syncthreads();
    if (tid == 0) {
      for (int s = S - 1; s >= 0; --s) { nexts[s] = first[ext[s]]; first[ext[s]] = s; if (g_rows > 0) ++cnt[ext[s]]; }
      if (g_rows > 0) {           // the classes whose lists are long (the blank: L + 1 states), in class order
        int n = 0;
        for (int c = 0; c < C; ++c) if (cnt[c] > LONG_LIST) cnt[C + 1 + n++] = c;
        cnt[C] = n;
      }
    }
    __syncthreads();
  }
  if (lists && g_rows > 0) {
    // frames in blocks of g_rows: alpha + beta of the block's states into the LDS (coalesced rows), then an element (t, c) walks its
    // class's list there -- the blank's is L + 1 states long, and a walk through global memory was a dependent load per state
    for (int t0 = 0; t0 < Tb; t0 += g_rows) {
      const int nr = min(g_rows, Tb - t0);
      __syncthreads();
      for (int i = tid; i < nr * S; i += NT) {
        const int r = i / S, st = i - r * S;
        G[r * Sp + st] = A[(int64_t)(t0 + r) * Sp + st] + Bt[(int64_t)(t0 + r) * Sp + st];
      }
      __syncthreads();
      // a wave's 64 lanes walk in lockstep: one lane with the blank's list (L + 1 states) among 63 with a state or two made every
      // pass cost the long walk (the whole phase 0.43 ms at cfg4).  The long classes first, a FRAME per lane (equal walks side by
      // side), then everything else
      const int nlong = cnt[C];
      auto element = [&](int r, int c) {
        const int t = t0 + r;
        const float l = lp[(int64_t)t * ldl + c];
        float acc = NEG;
        for (int st = first[c]; st >= 0; st = nexts[st]) acc = lse2(acc, G[r * Sp + st]);
        const float g = __expf(l) - (acc > NEG ? __expf(acc - l - lp_total) : 0.f);
        dl[(int64_t)t * ldl + c] = las_f2bf(g * grad_scale);
      };
      for (int i = tid; i < nr * nlong; i += NT) element(i % nr, cnt[C + 1 + i / nr]);
      for (int i = tid; i < nr * C; i += NT) {
        const int r = i / C, c = i - r * C;
        if (cnt[c] <= LONG_LIST) element(r, c);
      }
    }
    for (int i = Tb * C + tid; i < T * C; i += NT) dl[(int64_t)(i / C) * ldl + i % C] = las_f2bf(0.f * grad_scale);
  } else
  for (int i = tid; i < T * C; i += NT) {
    const int t = i / C, c = i % C;
    float g = 0.f;
    if (t < Tb) {
      const float l = lp[(int64_t)t * ldl + c];
      float acc = NEG;
      if (lists) {
        for (int s = first[c]; s >= 0; s = nexts[s]) acc = lse2(acc, A[(int64_t)t * Sp + s] + Bt[(int64_t)t * Sp + s]);
      } else {
        for (int s = 0; s < S; ++s)
          if (ext[s] == c) acc = lse2(acc, A[(int64_t)t * Sp + s] + Bt[(int64_t)t * Sp + s]);
      }
      g = __expf(l) - (acc > NEG ? __expf(acc - l - lp_total) : 0.f);
    }
    dl[(int64_t)t * ldl + c] = las_f2bf(g * grad_scale);
  }
  for (int i = tid; i < T * ((int)ldl - C); i += NT) {      // zero the pad columns
    const int t = i / ((int)ldl - C), c = C + i % ((int)ldl - C);
    dl[(int64_t)t * ldl + c] = 0;
  }
}

}  // namespace

extern "C" size_t las_ctc_workspace_bytes(int B, int T, int C_padded, int U) {
  const size_t Sp = 2 * (size_t)U + 1 + 3;
  return ((size_t)B * T * C_padded + (size_t)B * 2 * T * Sp) * sizeof(float);
}

extern "C" int las_ctc_loss(const float* logits, int64_t ldl, const int32_t* labels, int64_t ldlab, const int32_t* label_len,
                            const int32_t* logit_len, int B, int T, int C, int U, int blank, float loss_scale,
                            float grad_scale, void* workspace, float* loss_out, float* per_example, las_bf16* dlogits,
                            void* stream) {
  LAS_REQUIRE(B > 0 && T > 0 && C > 1 && U > 0 && ldl >= C && blank >= 0 && blank < C, "las_ctc_loss: bad shape");
  const int Sp = 2 * U + 1 + 3;
  const size_t lds0 = (size_t)(2 * Sp) * sizeof(float) + (size_t)Sp * sizeof(int);
  LAS_REQUIRE(lds0 <= 64 * 1024, "las_ctc_loss: label length %d too long", U);
  // + the gradient's staging area: as many frames of alpha + beta as 96 KiB hold (cfg4: all 100)
  const int g_rows = (int)((96 * 1024) / ((size_t)Sp * sizeof(float))) < T ? (int)((96 * 1024) / ((size_t)Sp * sizeof(float))) : T;
  const size_t lds = lds0 + (size_t)g_rows * Sp * sizeof(float) + (size_t)(2 * C + 2) * sizeof(int) + (size_t)(2 * Sp) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ctc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    attr_set = true;
  }
  float* lp_ws = reinterpret_cast<float*>(workspace);
  float* ab_ws = lp_ws + (size_t)B * T * ldl;
  hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, logits, ldl, labels, ldlab, label_len, logit_len, T,
                     C, U, blank, loss_scale, grad_scale, lp_ws, ab_ws, Sp, loss_out, per_example, dlogits, g_rows);
  LAS_LAUNCH_CHECK("ctc launch");
  return LAS_OK;
}
