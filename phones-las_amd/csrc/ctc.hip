// CTC head of model_helper.py:347-367: tf.nn.ctc_loss_v2 on dense labels (blank index 0 — the reference's quirk
// B7 — labels include the trailing </s>, label_length = target_sequence_length, logit_length = the reduced
// encoder length), mean over the batch, times ctc_weight.  One workgroup per utterance: log-softmax rows, the
// alpha and beta recursions in log space over the blank-extended label sequence (serial in time, parallel over the
// 2L+1 states, previous column in LDS), then the closed-form gradient w.r.t. the logits.
#include "las_common.h"

namespace {

constexpr float NEG = -1e30f;

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m <= NEG) return NEG;
  return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) { return lse2(lse2(a, b), c); }

__global__ __launch_bounds__(256) void ctc_kernel(const float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ labels,
                                                  int64_t ldlab, const int32_t* __restrict__ label_len,
                                                  const int32_t* __restrict__ logit_len, int T, int C, int U, int blank,
                                                  float loss_scale, float grad_scale, float* __restrict__ lp_ws,
                                                  float* __restrict__ ab_ws, int Sp, float* __restrict__ loss_out,
                                                  float* __restrict__ per_example, unsigned short* __restrict__ dlogits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* col = reinterpret_cast<float*>(smem);          // [2][Sp] previous / current column
  int* ext = reinterpret_cast<int*>(col + 2 * Sp);      // [Sp]
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

  for (int s = tid; s < S; s += 256) ext[s] = (s & 1) ? labels[(int64_t)b * ldlab + (s >> 1)] : blank;
  // log-softmax rows
  for (int t = tid; t < Tb; t += 256) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lg[(int64_t)t * ldl + c]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += __expf(lg[(int64_t)t * ldl + c] - mx);
    const float lse = mx + __logf(sum);
    for (int c = 0; c < C; ++c) lp[(int64_t)t * ldl + c] = lg[(int64_t)t * ldl + c] - lse;
  }
  __syncthreads();
  if (Tb <= 0) {
    if (tid == 0) { if (per_example) per_example[b] = 0.f; }
    if (dl) for (int i = tid; i < T * (int)ldl; i += 256) dl[i] = 0;
    return;
  }

  // alpha
  int cur = 0;
  for (int s = tid; s < S; s += 256) {
    const float v = (s < 2) ? lp[ext[s]] : NEG;
    col[s] = v;
    A[s] = v;
  }
  __syncthreads();
  for (int t = 1; t < Tb; ++t) {
    const float* prev = col + cur * Sp;
    float* nxt = col + (cur ^ 1) * Sp;
    for (int s = tid; s < S; s += 256) {
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
  if (tid == 0) {
    const float* last = col + cur * Sp;
    total_lp = S > 1 ? lse2(last[S - 1], last[S - 2]) : last[S - 1];
  }
  __syncthreads();
  const float lp_total = total_lp;

  // beta (includes the emission at t, like alpha)
  cur = 0;
  for (int s = tid; s < S; s += 256) {
    const float v = (s >= S - 2) ? lp[(int64_t)(Tb - 1) * ldl + ext[s]] : NEG;
    col[s] = v;
    Bt[(int64_t)(Tb - 1) * Sp + s] = v;
  }
  __syncthreads();
  for (int t = Tb - 2; t >= 0; --t) {
    const float* prev = col + cur * Sp;
    float* nxt = col + (cur ^ 1) * Sp;
    for (int s = tid; s < S; s += 256) {
      const float b0 = prev[s];
      const float b1 = s + 1 < S ? prev[s + 1] : NEG;
      const float b2 = (s + 2 < S && ext[s + 2] != blank && ext[s + 2] != ext[s]) ? prev[s + 2] : NEG;
      const float v = lse3(b0, b1, b2) + lp[(int64_t)t * ldl + ext[s]];
      nxt[s] = v;
      Bt[(int64_t)t * Sp + s] = v;
    }
    __syncthreads();
    cur ^= 1;
  }

  if (tid == 0) {
    const float loss = -lp_total;
    if (per_example) per_example[b] = loss;
    if (loss_out) atomicAdd(loss_out, loss * loss_scale);
  }
  if (!dl) return;
  // gradient: y - (1/p) sum_{s: ext[s]=c} alpha_t(s) beta_t(s) / y
  for (int i = tid; i < T * C; i += 256) {
    const int t = i / C, c = i % C;
    float g = 0.f;
    if (t < Tb) {
      const float l = lp[(int64_t)t * ldl + c];
      float acc = NEG;
      for (int s = 0; s < S; ++s)
        if (ext[s] == c) acc = lse2(acc, A[(int64_t)t * Sp + s] + Bt[(int64_t)t * Sp + s]);
      g = __expf(l) - (acc > NEG ? __expf(acc - l - lp_total) : 0.f);
    }
    dl[(int64_t)t * ldl + c] = las_f2bf(g * grad_scale);
  }
  for (int i = tid; i < T * ((int)ldl - C); i += 256) {      // zero the pad columns
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
  const size_t lds = (size_t)(2 * Sp) * sizeof(float) + (size_t)Sp * sizeof(int);
  LAS_REQUIRE(lds <= 64 * 1024, "las_ctc_loss: label length %d too long", U);
  float* lp_ws = reinterpret_cast<float*>(workspace);
  float* ab_ws = lp_ws + (size_t)B * T * ldl;
  hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, logits, ldl, labels, ldlab, label_len, logit_len, T,
                     C, U, blank, loss_scale, grad_scale, lp_ws, ab_ws, Sp, loss_out, per_example, dlogits);
  LAS_LAUNCH_CHECK("ctc launch");
  return LAS_OK;
}
