// Speller step kernels: one AttentionWrapper(LSTMCell, Luong|Bahdanau) step of las/model.py:145-202 /
// SURVEY.md Appendix A.5-A.7, forward and backward, one 256-thread workgroup per utterance.
//
// The dense parts of a step ([attention_{t-1}, h_{t-1}] * K and its transpose) run in las_gemm_nt; these
// kernels fuse everything else of the step: token-row gather + bias + LSTM gate math, the score of h_t
// against the keys, the length-masked softmax (wave-shuffle + LDS reductions) and the context
// sum_t' align * values.  keys/values rows are read as 16-byte bf16x8 pieces (16 lanes per memory frame
// for scores, 4 columns per lane for the context).
#include "las_common.h"

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

__device__ __forceinline__ float dot8(const uint4& k, const float* q) {
  const unsigned short* e = reinterpret_cast<const unsigned short*>(&k);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += las_bf2f(e[j]) * q[j];
  return s;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_step_fwd_kernel(las_dec_step s) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* hq = sm;                 // [Hd] h_t (bf16-rounded) as float
  float* pq = hq + s.Hd;          // [Hd] processed query (Bahdanau)
  float* sc = pq + s.Hd;          // [Tm] scores -> probabilities
  float* red = sc + s.Tm;         // [8]

  const int b = blockIdx.x;
  const int part = blockIdx.y, nparts = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hd = s.Hd, M = s.M, Tm = s.Tm;
  const int len = min(s.mem_len[b], Tm);
  const bool writer = (part == 0);

  // ---- LSTM cell (Appendix A.1) ----
  const int tok = s.tok_rows ? s.tok_ids[(int64_t)b * s.tok_stride] : 0;
  for (int u = tid; u < Hd; u += 256) {
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      z[g] = s.z[(int64_t)b * 4 * Hd + g * Hd + u] + s.bias[g * Hd + u];
      if (s.tok_rows) z[g] += las_bf2f(s.tok_rows[(int64_t)tok * 4 * Hd + g * Hd + u]);
    }
    const float gi = las_sigmoid(z[0]), gj = las_tanh(z[1]), gf = las_sigmoid(z[2] + 1.0f), go = las_sigmoid(z[3]);
    const float cn = gf * s.c_prev[(int64_t)b * s.ldcp + u] + gi * gj;
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
  __syncthreads();

  // ---- processed query (Bahdanau): pq[a] = sum_u Wq[u][a] h[u]  (TF Dense kernel layout [in,out]) ----
  if (s.attention == LAS_ATT_BAHDANAU) {
    for (int a = tid; a < Hd; a += 256) {
      float acc = 0.f;
      for (int u = 0; u < Hd; ++u) acc += las_bf2f(s.wq[(int64_t)u * Hd + a]) * hq[u];
      pq[a] = acc;
      if (writer && s.pq_out) s.pq_out[(int64_t)b * s.ldpq + a] = acc;
    }
    __syncthreads();
  }

  // ---- scores: 16 lanes per memory frame, 16 frames per pass ----
  const unsigned short* keys = s.keys + (int64_t)b * Tm * Hd;
  const int sub = lane & 15, grp = lane >> 4;
  for (int t0 = 0; t0 < Tm; t0 += 16) {
    const int t = t0 + wave * 4 + grp;
    float part_sum = 0.f;
    if (t < len) {
      for (int k = sub * 8; k < Hd; k += 128) {
        const uint4 kv = *reinterpret_cast<const uint4*>(keys + (int64_t)t * Hd + k);
        if (s.attention == LAS_ATT_LUONG) {
          part_sum += dot8(kv, hq + k);
        } else {
          const unsigned short* e = reinterpret_cast<const unsigned short*>(&kv);
#pragma unroll
          for (int j = 0; j < 8; ++j) part_sum += s.att_v[k + j] * las_tanh(las_bf2f(e[j]) + pq[k + j]);
        }
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) part_sum += __shfl_xor(part_sum, o, 64);
    if (sub == 0 && t < Tm) sc[t] = (t < len) ? part_sum : -INFINITY;
  }
  __syncthreads();

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

  // ---- context = sum_t' p[t'] * values[b,t',:]  (this workgroup's column range) ----
  const unsigned short* vals = s.values + (int64_t)b * Tm * M;
  const int cols_per = ((M / 4 + nparts - 1) / nparts) * 4;
  const int c_begin = part * cols_per, c_end = min(M, c_begin + cols_per);
  for (int m = c_begin + tid * 4; m < c_end; m += 1024) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int t = 0; t < len; ++t) {
      const uint2 v = *reinterpret_cast<const uint2*>(vals + (int64_t)t * M + m);
      const float p = sc[t];
      a0 += p * __uint_as_float(v.x << 16);
      a1 += p * __uint_as_float(v.x & 0xffff0000u);
      a2 += p * __uint_as_float(v.y << 16);
      a3 += p * __uint_as_float(v.y & 0xffff0000u);
    }
    uint2 o;
    o.x = (unsigned)las_f2bf(a0) | ((unsigned)las_f2bf(a1) << 16);
    o.y = (unsigned)las_f2bf(a2) | ((unsigned)las_f2bf(a3) << 16);
    *reinterpret_cast<uint2*>(s.ctx_out + (int64_t)b * s.ldc + m) = o;
    if (s.ctx_out2) *reinterpret_cast<uint2*>(s.ctx_out2 + (int64_t)b * s.ldc2 + m) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_step_bwd_kernel(las_dec_step_bwd s) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* dctx = sm;               // [M]
  float* ds = dctx + s.M;         // [Tm] dalign -> dscore
  float* dhs = ds + s.Tm;         // [4][Hd] per-wave partial d h (score path) / dpq
  float* red = dhs + 4 * s.Hd;    // [8]

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hd = s.Hd, M = s.M, Tm = s.Tm;
  const int len = min(s.mem_len[b], Tm);

  // total gradient w.r.t. the context of this step; keep a bf16 copy for the d(memory) batched GEMM
  for (int m = tid; m < M; m += 256) {
    float v = s.dctx_a ? s.dctx_a[(int64_t)b * s.ldda + m] : 0.f;
    if (s.dctx_b) v += s.dctx_b[(int64_t)b * s.lddb + m];
    dctx[m] = v;
    if (s.dctx_save) s.dctx_save[(int64_t)b * s.ldds + m] = las_f2bf(v);
  }
  __syncthreads();

  // dalign[t'] = values[b,t',:] . dctx
  const unsigned short* vals = s.values + (int64_t)b * Tm * M;
  const int sub = lane & 15, grp = lane >> 4;
  for (int t0 = 0; t0 < Tm; t0 += 16) {
    const int t = t0 + wave * 4 + grp;
    float acc = 0.f;
    if (t < len) {
      for (int k = sub * 8; k < M; k += 128) {
        const uint4 kv = *reinterpret_cast<const uint4*>(vals + (int64_t)t * M + k);
        acc += dot8(kv, dctx + k);
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0 && t < Tm) ds[t] = (t < len) ? acc : 0.f;
  }
  __syncthreads();

  // softmax backward: ds = p * (dalign - sum p*dalign)
  const float* align = s.align + (int64_t)b * s.lda;
  float dot = 0.f;
  for (int t = tid; t < len; t += 256) dot += align[t] * ds[t];
  dot = block_reduce(dot, red, false);
  for (int t = tid; t < Tm; t += 256) {
    const float v = (t < len) ? align[t] * (ds[t] - dot) : 0.f;
    ds[t] = v;
    if (s.ds_out) s.ds_out[(int64_t)b * s.ldso + t] = las_f2bf(v);
  }
  __syncthreads();

  // gradient into the query path: wave w takes frames t' = w, w+4, ...; lane takes 4 columns
  const unsigned short* keys = s.keys + (int64_t)b * Tm * Hd;
  for (int u = lane * 4; u < Hd; u += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (s.attention == LAS_ATT_LUONG) {
      for (int t = wave; t < len; t += 4) {
        const uint2 v = *reinterpret_cast<const uint2*>(keys + (int64_t)t * Hd + u);
        const float d = ds[t];
        a0 += d * __uint_as_float(v.x << 16);
        a1 += d * __uint_as_float(v.x & 0xffff0000u);
        a2 += d * __uint_as_float(v.y << 16);
        a3 += d * __uint_as_float(v.y & 0xffff0000u);
      }
    } else {
      // Bahdanau: score = sum_a v[a] tanh(keys[t',a] + pq[a]); d_pre = ds * v * (1 - tanh^2)
      const float* pq = s.pq + (int64_t)b * s.ldpq;
      float dv0 = 0.f, dv1 = 0.f, dv2 = 0.f, dv3 = 0.f;
      const float v0 = s.att_v[u], v1 = s.att_v[u + 1], v2 = s.att_v[u + 2], v3 = s.att_v[u + 3];
      const float q0 = pq[u], q1 = pq[u + 1], q2 = pq[u + 2], q3 = pq[u + 3];
      for (int t = wave; t < len; t += 4) {
        const uint2 v = *reinterpret_cast<const uint2*>(keys + (int64_t)t * Hd + u);
        const float d = ds[t];
        const float t0 = las_tanh(__uint_as_float(v.x << 16) + q0);
        const float t1 = las_tanh(__uint_as_float(v.x & 0xffff0000u) + q1);
        const float t2 = las_tanh(__uint_as_float(v.y << 16) + q2);
        const float t3 = las_tanh(__uint_as_float(v.y & 0xffff0000u) + q3);
        dv0 += d * t0; dv1 += d * t1; dv2 += d * t2; dv3 += d * t3;
        const float p0 = d * v0 * (1.f - t0 * t0), p1 = d * v1 * (1.f - t1 * t1);
        const float p2 = d * v2 * (1.f - t2 * t2), p3 = d * v3 * (1.f - t3 * t3);
        a0 += p0; a1 += p1; a2 += p2; a3 += p3;
        float* dk = s.dkeys_acc + ((int64_t)b * Tm + t) * Hd + u;   // this workgroup owns utterance b
        dk[0] += p0; dk[1] += p1; dk[2] += p2; dk[3] += p3;
      }
      atomicAdd(s.dv_acc + u, dv0); atomicAdd(s.dv_acc + u + 1, dv1);
      atomicAdd(s.dv_acc + u + 2, dv2); atomicAdd(s.dv_acc + u + 3, dv3);
    }
    float* o = dhs + wave * Hd + u;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
  }
  __syncthreads();
  // reduce the 4 per-wave partials
  for (int u = tid; u < Hd; u += 256) dhs[u] = dhs[u] + dhs[Hd + u] + dhs[2 * Hd + u] + dhs[3 * Hd + u];
  __syncthreads();
  if (s.attention == LAS_ATT_BAHDANAU) {
    // dhs holds d(processed query); save it (bf16) for d(query_layer) and map back: dh[u] = sum_a dpq[a] Wq[u][a]
    float* tmp = dhs + Hd;
    for (int u = tid; u < Hd; u += 256) {
      if (s.dpq_out) s.dpq_out[(int64_t)b * s.lddpq + u] = las_f2bf(dhs[u]);
      float acc = 0.f;
      for (int a = 0; a < Hd; ++a) acc += las_bf2f(s.wq_t[(int64_t)a * Hd + u]) * dhs[a];
      tmp[u] = acc;
    }
    __syncthreads();
    for (int u = tid; u < Hd; u += 256) dhs[u] = tmp[u];
    __syncthreads();
  }

  // ---- LSTM cell backward (Appendix F) ----
  for (int u = tid; u < Hd; u += 256) {
    const float* gp = s.gates + (int64_t)b * s.ldg + u;
    const float gi = gp[0], gj = gp[Hd], gf = gp[2 * Hd], go = gp[3 * Hd];
    const float ct = s.c_new[(int64_t)b * s.ldcn + u];
    const float cp = s.c_prev[(int64_t)b * s.ldcp + u];
    float dht = dhs[u];
    if (s.dh_rec) dht += s.dh_rec[(int64_t)b * s.ldr + u];
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
  if (lane == 0 && local_loss != 0.f) atomicAdd(loss_out, local_loss);
}

}  // namespace

extern "C" int las_decoder_step_fwd(const las_dec_step* s, int parts, void* stream) {
  LAS_REQUIRE(s->B > 0 && s->Hd % 8 == 0 && s->M % 8 == 0 && s->Tm > 0, "las_decoder_step_fwd: bad shape");
  LAS_REQUIRE(s->attention == LAS_ATT_LUONG || (s->wq && s->att_v), "las_decoder_step_fwd: Bahdanau needs wq and att_v");
  if (parts < 1) parts = 1;
  const size_t lds = (size_t)(2 * s->Hd + s->Tm + 8) * sizeof(float);
  LAS_REQUIRE(lds <= 64 * 1024, "las_decoder_step_fwd: memory length %d too long for the LDS score buffer", s->Tm);
  hipLaunchKernelGGL(dec_step_fwd_kernel, dim3(s->B, parts), dim3(256), lds, (hipStream_t)stream, *s);
  LAS_LAUNCH_CHECK("decoder step fwd launch");
  return LAS_OK;
}

extern "C" int las_decoder_step_bwd(const las_dec_step_bwd* s, void* stream) {
  LAS_REQUIRE(s->B > 0 && s->Hd % 8 == 0 && s->M % 8 == 0 && s->Tm > 0, "las_decoder_step_bwd: bad shape");
  LAS_REQUIRE(s->attention == LAS_ATT_LUONG || (s->wq_t && s->att_v && s->pq && s->dkeys_acc && s->dv_acc),
              "las_decoder_step_bwd: Bahdanau needs wq_t, att_v, pq, dkeys_acc, dv_acc");
  const size_t lds = (size_t)(s->M + s->Tm + 4 * s->Hd + 8) * sizeof(float);
  LAS_REQUIRE(lds <= 64 * 1024, "las_decoder_step_bwd: shapes exceed the LDS budget");
  hipLaunchKernelGGL(dec_step_bwd_kernel, dim3(s->B), dim3(256), lds, (hipStream_t)stream, *s);
  LAS_LAUNCH_CHECK("decoder step bwd launch");
  return LAS_OK;
}

extern "C" int las_seq_ce_loss(const float* logits, int64_t ldl, const int32_t* targets, const int32_t* target_len,
                               int B, int U, int V, float grad_scale, float* loss_out, las_bf16* dlogits, int64_t ldd,
                               void* stream) {
  LAS_REQUIRE(B > 0 && U > 0 && V > 0, "las_seq_ce_loss: bad shape");
  int blocks = (B * U + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(seq_ce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, ldl, targets, target_len, B,
                     U, V, grad_scale, loss_out, dlogits, ldd);
  LAS_LAUNCH_CHECK("seq ce launch");
  return LAS_OK;
}
