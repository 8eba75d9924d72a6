// Stochastic pieces of the TRAIN graph, driven by a counter-based generator so that forward and backward
// regenerate the same draws without storing masks:
//   - DropoutWrapper(input_keep_prob) on LSTM cell inputs (las/ops.py:14-18): inverted dropout,
//     an independent Bernoulli(keep) per (utterance, time step, input element);
//   - scheduled sampling of TPUScheduledEmbeddingTrainingHelper (utils/training_helper.py:48-87):
//     per (utterance, step) with probability p the next decoder input is a token drawn from Categorical(logits).
// The draws cannot match TensorFlow's Philox streams; parity tests export the realised masks / tokens and replay
// them through the oracle.
#include "las_common.h"

namespace {

__global__ void dropout_bf16_kernel(const unsigned short* x, int64_t ldx, unsigned short* y, int64_t ldy, int rows, int cols,
                                    float keep, unsigned seed, unsigned stream) {
  const float inv = 1.0f / keep;
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    const float m = las_uniform(seed, stream, (unsigned long long)i) < keep ? inv : 0.f;
    y[(int64_t)r * ldy + c] = las_f2bf(las_bf2f(x[(int64_t)r * ldx + c]) * m);
  }
}

// both directions' masked copies of one input in one pass (x is read once): 8 elements per thread when the rows allow
__global__ void dropout_bf16_pair_kernel(const unsigned short* x, int64_t ldx, unsigned short* y0, unsigned short* y1, int64_t ldy,
                                         int rows, int cols, float keep, unsigned seed, unsigned stream0, unsigned stream1) {
  const float inv = 1.0f / keep;
  const int c8 = cols / 8;
  const int64_t total = (int64_t)rows * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / c8), c = (int)(i % c8) * 8;
    const uint4 v = *reinterpret_cast<const uint4*>(x + (int64_t)r * ldx + c);
    const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
    uint4 o0, o1;
    unsigned short* a = reinterpret_cast<unsigned short*>(&o0);
    unsigned short* b = reinterpret_cast<unsigned short*>(&o1);
    const unsigned long long base = (unsigned long long)r * cols + c;       // (the element index las_dropout_bf16 uses)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = las_bf2f(e[j]);
      a[j] = las_f2bf(f * (las_uniform(seed, stream0, base + j) < keep ? inv : 0.f));
      b[j] = las_f2bf(f * (las_uniform(seed, stream1, base + j) < keep ? inv : 0.f));
    }
    *reinterpret_cast<uint4*>(y0 + (int64_t)r * ldy + c) = o0;
    *reinterpret_cast<uint4*>(y1 + (int64_t)r * ldy + c) = o1;
  }
}

// the input masks of U decoder steps in one pass, in place: element (b, t, c), c < cols, of rows x[b * ldb + t * ldt ..] is
// draw (seed, stream0 + t, b * cols + c) -- U x las_dropout_bf16 with the streams stream0 + t (the one-launch decoders leave
// the operand rows undropped; the weight-gradient products read them dropped)
__global__ void dropout_bf16_steps_kernel(unsigned short* x, int64_t ldb, int64_t ldt, int B, int U, int cols, int idx_cols, float keep,
                                          unsigned seed, unsigned stream0) {
  const float inv = 1.0f / keep;
  const int c8 = cols / 8;
  const int64_t total = (int64_t)B * U * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8, t = (int)((i / c8) % U), b = (int)(i / ((int64_t)c8 * U));
    unsigned short* row = x + (int64_t)b * ldb + (int64_t)t * ldt + c;
    uint4 v = *reinterpret_cast<const uint4*>(row);
    unsigned short* e = reinterpret_cast<unsigned short*>(&v);
    const unsigned long long base = (unsigned long long)b * idx_cols + c;
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = las_f2bf(las_bf2f(e[j]) * (las_uniform(seed, stream0 + (unsigned)t, base + j) < keep ? inv : 0.f));
    *reinterpret_cast<uint4*>(row) = v;
  }
}

// out = a * mask_a (+ b * mask_b): gradient through the input dropout of the fw (and bw) cell
__global__ void dropout_bwd_kernel(const float* a, const float* b, float* out, int rows, int cols, float keep, unsigned seed,
                                   unsigned stream_a, unsigned stream_b) {
  const float inv = 1.0f / keep;
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float v = a[i] * (las_uniform(seed, stream_a, (unsigned long long)i) < keep ? inv : 0.f);
    if (b) v += b[i] * (las_uniform(seed, stream_b, (unsigned long long)i) < keep ? inv : 0.f);
    out[i] = v;
  }
}

__global__ void dropout_mask_kernel(float* out, int64_t total, float keep, unsigned seed, unsigned stream) {
  const float inv = 1.0f / keep;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = las_uniform(seed, stream, (unsigned long long)i) < keep ? inv : 0.f;
}

// onehot[row, ids[row]] = scale(row) for the token-row gradient GEMM; scale = dropout factor of that one-hot entry
__global__ void onehot_kernel(const int32_t* ids, int64_t id_stride_b, int B, int U, int V, unsigned short* out, int64_t ldo,
                              float keep, unsigned seed, unsigned stream, int feed_width) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;     // row = b*U + t
  if (row >= B * U) return;
  const int b = row / U, t = row % U;
  const int id = ids[(int64_t)b * id_stride_b + t];
  float sc = 1.0f;
  if (keep < 1.0f) {
    const unsigned long long idx = ((unsigned long long)t * B + b) * feed_width + id;
    sc = las_uniform(seed, stream, idx) < keep ? 1.0f / keep : 0.f;
  }
  if (id >= 0 && id < V) out[(int64_t)row * ldo + id] = las_f2bf(sc);
}

// scheduled sampling: one wave per utterance.  next[b] = Categorical(logits[b,:V]) with probability p, else teacher[b]
__global__ void sample_kernel(const float* logits, int64_t ldl, int V, const int32_t* teacher, int64_t teacher_stride,
                              int32_t* next, int64_t next_stride, int B, float prob, unsigned seed, unsigned step) {
  const int b = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  const bool select = las_uniform(seed, 0x5e1ec7u, (unsigned long long)step * B + b) < prob;
  int out = teacher ? teacher[(int64_t)b * teacher_stride] : 0;
  if (select) {
    // Gumbel-max: argmax_v logits[v] - log(-log(u_v))
    float best = -INFINITY;
    int arg = 0;
    for (int v = lane; v < V; v += 64) {
      const float u = fmaxf(las_uniform(seed, 0x9a3b1eu, ((unsigned long long)step * B + b) * V + v), 1e-12f);
      const float g = logits[(int64_t)b * ldl + v] - __logf(-__logf(u));
      if (g > best) { best = g; arg = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oa = __shfl_xor(arg, o, 64);
      if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    out = arg;
  }
  if (lane == 0) next[(int64_t)b * next_stride] = out;
}

// p[i] += std * N(0,1)  (Box-Muller on two counter-based uniforms): weight noise of model_helper.py:418-432
__global__ void add_noise_kernel(float* p, int64_t n, float std, unsigned seed, unsigned stream) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    p[i] += std * las_normal(seed, stream, (unsigned long long)i);
  }
}

__global__ void normal_fill_kernel(float* out, int64_t n, unsigned seed, unsigned stream) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = las_normal(seed, stream, (unsigned long long)i);
}

__global__ void relu_bf16_kernel(unsigned short* x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (x[i] & 0x8000u) x[i] = 0;          // negative (or -0): clamp to +0
}

__global__ void relu_bwd_kernel(float* d, const unsigned short* y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (!(las_bf2f(y[i]) > 0.f)) d[i] = 0.f;
}

__global__ void add_masked_kernel(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int rows,
                                  int cols, float keep, unsigned seed, unsigned stream, unsigned long long idx_base, int64_t idx_ld) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    float v = b[(int64_t)r * ldb + c];
    if (keep < 1.0f) v = las_uniform(seed, stream, idx_base + (unsigned long long)r * idx_ld + c) < keep ? v / keep : 0.f;
    if (a) v += a[(int64_t)r * lda + c];
    out[(int64_t)r * ldo + c] = v;
  }
}

int blocks_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 2048 ? 2048 : b);
}

}  // namespace

extern "C" int las_dropout_bf16(const las_bf16* x, int64_t ldx, las_bf16* y, int64_t ldy, int rows, int cols, float keep,
                                uint32_t seed, uint32_t stream_id, void* stream) {
  LAS_REQUIRE(rows > 0 && cols > 0 && keep > 0.f && keep <= 1.f, "las_dropout_bf16: bad arguments");
  hipLaunchKernelGGL(dropout_bf16_kernel, dim3(blocks_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, x, ldx, y,
                     ldy, rows, cols, keep, seed, stream_id);
  LAS_LAUNCH_CHECK("dropout launch");
  return LAS_OK;
}

extern "C" int las_dropout_bf16_pair(const las_bf16* x, int64_t ldx, las_bf16* y0, las_bf16* y1, int64_t ldy, int rows, int cols,
                                     float keep, uint32_t seed, uint32_t stream0, uint32_t stream1, void* stream) {
  LAS_REQUIRE(rows > 0 && cols > 0 && keep > 0.f && keep <= 1.f, "las_dropout_bf16_pair: bad arguments");
  LAS_REQUIRE(cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y0 % 16 == 0) && ((uintptr_t)y1 % 16 == 0),
              "las_dropout_bf16_pair: columns and row strides in multiples of 8, 16-byte aligned buffers");
  hipLaunchKernelGGL(dropout_bf16_pair_kernel, dim3(blocks_for((int64_t)rows * (cols / 8))), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     y0, y1, ldy, rows, cols, keep, seed, stream0, stream1);
  LAS_LAUNCH_CHECK("dropout pair launch");
  return LAS_OK;
}

extern "C" int las_dropout_bf16_steps(las_bf16* x, int64_t ldb, int64_t ldt, int B, int U, int cols, int idx_cols, float keep,
                                      uint32_t seed, uint32_t stream0, void* stream) {
  if (idx_cols <= 0) idx_cols = cols;
  LAS_REQUIRE(x && B > 0 && U > 0 && cols > 0 && idx_cols >= cols && keep > 0.f && keep <= 1.f, "las_dropout_bf16_steps: bad arguments");
  LAS_REQUIRE(cols % 8 == 0 && ldb % 8 == 0 && ldt % 8 == 0 && ((uintptr_t)x % 16 == 0),
              "las_dropout_bf16_steps: columns and strides in multiples of 8, 16-byte aligned rows");
  hipLaunchKernelGGL(dropout_bf16_steps_kernel, dim3(blocks_for((int64_t)B * U * (cols / 8))), dim3(256), 0, (hipStream_t)stream, x, ldb,
                     ldt, B, U, cols, idx_cols, keep, seed, stream0);
  LAS_LAUNCH_CHECK("dropout steps launch");
  return LAS_OK;
}

extern "C" int las_dropout_bwd(const float* a, const float* b, float* out, int rows, int cols, float keep, uint32_t seed,
                               uint32_t stream_a, uint32_t stream_b, void* stream) {
  LAS_REQUIRE(rows > 0 && cols > 0 && keep > 0.f && keep <= 1.f, "las_dropout_bwd: bad arguments");
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(blocks_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, a, b, out,
                     rows, cols, keep, seed, stream_a, stream_b);
  LAS_LAUNCH_CHECK("dropout bwd launch");
  return LAS_OK;
}

extern "C" int las_dropout_mask(float* out, int64_t total, float keep, uint32_t seed, uint32_t stream_id, void* stream) {
  LAS_REQUIRE(total > 0 && keep > 0.f && keep <= 1.f, "las_dropout_mask: bad arguments");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, out, total, keep, seed,
                     stream_id);
  LAS_LAUNCH_CHECK("dropout mask launch");
  return LAS_OK;
}

extern "C" int las_onehot_bf16(const int32_t* ids, int64_t id_stride_b, int B, int U, int V, las_bf16* out, int64_t ldo,
                               float keep, uint32_t seed, uint32_t stream_id, int feed_width, void* stream) {
  LAS_REQUIRE(B > 0 && U > 0 && V > 0 && ldo >= V, "las_onehot_bf16: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int rc = las_check_hip(hipMemsetAsync(out, 0, (size_t)B * U * ldo * sizeof(las_bf16), st), "memset onehot");
  if (rc) return rc;
  hipLaunchKernelGGL(onehot_kernel, dim3((B * U + 255) / 256), dim3(256), 0, st, ids, id_stride_b, B, U, V, out, ldo, keep,
                     seed, stream_id, feed_width);
  LAS_LAUNCH_CHECK("onehot launch");
  return LAS_OK;
}

extern "C" int las_sample_tokens(const float* logits, int64_t ldl, int V, const int32_t* teacher, int64_t teacher_stride,
                                 int32_t* next, int64_t next_stride, int B, float prob, uint32_t seed, uint32_t step,
                                 void* stream) {
  LAS_REQUIRE(B > 0 && V > 0 && prob >= 0.f && prob <= 1.f, "las_sample_tokens: bad arguments");
  hipLaunchKernelGGL(sample_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, teacher, teacher_stride,
                     next, next_stride, B, prob, seed, step);
  LAS_LAUNCH_CHECK("sample launch");
  return LAS_OK;
}

extern "C" int las_normal_fill(float* out, int64_t n, uint32_t seed, uint32_t stream_id, void* stream) {
  LAS_REQUIRE(n > 0 && out, "las_normal_fill: bad arguments");
  hipLaunchKernelGGL(normal_fill_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, out, n, seed, stream_id);
  LAS_LAUNCH_CHECK("normal fill launch");
  return LAS_OK;
}

extern "C" int las_relu_bf16(las_bf16* x, int64_t n, void* stream) {
  LAS_REQUIRE(n > 0 && x, "las_relu_bf16: bad arguments");
  hipLaunchKernelGGL(relu_bf16_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, n);
  LAS_LAUNCH_CHECK("relu launch");
  return LAS_OK;
}

extern "C" int las_relu_bwd(float* d, const las_bf16* y, int64_t n, void* stream) {
  LAS_REQUIRE(n > 0 && d && y, "las_relu_bwd: bad arguments");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, d, y, n);
  LAS_LAUNCH_CHECK("relu bwd launch");
  return LAS_OK;
}

extern "C" int las_add_noise(float* p, int64_t n, float std, uint32_t seed, uint32_t stream_id, void* stream) {
  LAS_REQUIRE(n > 0 && std >= 0.f, "las_add_noise: bad arguments");
  hipLaunchKernelGGL(add_noise_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, p, n, std, seed, stream_id);
  LAS_LAUNCH_CHECK("add noise launch");
  return LAS_OK;
}

extern "C" int las_add_masked(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int rows,
                              int cols, float keep, uint32_t seed, uint32_t stream_id, uint64_t idx_base, int64_t idx_ld,
                              void* stream) {
  LAS_REQUIRE(rows > 0 && cols > 0 && b != nullptr && keep > 0.f, "las_add_masked: bad arguments");
  hipLaunchKernelGGL(add_masked_kernel, dim3(blocks_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
                     out, ldo, rows, cols, keep, seed, stream_id, (unsigned long long)idx_base, idx_ld);
  LAS_LAUNCH_CHECK("add masked launch");
  return LAS_OK;
}
