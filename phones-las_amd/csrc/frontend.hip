// Acoustic front-end kernels (fp32): framing + window + DFT, filterbank / DCT products, dB compression, RMS energy
// and Savitzky-Golay deltas — the device side of preprocess_all.py:69-130 (librosa semantics) and of
// utils/features_utils.py:5-20 (tf.contrib.signal semantics).  The transforms are table driven (window, DFT
// twiddles, mel basis, DCT basis, Savitzky-Golay taps are built on the host in float64 and passed as fp32), so one
// set of kernels serves both pipelines.  This is offline, HBM-light work: one workgroup per frame / row, coalesced
// table reads, no attempt to reach MFMA (log-mel needs more than bf16 operands).
#include "las_common.h"

namespace {

// out[f, k] = | sum_n w[n] x[f*hop + n - pad] e^{-2 pi i k n / n_fft} | ^ power      (power = 1 or 2)
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ wave, int N, int n_fft, int hop, int pad,
                                                   const float* __restrict__ window, const float* __restrict__ costab,
                                                   const float* __restrict__ sintab, int bins, int power,
                                                   float* __restrict__ out, int64_t ldo) {
  extern __shared__ float fr[];
  const int f = blockIdx.x;
  for (int n = threadIdx.x; n < n_fft; n += 256) {
    int i = f * hop + n - pad;
    if (pad > 0) {                       // np.pad(mode='reflect')
      if (i < 0) i = -i;
      if (i >= N) i = 2 * (N - 1) - i;
    }
    fr[n] = (i >= 0 && i < N) ? wave[i] * window[n] : 0.f;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < bins; k += 256) {
    float re = 0.f, im = 0.f;
    for (int n = 0; n < n_fft; ++n) {
      const float v = fr[n];
      re = __builtin_fmaf(v, costab[(int64_t)n * bins + k], re);
      im = __builtin_fmaf(v, sintab[(int64_t)n * bins + k], im);
    }
    const float p2 = __builtin_fmaf(im, im, re * re);      // (spelled out: the kernels of this file must round alike)
    out[(int64_t)f * ldo + k] = power == 2 ? p2 : sqrtf(p2);
  }
}

// C[m, n] = epi(sum_k A[m, k] W[k, n]);  epi 0: id, 1: log(x + eps), 2: 10 log10(max(x, eps))
__global__ __launch_bounds__(256) void matmul_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                     int64_t ldw, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                     int epi, float eps) {
  extern __shared__ float row[];
  const int m = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += 256) row[k] = A[(int64_t)m * lda + k];
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += 256) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = __builtin_fmaf(row[k], W[(int64_t)k * ldw + n], acc);
    if (epi == 1) acc = logf(acc + eps);
    else if (epi == 2) acc = 10.0f * log10f(fmaxf(acc, eps));
    C[(int64_t)m * ldc + n] = acc;
  }
}

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  // order-preserving integer trick (works for mixed signs)
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

__global__ void max_kernel(const float* x, int64_t ldx, int rows, int cols, float* out) {
  float m = -INFINITY;
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, x[(i / cols) * ldx + (i % cols)]);
  m = las_wave_max(m);
  if ((threadIdx.x & 63) == 0) atomic_max_float(out, m);
}

__global__ void floor_kernel(float* x, int64_t ldx, int rows, int cols, const float* mx, float top_db) {
  const float lo = *mx - top_db;
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float* p = x + (i / cols) * ldx + (i % cols);
    *p = fmaxf(*p, lo);
  }
}

// librosa.feature.rms(center=True): sqrt(mean(frame^2)), reflect padded
__global__ __launch_bounds__(256) void rms_kernel(const float* wave, int N, int frame_length, int hop, int pad, float* out,
                                                  int64_t ldo) {
  __shared__ float red[4];
  const int f = blockIdx.x;
  float acc = 0.f;
  for (int n = threadIdx.x; n < frame_length; n += 256) {
    int i = f * hop + n - pad;
    if (i < 0) i = -i;
    if (i >= N) i = 2 * (N - 1) - i;
    const float v = (i >= 0 && i < N) ? wave[i] : 0.f;
    acc += v * v;
  }
  acc = las_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[(int64_t)f * ldo] = sqrtf((red[0] + red[1] + red[2] + red[3]) / frame_length);
}

// Savitzky-Golay derivative along time (scipy.signal.savgol_filter(mode='interp')): interior = correlation with
// `taps[width]`; the first/last `half` rows use the edge matrices [half, width] applied to the first/last window.
__global__ void delta_kernel(const float* x, int64_t ldx, int T, int F, const float* taps, const float* edge_lo,
                             const float* edge_hi, int width, float* out, int64_t ldo, int out_stride, int out_offset) {
  const int half = width / 2;
  const int64_t total = (int64_t)T * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i / F), c = (int)(i % F);
    float acc = 0.f;
    if (t < half) {
      for (int j = 0; j < width; ++j) acc += edge_lo[t * width + j] * x[(int64_t)j * ldx + c];
    } else if (t >= T - half) {
      const int r = t - (T - half);
      for (int j = 0; j < width; ++j) acc += edge_hi[r * width + j] * x[(int64_t)(T - width + j) * ldx + c];
    } else {
      for (int j = 0; j < width; ++j) acc += taps[j] * x[(int64_t)(t - half + j) * ldx + c];
    }
    out[(int64_t)t * ldo + c * out_stride + out_offset] = acc;
  }
}

__global__ void copy_strided_kernel(const float* x, int64_t ldx, int T, int F, float* out, int64_t ldo, int out_stride,
                                    int out_offset) {
  const int64_t total = (int64_t)T * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    out[(i / F) * ldo + (i % F) * out_stride + out_offset] = x[(i / F) * ldx + (i % F)];
}

__global__ void fill_f32_kernel(float* x, int n, float v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}

int nblocks(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b)); }

// ------------------------------------------------------------------------------------------------
// The same pipeline over a BATCH of utterances in three launches (round 3): the waveforms lie back to back in one buffer
// (wave_off[u] = first sample of utterance u), their frames back to back in the outputs (frame_off[u] = first frame).
// Every stage is one workgroup per frame (or one thread per output element) over ALL frames of the batch; what is
// per-utterance in the reference -- reflect padding at the signal's ends, the top_db floor under the utterance's own
// maximum, the Savitzky-Golay edge windows -- is resolved through the offset tables.  The arithmetic of a frame is the
// arithmetic of the per-utterance kernels above, in the same order: the batched features are bit-identical to them.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_utt(const int32_t* frame_off, int n_utt, int f) {
  int lo = 0, hi = n_utt - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (frame_off[mid] <= f) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// frame -> windowed DFT -> |.|^power -> filterbank [bins, n_mels] -> epilogue (as matmul_kernel: 0 id, 1 log(x + eps),
// 2 10 log10(max(x, eps))) * scale; utt_max[u] (nullable) receives the utterance's maximum (for the top_db floor).
// LDS: n_fft + bins floats.
__global__ __launch_bounds__(256) void batch_melspec_kernel(const float* __restrict__ waves, const int64_t* __restrict__ wave_off,
                                                            const int32_t* __restrict__ frame_off, int n_utt, int n_fft, int hop,
                                                            int pad, int power, const float* __restrict__ window,
                                                            const float* __restrict__ costab, const float* __restrict__ sintab,
                                                            int bins, const float* __restrict__ mel, int n_mels, int epi, float eps,
                                                            float scale, float* __restrict__ out, float* __restrict__ utt_max) {
  extern __shared__ float fr[];
  float* spec = fr + n_fft;
  __shared__ int s_utt;
  const int f = blockIdx.x;
  if (threadIdx.x == 0) s_utt = find_utt(frame_off, n_utt, f);
  __syncthreads();
  const int u = s_utt;
  const float* wave = waves + wave_off[u];
  const int N = (int)(wave_off[u + 1] - wave_off[u]);
  const int fl = f - frame_off[u];
  for (int n = threadIdx.x; n < n_fft; n += 256) {
    int i = fl * hop + n - pad;
    if (pad > 0) {
      if (i < 0) i = -i;
      if (i >= N) i = 2 * (N - 1) - i;
    }
    fr[n] = (i >= 0 && i < N) ? wave[i] * window[n] : 0.f;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < bins; k += 256) {
    float re = 0.f, im = 0.f;
    for (int n = 0; n < n_fft; ++n) {
      const float v = fr[n];
      re = __builtin_fmaf(v, costab[(int64_t)n * bins + k], re);
      im = __builtin_fmaf(v, sintab[(int64_t)n * bins + k], im);
    }
    const float p2 = __builtin_fmaf(im, im, re * re);      // (spelled out: the kernels of this file must round alike)
    spec[k] = power == 2 ? p2 : sqrtf(p2);
  }
  __syncthreads();
  float m = -INFINITY;
  for (int n = threadIdx.x; n < n_mels; n += 256) {
    float acc = 0.f;
    for (int k = 0; k < bins; ++k) acc = __builtin_fmaf(spec[k], mel[(int64_t)k * n_mels + n], acc);
    if (epi == 1) acc = logf(acc + eps);
    else if (epi == 2) acc = 10.0f * log10f(fmaxf(acc, eps));
    acc *= scale;
    out[(int64_t)f * n_mels + n] = acc;
    m = fmaxf(m, acc);
  }
  if (utt_max) {
    m = las_wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > -INFINITY) atomic_max_float(utt_max + u, m);
  }
}

// The same stage with FB frames per workgroup (round 4): the DFT tables (2 x n_fft x bins floats = 412 KB at 20-ms windows) were
// read from L2 once per FRAME -- 84 GB of L2 traffic for 256 utterances of 8 s, the whole 4 ms of the stage; here a table element
// is loaded once and used for FB frames held in LDS (frame-minor layout: the FB samples of one n are two 16-byte reads).  Every
// (frame, bin) sum runs over n in the same order with the same fused multiply-adds as above, and every mel sum over k likewise:
// the results are bit-identical to the one-frame kernels (tests/test_gpu_frontend.py compares with torch.equal).
// LDS: FB * (n_fft + bins) floats + FB maxima.
template <int FB>
__global__ __launch_bounds__(256) void batch_melspec_blocked_kernel(const float* __restrict__ waves, const int64_t* __restrict__ wave_off,
                                                                    const int32_t* __restrict__ frame_off, int n_utt, int total_frames,
                                                                    int n_fft, int hop, int pad, int power, const float* __restrict__ window,
                                                                    const float* __restrict__ costab, const float* __restrict__ sintab,
                                                                    int bins, const float* __restrict__ mel, int n_mels, int epi, float eps,
                                                                    float scale, float* __restrict__ out, float* __restrict__ utt_max) {
  static_assert(FB % 4 == 0, "16-byte LDS reads of the frames' samples");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* frT = sm;                         // [n_fft][FB]
  float* spec = sm + (size_t)FB * n_fft;   // [FB][bins]
  __shared__ int s_utt[FB];
  __shared__ float s_max[FB];
  const int tid = threadIdx.x, f0 = blockIdx.x * FB;
  if (tid < FB) {
    s_utt[tid] = f0 + tid < total_frames ? find_utt(frame_off, n_utt, f0 + tid) : -1;
    s_max[tid] = -INFINITY;
  }
  __syncthreads();
  for (int e = tid; e < FB * n_fft; e += 256) {
    const int j = e / n_fft, n = e - j * n_fft, u = s_utt[j];
    float v = 0.f;
    if (u >= 0) {
      const float* wave = waves + wave_off[u];
      const int N = (int)(wave_off[u + 1] - wave_off[u]);
      int i = (f0 + j - frame_off[u]) * hop + n - pad;
      if (pad > 0) {
        if (i < 0) i = -i;
        if (i >= N) i = 2 * (N - 1) - i;
      }
      v = (i >= 0 && i < N) ? wave[i] * window[n] : 0.f;
    }
    frT[n * FB + j] = v;
  }
  __syncthreads();
  for (int k = tid; k < bins; k += 256) {
    float re[FB], im[FB];
#pragma unroll
    for (int j = 0; j < FB; ++j) re[j] = im[j] = 0.f;
    for (int n = 0; n < n_fft; ++n) {
      const float c = costab[(int64_t)n * bins + k], sn = sintab[(int64_t)n * bins + k];
      float v[FB];
#pragma unroll
      for (int q = 0; q < FB / 4; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(frT + n * FB + 4 * q);
        v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
      }
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        re[j] = __builtin_fmaf(v[j], c, re[j]);
        im[j] = __builtin_fmaf(v[j], sn, im[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < FB; ++j) {
      const float p2 = __builtin_fmaf(im[j], im[j], re[j] * re[j]);
      spec[j * bins + k] = power == 2 ? p2 : sqrtf(p2);
    }
  }
  __syncthreads();
  for (int e = tid; e < FB * n_mels; e += 256) {
    const int j = e / n_mels, n = e - j * n_mels;
    if (s_utt[j] < 0) continue;
    const float* sp = spec + j * bins;
    float acc = 0.f;
    for (int k = 0; k < bins; ++k) acc = __builtin_fmaf(sp[k], mel[(int64_t)k * n_mels + n], acc);
    if (epi == 1) acc = logf(acc + eps);
    else if (epi == 2) acc = 10.0f * log10f(fmaxf(acc, eps));
    acc *= scale;
    out[(int64_t)(f0 + j) * n_mels + n] = acc;
    if (utt_max) atomic_max_float(&s_max[j], acc);            // (LDS; the order of a maximum does not matter)
  }
  if (utt_max) {
    __syncthreads();
    if (tid < FB && s_utt[tid] >= 0 && s_max[tid] > -INFINITY) atomic_max_float(utt_max + s_utt[tid], s_max[tid]);
  }
}

// mel_db row -> floor at utt_max[u] - top_db (utt_max nullable: no floor) -> DCT [n_mels, n_out] (nullable: the row itself)
// -> out[f, 0..n_out); energy != 0: out[f, n_out] = RMS of the frame's samples (reflect padded), as rms_kernel.
// LDS: n_mels floats.
__global__ __launch_bounds__(256) void batch_finish_kernel(const float* __restrict__ mel_db, int n_mels,
                                                           const int32_t* __restrict__ frame_off, int n_utt,
                                                           const float* __restrict__ utt_max, float top_db,
                                                           const float* __restrict__ dct, int n_out,
                                                           const float* __restrict__ waves, const int64_t* __restrict__ wave_off,
                                                           int frame_length, int hop, int energy, float* __restrict__ out, int64_t ldo) {
  extern __shared__ float row[];
  __shared__ float red[4];
  __shared__ int s_utt;
  const int f = blockIdx.x;
  if (threadIdx.x == 0) s_utt = find_utt(frame_off, n_utt, f);
  __syncthreads();
  const int u = s_utt;
  const float lo = utt_max ? utt_max[u] - top_db : -INFINITY;
  for (int k = threadIdx.x; k < n_mels; k += 256) row[k] = fmaxf(mel_db[(int64_t)f * n_mels + k], lo);
  __syncthreads();
  for (int n = threadIdx.x; n < n_out; n += 256) {
    float acc;
    if (dct) {
      acc = 0.f;
      for (int k = 0; k < n_mels; ++k) acc = __builtin_fmaf(row[k], dct[(int64_t)k * n_out + n], acc);
    } else {
      acc = row[n];
    }
    out[(int64_t)f * ldo + n] = acc;
  }
  if (energy) {
    const float* wave = waves + wave_off[u];
    const int N = (int)(wave_off[u + 1] - wave_off[u]);
    const int fl = f - frame_off[u], pad = frame_length / 2;
    float acc = 0.f;
    for (int n = threadIdx.x; n < frame_length; n += 256) {
      int i = fl * hop + n - pad;
      if (i < 0) i = -i;
      if (i >= N) i = 2 * (N - 1) - i;
      const float v = (i >= 0 && i < N) ? wave[i] : 0.f;
      acc += v * v;
    }
    acc = las_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[(int64_t)f * ldo + n_out] = sqrtf((red[0] + red[1] + red[2] + red[3]) / frame_length);
  }
}

// [c, delta(c), delta-delta(c)] interleaved per feature (preprocess_all.py:120-129) with every utterance's own edge windows
__global__ void batch_delta_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ frame_off, int n_utt,
                                   int total_frames, int F, const float* __restrict__ taps1, const float* __restrict__ lo1,
                                   const float* __restrict__ hi1, const float* __restrict__ taps2, const float* __restrict__ lo2,
                                   const float* __restrict__ hi2, int width, float* __restrict__ out, int64_t ldo) {
  const int half = width / 2;
  const int64_t total = (int64_t)total_frames * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i / F), c = (int)(i % F);
    const int u = find_utt(frame_off, n_utt, f);
    const int f0 = frame_off[u], T = frame_off[u + 1] - f0, t = f - f0;
    const float* xu = x + (int64_t)f0 * ldx;
    float a1 = 0.f, a2 = 0.f;
    if (t < half) {
      for (int j = 0; j < width; ++j) { const float v = xu[(int64_t)j * ldx + c]; a1 += lo1[t * width + j] * v; a2 += lo2[t * width + j] * v; }
    } else if (t >= T - half) {
      const int r = t - (T - half);
      for (int j = 0; j < width; ++j) { const float v = xu[(int64_t)(T - width + j) * ldx + c]; a1 += hi1[r * width + j] * v; a2 += hi2[r * width + j] * v; }
    } else {
      for (int j = 0; j < width; ++j) { const float v = xu[(int64_t)(t - half + j) * ldx + c]; a1 += taps1[j] * v; a2 += taps2[j] * v; }
    }
    float* o = out + (int64_t)f * ldo + c * 3;
    o[0] = x[(int64_t)f * ldx + c];
    o[1] = a1;
    o[2] = a2;
  }
}

}  // namespace

extern "C" int las_fe_stft(const float* wave, int num_samples, int n_fft, int hop, int center, int power,
                           const float* window, const float* costab, const float* sintab, int bins, float* out, int64_t ldo,
                           int frames, void* stream) {
  LAS_REQUIRE(num_samples > 0 && n_fft > 0 && hop > 0 && frames > 0 && bins > 0 && (power == 1 || power == 2) && n_fft <= 8192,
              "las_fe_stft: bad arguments");
  hipLaunchKernelGGL(stft_kernel, dim3(frames), dim3(256), n_fft * sizeof(float), (hipStream_t)stream, wave, num_samples, n_fft,
                     hop, center ? n_fft / 2 : 0, window, costab, sintab, bins, power, out, ldo);
  LAS_LAUNCH_CHECK("stft launch");
  return LAS_OK;
}

extern "C" int las_fe_matmul(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int M, int N,
                             int K, int epilogue, float eps, void* stream) {
  LAS_REQUIRE(M > 0 && N > 0 && K > 0 && K <= 8192, "las_fe_matmul: bad shape");
  hipLaunchKernelGGL(matmul_kernel, dim3(M), dim3(256), K * sizeof(float), (hipStream_t)stream, A, lda, W, ldw, C, ldc, M, N, K,
                     epilogue, eps);
  LAS_LAUNCH_CHECK("fe matmul launch");
  return LAS_OK;
}

extern "C" int las_fe_top_db(float* x, int64_t ldx, int rows, int cols, float top_db, float* scratch, void* stream) {
  LAS_REQUIRE(rows > 0 && cols > 0, "las_fe_top_db: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const float ninf = -INFINITY;
  int rc = las_check_hip(hipMemcpyAsync(scratch, &ninf, sizeof(float), hipMemcpyHostToDevice, st), "init max");
  if (rc) return rc;
  hipLaunchKernelGGL(max_kernel, dim3(nblocks((int64_t)rows * cols)), dim3(256), 0, st, x, ldx, rows, cols, scratch);
  hipLaunchKernelGGL(floor_kernel, dim3(nblocks((int64_t)rows * cols)), dim3(256), 0, st, x, ldx, rows, cols, scratch, top_db);
  LAS_LAUNCH_CHECK("top_db launch");
  return LAS_OK;
}

extern "C" int las_fe_rms(const float* wave, int num_samples, int frame_length, int hop, float* out, int64_t ldo, int frames,
                          void* stream) {
  LAS_REQUIRE(num_samples > 0 && frame_length > 0 && hop > 0 && frames > 0, "las_fe_rms: bad arguments");
  hipLaunchKernelGGL(rms_kernel, dim3(frames), dim3(256), 0, (hipStream_t)stream, wave, num_samples, frame_length, hop,
                     frame_length / 2, out, ldo);
  LAS_LAUNCH_CHECK("rms launch");
  return LAS_OK;
}

extern "C" int las_fe_delta(const float* x, int64_t ldx, int T, int F, const float* taps, const float* edge_lo,
                            const float* edge_hi, int width, float* out, int64_t ldo, int out_stride, int out_offset,
                            void* stream) {
  LAS_REQUIRE(T >= width && F > 0 && width > 0 && (width & 1), "las_fe_delta: need T >= width (odd)");
  hipStream_t st = (hipStream_t)stream;
  if (taps) {
    hipLaunchKernelGGL(delta_kernel, dim3(nblocks((int64_t)T * F)), dim3(256), 0, st, x, ldx, T, F, taps, edge_lo, edge_hi, width,
                       out, ldo, out_stride, out_offset);
  } else {
    hipLaunchKernelGGL(copy_strided_kernel, dim3(nblocks((int64_t)T * F)), dim3(256), 0, st, x, ldx, T, F, out, ldo, out_stride,
                       out_offset);
  }
  LAS_LAUNCH_CHECK("delta launch");
  return LAS_OK;
}

extern "C" int las_fe_batch_melspec(const float* waves, const int64_t* wave_off, const int32_t* frame_off, int n_utt,
                                    int total_frames, int n_fft, int hop, int center, int power, const float* window,
                                    const float* costab, const float* sintab, int bins, const float* mel, int n_mels,
                                    int epilogue, float eps, float scale, float* out, float* utt_max, void* stream) {
  LAS_REQUIRE(waves && wave_off && frame_off && n_utt > 0 && total_frames > 0 && n_fft > 0 && n_fft <= 8192 && hop > 0 && bins > 0 &&
                  bins <= 8192 && (power == 1 || power == 2) && mel && n_mels > 0 && out, "las_fe_batch_melspec: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (utt_max) {
    // (-inf in every slot: the order-preserving integer maximum starts below everything)
    hipLaunchKernelGGL(fill_f32_kernel, dim3((n_utt + 255) / 256), dim3(256), 0, st, utt_max, n_utt, -INFINITY);
    LAS_LAUNCH_CHECK("fe batch max init");
  }
  const int blocked = las_knob("LAS_FE_BLOCKED", 1);          // 0: one frame per workgroup again (diagnostics, A/B timing)
  const int fb16 = las_knob("LAS_FE_FB", 8) == 16 ? 1 : 0;     // 16: sixteen frames per workgroup (A/B)
  constexpr int FB = 8;
  if (blocked && fb16 && (size_t)16 * (n_fft + bins) * sizeof(float) <= 60 * 1024 && total_frames >= 64)
    hipLaunchKernelGGL(batch_melspec_blocked_kernel<16>, dim3((total_frames + 15) / 16), dim3(256),
                       (size_t)16 * (n_fft + bins) * sizeof(float), st, waves, wave_off, frame_off, n_utt, total_frames, n_fft, hop,
                       center ? n_fft / 2 : 0, power, window, costab, sintab, bins, mel, n_mels, epilogue, eps, scale, out, utt_max);
  else if (blocked && (size_t)FB * (n_fft + bins) * sizeof(float) <= 60 * 1024 && total_frames >= 4 * FB)
    hipLaunchKernelGGL(batch_melspec_blocked_kernel<FB>, dim3((total_frames + FB - 1) / FB), dim3(256),
                       (size_t)FB * (n_fft + bins) * sizeof(float), st, waves, wave_off, frame_off, n_utt, total_frames, n_fft, hop,
                       center ? n_fft / 2 : 0, power, window, costab, sintab, bins, mel, n_mels, epilogue, eps, scale, out, utt_max);
  else
  hipLaunchKernelGGL(batch_melspec_kernel, dim3(total_frames), dim3(256), (size_t)(n_fft + bins) * sizeof(float), st, waves, wave_off,
                     frame_off, n_utt, n_fft, hop, center ? n_fft / 2 : 0, power, window, costab, sintab, bins, mel, n_mels, epilogue,
                     eps, scale, out, utt_max);
  LAS_LAUNCH_CHECK("fe batch melspec launch");
  return LAS_OK;
}

extern "C" int las_fe_batch_finish(const float* mel_db, int n_mels, const int32_t* frame_off, int n_utt, int total_frames,
                                   const float* utt_max, float top_db, const float* dct, int n_out, const float* waves,
                                   const int64_t* wave_off, int frame_length, int hop, int energy, float* out, int64_t ldo,
                                   void* stream) {
  LAS_REQUIRE(mel_db && frame_off && n_utt > 0 && total_frames > 0 && n_mels > 0 && n_mels <= 8192 && n_out > 0 && out &&
                  ldo >= n_out + (energy ? 1 : 0) && (dct || n_out <= n_mels) && (!energy || (waves && wave_off && frame_length > 0 && hop > 0)),
              "las_fe_batch_finish: bad arguments");
  hipLaunchKernelGGL(batch_finish_kernel, dim3(total_frames), dim3(256), (size_t)n_mels * sizeof(float), (hipStream_t)stream, mel_db,
                     n_mels, frame_off, n_utt, utt_max, top_db, dct, n_out, waves, wave_off, frame_length, hop, energy, out, ldo);
  LAS_LAUNCH_CHECK("fe batch finish launch");
  return LAS_OK;
}

extern "C" int las_fe_batch_delta(const float* x, int64_t ldx, const int32_t* frame_off, int n_utt, int total_frames, int F,
                                  const float* taps1, const float* lo1, const float* hi1, const float* taps2, const float* lo2,
                                  const float* hi2, int width, float* out, int64_t ldo, void* stream) {
  LAS_REQUIRE(x && frame_off && n_utt > 0 && total_frames > 0 && F > 0 && width > 0 && (width & 1) && taps1 && lo1 && hi1 && taps2 && lo2 &&
                  hi2 && out && ldo >= 3 * F, "las_fe_batch_delta: bad arguments (every utterance needs at least `width` frames)");
  hipLaunchKernelGGL(batch_delta_kernel, dim3(nblocks((int64_t)total_frames * F)), dim3(256), 0, (hipStream_t)stream, x, ldx, frame_off,
                     n_utt, total_frames, F, taps1, lo1, hi1, taps2, lo2, hi2, width, out, ldo);
  LAS_LAUNCH_CHECK("fe batch delta launch");
  return LAS_OK;
}
