// Train op of model_helper.py:403-417 as three multi-tensor kernels over FLAT fp32 buffers:
//   (1) g += l2 * theta (gradient of scale * sum(theta^2)/2 over ALL variables) and per-tensor ||g||^2,
//   (2) clip_by_norm(g, clip) per tensor,
//   (3) tf.train.AdamOptimizer update (epsilon outside the bias correction).
// Keeping parameters/gradients/slots flat makes the data-parallel exchange one RCCL all-reduce over the
// gradient buffer between (2) and (3) (CrossShardOptimizer order: clip locally, then sum).
#include "las_common.h"

namespace {

constexpr int CHUNK = 4096;   // elements per workgroup pass

__device__ __forceinline__ int find_seg(const int64_t* off, int nseg, int64_t i) {
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void l2_norm_kernel(float* g, const float* p, const int64_t* off, int nseg, int64_t total,
                                                      float l2, float* sumsq) {
  __shared__ float red[4];
  for (int64_t base = (int64_t)blockIdx.x * CHUNK; base < total; base += (int64_t)gridDim.x * CHUNK) {
    const int64_t end = min(total, base + CHUNK);
    const int seg0 = find_seg(off, nseg, base);
    const int seg1 = find_seg(off, nseg, end - 1);
    if (seg0 == seg1) {
      // the common case: the whole chunk lies inside one tensor (no per-element search)
      float acc = 0.f;
      for (int64_t i = base + threadIdx.x; i < end; i += 256) {
        const float v = g[i] + l2 * p[i];
        g[i] = v;
        acc += v * v;
      }
      const float v = las_wave_sum(acc);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
      __syncthreads();
      if (threadIdx.x == 0) atomicAdd(sumsq + seg0, red[0] + red[1] + red[2] + red[3]);
    } else {
      // the chunk straddles tensor boundaries: accumulate per run of equal segment
      int seg = -1;
      float acc = 0.f;
      for (int64_t i = base + threadIdx.x; i < end; i += 256) {
        const int sgi = find_seg(off, nseg, i);
        const float v = g[i] + l2 * p[i];
        g[i] = v;
        if (sgi != seg) {
          if (seg >= 0) atomicAdd(sumsq + seg, acc);
          seg = sgi;
          acc = 0.f;
        }
        acc += v * v;
      }
      if (seg >= 0) atomicAdd(sumsq + seg, acc);
    }
  }
}

__global__ __launch_bounds__(256) void clip_kernel(float* g, const int64_t* off, int nseg, int64_t total, const float* sumsq,
                                                   float clip) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int sgi = find_seg(off, nseg, i);
    const float n = sqrtf(sumsq[sgi]);
    g[i] = g[i] * (clip / fmaxf(n, clip));
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, float* m, float* v, const float* g, int64_t total, float lr,
                                                   float b1, float b2, float eps, int step, const int32_t* step_dev) {
  // lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); t read from device memory when given (graph replay)
  const double t = (double)(step_dev ? *step_dev : step);
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* x, int64_t n, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += x[i] * x[i];
  acc = las_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ void counter_add_kernel(int32_t* c, int32_t d) { if (threadIdx.x == 0 && blockIdx.x == 0) *c += d; }

}  // namespace

extern "C" int las_counter_add(int32_t* counter, int32_t delta, void* stream) {
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, delta);
  LAS_LAUNCH_CHECK("counter launch");
  return LAS_OK;
}

extern "C" int las_sumsq(const float* x, int64_t n, float* out, void* stream) {
  LAS_REQUIRE(n > 0, "las_sumsq: empty");
  int blocks = (int)((n + 4095) / 4096);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, out);
  LAS_LAUNCH_CHECK("sumsq launch");
  return LAS_OK;
}

extern "C" int las_grad_l2_norms(float* grads, const float* params, const int64_t* seg_offsets, int nseg, int64_t total,
                                 float l2_scale, float* sumsq, void* stream) {
  LAS_REQUIRE(nseg > 0 && total > 0, "las_grad_l2_norms: empty");
  hipStream_t st = (hipStream_t)stream;
  int rc = las_check_hip(hipMemsetAsync(sumsq, 0, sizeof(float) * nseg, st), "memset sumsq");
  if (rc) return rc;
  int blocks = (int)((total + CHUNK - 1) / CHUNK);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(l2_norm_kernel, dim3(blocks), dim3(256), 0, st, grads, params, seg_offsets, nseg, total, l2_scale, sumsq);
  LAS_LAUNCH_CHECK("l2 norm launch");
  return LAS_OK;
}

extern "C" int las_grad_clip(float* grads, const int64_t* seg_offsets, int nseg, int64_t total, const float* sumsq,
                             float clip, void* stream) {
  LAS_REQUIRE(nseg > 0 && total > 0 && clip > 0.f, "las_grad_clip: bad arguments");
  int blocks = (int)((total + 1023) / 1024);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(clip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grads, seg_offsets, nseg, total, sumsq, clip);
  LAS_LAUNCH_CHECK("clip launch");
  return LAS_OK;
}

extern "C" int las_adam_update(float* params, float* m, float* v, const float* grads, int64_t total, float lr,
                               float beta1, float beta2, float eps, int step, const int32_t* step_dev, void* stream) {
  LAS_REQUIRE(total > 0 && (step >= 1 || step_dev), "las_adam_update: bad arguments");
  int blocks = (int)((total + 1023) / 1024);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, m, v, grads, total, lr,
                     beta1, beta2, eps, step, step_dev);
  LAS_LAUNCH_CHECK("adam launch");
  return LAS_OK;
}
