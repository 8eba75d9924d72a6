// Train op of model_helper.py:403-417 as three multi-tensor kernels over FLAT fp32 buffers:
//   (1) g += l2 * theta (gradient of scale * sum(theta^2)/2 over ALL variables) and per-tensor ||g||^2,
//   (2) clip_by_norm(g, clip) per tensor,
//   (3) tf.train.AdamOptimizer update (epsilon outside the bias correction).
// Keeping parameters/gradients/slots flat makes the data-parallel exchange one RCCL all-reduce over the
// gradient buffer between (2) and (3) (CrossShardOptimizer order: clip locally, then sum).
#include "las_common.h"

namespace {

constexpr int SPAN = 16384;   // elements per workgroup (a contiguous range: one segment search per workgroup)

__device__ __forceinline__ int find_seg(const int64_t* off, int nseg, int64_t i) {
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = las_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Walks the pieces [cur, stop) of this workgroup's range that lie inside one tensor and calls body(i, n) for runs of
// n = 4 (16-byte aligned) or n = 1 elements; `flush(seg)` runs after every piece.  Same-address atomics are the cost
// to avoid here: one per (workgroup, tensor piece) instead of one per 4096 elements.
template <typename Body, typename Flush>
__device__ __forceinline__ void for_each_piece(const int64_t* off, int nseg, int64_t total, Body body, Flush flush) {
  const int64_t start = (int64_t)blockIdx.x * SPAN;
  const int64_t end = min(total, start + SPAN);
  if (start >= end) return;
  int seg = find_seg(off, nseg, start);
  int64_t cur = start;
  while (cur < end) {
    const int64_t stop = (seg + 1 < nseg) ? min(end, off[seg + 1]) : end;
    if ((cur & 3) == 0) {
      const int64_t n4 = (stop - cur) >> 2;
      for (int64_t q = threadIdx.x; q < n4; q += 256) body(cur + 4 * q, 4, seg);
      for (int64_t i = cur + 4 * n4 + threadIdx.x; i < stop; i += 256) body(i, 1, seg);
    } else {
      for (int64_t i = cur + threadIdx.x; i < stop; i += 256) body(i, 1, seg);
    }
    flush(seg);
    cur = stop;
    ++seg;
  }
}

// ws != nullptr: no fp32 atomics -- every workgroup stores its per-tensor partial sums as a dense row [nseg + 1] (last column:
// sum theta^2) of the workspace, and the workgroup that finishes LAST (a counter in the workspace head, reset by that workgroup
// for the next launch) adds the rows in workgroup order, one wave per tensor with a fixed lane -> row mapping: the norms, hence
// the clip factors, hence the whole update are bit-identical from run to run.
__global__ __launch_bounds__(256) void l2_norm_kernel(float* g, const float* p, const int64_t* off, int nseg, int64_t total,
                                                      float l2, float* sumsq, float* param_sumsq, unsigned* ws) {
  __shared__ float red[4];
  __shared__ unsigned last;
  float* rows = ws ? reinterpret_cast<float*>(ws + 16) : nullptr;      // [gridDim.x][nseg + 1]
  float* my = rows ? rows + (int64_t)blockIdx.x * (nseg + 1) : nullptr;
  // The row is written by thread 0 alone with WRITE-THROUGH stores (and read by the last workgroup with L2-bypassing loads): a
  // release fence at agent scope instead -- __threadfence() -- makes every workgroup write back its XCD's dirty L2 lines, i.e.
  // the 26 MB of gradients this very kernel has just written: 77 us instead of 21 (rocprofv3, metric-M).
  if (my && threadIdx.x == 0)
    for (int i = 0; i <= nseg; ++i) __hip_atomic_store(my + i, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float acc = 0.f, pacc = 0.f;
  for_each_piece(off, nseg, total,
      [&](int64_t i, int n, int) {
        if (n == 4) {
          float4 gv = *reinterpret_cast<const float4*>(g + i);
          const float4 pv = *reinterpret_cast<const float4*>(p + i);
          gv.x += l2 * pv.x; gv.y += l2 * pv.y; gv.z += l2 * pv.z; gv.w += l2 * pv.w;
          *reinterpret_cast<float4*>(g + i) = gv;
          acc += gv.x * gv.x + gv.y * gv.y + gv.z * gv.z + gv.w * gv.w;
          pacc += pv.x * pv.x + pv.y * pv.y + pv.z * pv.z + pv.w * pv.w;
        } else {
          const float pv = p[i], v = g[i] + l2 * pv;
          g[i] = v;
          acc += v * v;
          pacc += pv * pv;
        }
      },
      [&](int seg) {
        const float v = block_sum(acc, red);
        if (threadIdx.x == 0) {
          if (my) __hip_atomic_store(my + seg, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else atomicAdd(sumsq + seg, v);
        }
        acc = 0.f;
      });
  if (param_sumsq) {
    const float v = block_sum(pacc, red);
    if (threadIdx.x == 0) {
      if (my) __hip_atomic_store(my + nseg, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else atomicAdd(param_sumsq, v);
    }
  }
  if (!ws) return;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // thread 0's write-through stores of the row are acknowledged
    last = (__hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!last) return;
  // One wave per tensor (waves take tensors w, w + 4, ...).  A workgroup covers SPAN contiguous elements, so the rows that can
  // hold a partial sum of tensor `seg` are the workgroups off[seg] / SPAN ... (off[seg + 1] - 1) / SPAN and nothing else: the
  // wave reads those (lane l the rows b0 + l, b0 + l + 64, ...: L2-bypassing loads, independent, many in flight) and adds its
  // lanes in the butterfly's fixed order -- the sums depend on the layout only, not on scheduling.  (Until round 5 every one of
  // the nseg + 1 columns was read over ALL rows, eight loads in flight per thread: 2 228 rows x 41 columns took 170 of the
  // kernel's 281 us at metric-L, on the critical path behind the last recurrence; the column of sum theta^2 does need all rows
  // and is read by the whole workgroup.)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int seg = wave; seg < nseg; seg += 4) {
    const int64_t lo = off[seg], hi = (seg + 1 < nseg) ? off[seg + 1] : total;
    float t = 0.f;
    if (hi > lo) {
      const unsigned b0 = (unsigned)(lo / SPAN), b1 = (unsigned)((hi - 1) / SPAN);
      const float* src = rows + seg;
#pragma unroll 4
      for (unsigned b = b0 + lane; b <= b1; b += 64) t += __hip_atomic_load(src + (int64_t)b * (nseg + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    t = las_wave_sum(t);
    if (lane == 0 && hi > lo) sumsq[seg] += t;
  }
  if (param_sumsq) {
    float t = 0.f;
    const float* src = rows + nseg;
#pragma unroll 4
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t += __hip_atomic_load(src + (int64_t)b * (nseg + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = block_sum(t, red);
    if (threadIdx.x == 0) *param_sumsq += t;
  }
  if (threadIdx.x == 0) *ws = 0u;
}

__global__ __launch_bounds__(256) void clip_kernel(float* g, const int64_t* off, int nseg, int64_t total, const float* sumsq,
                                                   float clip) {
  for_each_piece(off, nseg, total,
      [&](int64_t i, int n, int seg) {
        const float sc = clip / fmaxf(sqrtf(sumsq[seg]), clip);
        if (n == 4) {
          float4 gv = *reinterpret_cast<const float4*>(g + i);
          gv.x *= sc; gv.y *= sc; gv.z *= sc; gv.w *= sc;
          *reinterpret_cast<float4*>(g + i) = gv;
        } else {
          g[i] *= sc;
        }
      },
      [](int) {});
}

// lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); t read from device memory when given (graph replay)
__device__ __forceinline__ float adam_lr_t(float lr, float b1, float b2, int step, const int32_t* step_dev) {
  const double t = (double)(step_dev ? *step_dev : step);
  return (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
}

__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float gi, float lr_t, float b1, float b2, float eps) {
  m = b1 * m + (1.f - b1) * gi;
  v = b2 * v + (1.f - b2) * gi * gi;
  p = p - lr_t * m / (sqrtf(v) + eps);
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, float* m, float* v, const float* g, int64_t total, float lr,
                                                   float b1, float b2, float eps, int step, const int32_t* step_dev,
                                                   const float* skip) {
  if (skip && *skip != 0.f) return;        // a persistent kernel of this step (on any replica) reported a timeout: no update
  const float lr_t = adam_lr_t(lr, b1, b2, step, step_dev);
  const int64_t n4 = total >> 2;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    const float4 gv = reinterpret_cast<const float4*>(g)[q];
    float4 pv = reinterpret_cast<float4*>(p)[q], mv = reinterpret_cast<float4*>(m)[q], vv = reinterpret_cast<float4*>(v)[q];
    adam_one(pv.x, mv.x, vv.x, gv.x, lr_t, b1, b2, eps);
    adam_one(pv.y, mv.y, vv.y, gv.y, lr_t, b1, b2, eps);
    adam_one(pv.z, mv.z, vv.z, gv.z, lr_t, b1, b2, eps);
    adam_one(pv.w, mv.w, vv.w, gv.w, lr_t, b1, b2, eps);
    reinterpret_cast<float4*>(p)[q] = pv;
    reinterpret_cast<float4*>(m)[q] = mv;
    reinterpret_cast<float4*>(v)[q] = vv;
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    adam_one(p[i], m[i], v[i], g[i], lr_t, b1, b2, eps);
}

// clip_by_norm and the Adam update in one pass (single replica: nothing happens between them); the clipped gradient is
// written back so that the gradient buffer reads the same as after las_grad_clip.
__global__ __launch_bounds__(256) void clip_adam_kernel(float* p, float* m, float* v, float* g, const int64_t* off, int nseg,
                                                        int64_t total, const float* sumsq, float clip, float lr, float b1,
                                                        float b2, float eps, int step, const int32_t* step_dev,
                                                        const float* skip) {
  if (skip && *skip != 0.f) return;
  const float lr_t = adam_lr_t(lr, b1, b2, step, step_dev);
  for_each_piece(off, nseg, total,
      [&](int64_t i, int n, int seg) {
        const float sc = clip / fmaxf(sqrtf(sumsq[seg]), clip);
        if (n == 4) {
          float4 gv = *reinterpret_cast<const float4*>(g + i);
          float4 pv = *reinterpret_cast<float4*>(p + i), mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
          gv.x *= sc; gv.y *= sc; gv.z *= sc; gv.w *= sc;
          adam_one(pv.x, mv.x, vv.x, gv.x, lr_t, b1, b2, eps);
          adam_one(pv.y, mv.y, vv.y, gv.y, lr_t, b1, b2, eps);
          adam_one(pv.z, mv.z, vv.z, gv.z, lr_t, b1, b2, eps);
          adam_one(pv.w, mv.w, vv.w, gv.w, lr_t, b1, b2, eps);
          *reinterpret_cast<float4*>(g + i) = gv;
          *reinterpret_cast<float4*>(p + i) = pv;
          *reinterpret_cast<float4*>(m + i) = mv;
          *reinterpret_cast<float4*>(v + i) = vv;
        } else {
          const float gi = g[i] * sc;
          g[i] = gi;
          adam_one(p[i], m[i], v[i], gi, lr_t, b1, b2, eps);
        }
      },
      [](int) {});
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

__global__ void counter_add_kernel(int32_t* c, int32_t d, const float* skip) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !(skip && *skip != 0.f)) *c += d;
}

// flag = 1.0 when any of the n status words (first word of a persistent kernel's workspace) is non-zero, else 0.0
__global__ void status_collect_kernel(const unsigned* const* words, int n, float* flag) {
  unsigned any = 0;
  for (int i = threadIdx.x; i < n; i += 64) any |= __hip_atomic_load(words[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) any |= __shfl_xor(any, o, 64);
  if (threadIdx.x == 0) *flag = any ? 1.f : 0.f;
}

// status_collect_kernel + the zeroing of the norm accumulators the gradient passes of this step add into
__global__ void train_op_begin_kernel(const unsigned* const* words, int n, float* flag, float* sumsq, int nseg, float* psq, int npsq) {
  for (int i = threadIdx.x; i < nseg; i += 64) sumsq[i] = 0.f;
  for (int i = threadIdx.x; i < npsq; i += 64) psq[i] = 0.f;
  unsigned any = 0;
  for (int i = threadIdx.x; i < n; i += 64) any |= __hip_atomic_load(words[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) any |= __shfl_xor(any, o, 64);
  if (threadIdx.x == 0 && flag) *flag = any ? 1.f : 0.f;
}

// out = audio + half_l2 * sum(psq[0..npsq))
__global__ void total_loss_kernel(const float* audio, const float* psq, int npsq, float half_l2, float* out) {
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < npsq; ++i) s += psq[i];
    *out = (audio ? *audio : 0.f) + half_l2 * s;
  }
}

}  // namespace

extern "C" int las_train_op_begin(const uint32_t* const* status_words, int n, float* flag, float* sumsq, int nseg,
                                  float* param_sumsq, int npsq, void* stream) {
  LAS_REQUIRE(n >= 0 && nseg >= 0 && npsq >= 0 && (n == 0 || status_words != nullptr) && (nseg == 0 || sumsq != nullptr) &&
              (npsq == 0 || param_sumsq != nullptr), "las_train_op_begin: bad arguments");
  hipLaunchKernelGGL(train_op_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, status_words, n, flag, sumsq, nseg, param_sumsq, npsq);
  LAS_LAUNCH_CHECK("train op begin launch");
  return LAS_OK;
}

extern "C" int las_total_loss(const float* audio_loss, const float* param_sumsq, int npsq, float half_l2_scale, float* out, void* stream) {
  LAS_REQUIRE(out != nullptr && npsq >= 0 && (npsq == 0 || param_sumsq != nullptr), "las_total_loss: bad arguments");
  hipLaunchKernelGGL(total_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, audio_loss, param_sumsq, npsq, half_l2_scale, out);
  LAS_LAUNCH_CHECK("total loss launch");
  return LAS_OK;
}

extern "C" int las_status_collect(const uint32_t* const* status_words, int n, float* flag, void* stream) {
  LAS_REQUIRE(n >= 0 && flag != nullptr && (n == 0 || status_words != nullptr), "las_status_collect: bad arguments");
  hipLaunchKernelGGL(status_collect_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, status_words, n, flag);
  LAS_LAUNCH_CHECK("status collect launch");
  return LAS_OK;
}

extern "C" int las_counter_add_unless(int32_t* counter, int32_t delta, const float* skip_flag, void* stream) {
  LAS_REQUIRE(counter != nullptr, "las_counter_add_unless: null counter");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, delta, skip_flag);
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

extern "C" size_t las_grad_l2_norms_ws_bytes(int nseg, int64_t total);

static int grad_l2_norms(float* grads, const float* params, const int64_t* seg_offsets, int nseg, int64_t total,
                         float l2_scale, float* sumsq, float* param_sumsq, bool zero_first, void* workspace, size_t workspace_bytes,
                         void* stream) {
  LAS_REQUIRE(nseg > 0 && total > 0, "las_grad_l2_norms: empty");
  LAS_REQUIRE(((uintptr_t)grads % 16 == 0) && ((uintptr_t)params % 16 == 0), "las_grad_l2_norms: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (zero_first) {
    // one small launch, not hipMemsetAsync: as MEMSET NODES of a captured graph the 16- and 4-byte clears this used to enqueue left
    // garbage in the accumulators (round 5, tests/test_gpu_step_forms.py: six of six replays of the two-bucket exchange form
    // had a bottom-layer ||g||^2 of -7.8e31 -- harmless while negative, a zeroed gradient when it came out positive; the same
    // calls launched eagerly, and the large memsets of the recurrent / decoder workspaces inside graphs, were never seen wrong)
    hipLaunchKernelGGL(train_op_begin_kernel, dim3(1), dim3(64), 0, st, nullptr, 0, nullptr, sumsq, nseg, param_sumsq, param_sumsq ? 1 : 0);
    LAS_LAUNCH_CHECK("norm accumulators zeroing launch");
  }
  LAS_REQUIRE(workspace == nullptr || (workspace_bytes >= las_grad_l2_norms_ws_bytes(nseg, total) && (uintptr_t)workspace % 16 == 0),
              "las_grad_l2_norms: workspace of las_grad_l2_norms_ws_bytes(nseg, total) bytes needed");
  hipLaunchKernelGGL(l2_norm_kernel, dim3((unsigned)((total + SPAN - 1) / SPAN)), dim3(256), 0, st, grads, params, seg_offsets, nseg,
                     total, l2_scale, sumsq, param_sumsq, static_cast<unsigned*>(workspace));
  LAS_LAUNCH_CHECK("l2 norm launch");
  return LAS_OK;
}

extern "C" size_t las_grad_l2_norms_ws_bytes(int nseg, int64_t total) {
  return 64 + sizeof(float) * (size_t)((total + SPAN - 1) / SPAN) * (size_t)(nseg + 1);
}

extern "C" int las_grad_l2_norms(float* grads, const float* params, const int64_t* seg_offsets, int nseg, int64_t total,
                                 float l2_scale, float* sumsq, float* param_sumsq, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  return grad_l2_norms(grads, params, seg_offsets, nseg, total, l2_scale, sumsq, param_sumsq, true, workspace, workspace_bytes, stream);
}

extern "C" int las_grad_l2_norms_acc(float* grads, const float* params, const int64_t* seg_offsets, int nseg, int64_t total,
                                     float l2_scale, float* sumsq, float* param_sumsq, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  return grad_l2_norms(grads, params, seg_offsets, nseg, total, l2_scale, sumsq, param_sumsq, false, workspace, workspace_bytes, stream);
}

extern "C" int las_grad_clip(float* grads, const int64_t* seg_offsets, int nseg, int64_t total, const float* sumsq,
                             float clip, void* stream) {
  LAS_REQUIRE(nseg > 0 && total > 0 && clip > 0.f, "las_grad_clip: bad arguments");
  LAS_REQUIRE((uintptr_t)grads % 16 == 0, "las_grad_clip: buffer must be 16-byte aligned");
  hipLaunchKernelGGL(clip_kernel, dim3((unsigned)((total + SPAN - 1) / SPAN)), dim3(256), 0, (hipStream_t)stream, grads, seg_offsets,
                     nseg, total, sumsq, clip);
  LAS_LAUNCH_CHECK("clip launch");
  return LAS_OK;
}

extern "C" int las_adam_update(float* params, float* m, float* v, const float* grads, int64_t total, float lr,
                               float beta1, float beta2, float eps, int step, const int32_t* step_dev, const float* skip_flag,
                               void* stream) {
  LAS_REQUIRE(total > 0 && (step >= 1 || step_dev), "las_adam_update: bad arguments");
  LAS_REQUIRE((((uintptr_t)params | (uintptr_t)m | (uintptr_t)v | (uintptr_t)grads) % 16) == 0, "las_adam_update: buffers must be 16-byte aligned");
  int blocks = (int)((total + 4095) / 4096);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, m, v, grads, total, lr,
                     beta1, beta2, eps, step, step_dev, skip_flag);
  LAS_LAUNCH_CHECK("adam launch");
  return LAS_OK;
}

extern "C" int las_clip_adam_update(float* params, float* m, float* v, float* grads, const int64_t* seg_offsets, int nseg,
                                    int64_t total, const float* sumsq, float clip, float lr, float beta1, float beta2,
                                    float eps, int step, const int32_t* step_dev, const float* skip_flag, void* stream) {
  LAS_REQUIRE(nseg > 0 && total > 0 && clip > 0.f && (step >= 1 || step_dev), "las_clip_adam_update: bad arguments");
  LAS_REQUIRE((((uintptr_t)params | (uintptr_t)m | (uintptr_t)v | (uintptr_t)grads) % 16) == 0, "las_clip_adam_update: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)((total + SPAN - 1) / SPAN)), dim3(256), 0, (hipStream_t)stream, params, m, v,
                     grads, seg_offsets, nseg, total, sumsq, clip, lr, beta1, beta2, eps, step, step_dev, skip_flag);
  LAS_LAUNCH_CHECK("clip + adam launch");
  return LAS_OK;
}
