// Error state, version and the small element-wise / reduction kernels of liblas_hip.
#include "las_common.h"
#include <string.h>

int las_check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return LAS_OK;
  las_set_error("%s: %s", what, hipGetErrorString(e));
  return LAS_ERR_HIP;
}

// ---- knobs: name -> value, read from the environment once (first use), overridable through las_set_knob ----
#include <map>
#include <mutex>
#include <string>
#include <stdlib.h>
static std::mutex g_knob_mutex;
static std::map<std::string, int>& knob_table() { static std::map<std::string, int> t; return t; }

int las_knob(const char* name, int default_value) {
  std::lock_guard<std::mutex> lock(g_knob_mutex);
  auto& t = knob_table();
  auto it = t.find(name);
  if (it != t.end()) return it->second;
  const char* e = getenv(name);
  const int v = (e && *e) ? atoi(e) : default_value;
  t[name] = v;
  return v;
}

extern "C" int las_set_knob(const char* name, int value) {
  LAS_REQUIRE(name != nullptr && name[0] == 'L' && name[1] == 'A' && name[2] == 'S' && name[3] == '_', "las_set_knob: knob names start with LAS_");
  std::lock_guard<std::mutex> lock(g_knob_mutex);
  knob_table()[name] = value;
  return LAS_OK;
}

// 101 (round 6): host.cpp split off, the recurrent backward launches eight-wave workgroups at 64 / 128 / 256 units (same entry points,
// same results); 100 (round 5) removed five entry points and gave las_lstm_fwd::reserved1 a meaning (rows_per_slice).
extern "C" int las_version(void) { return 102; }

namespace {

__global__ void cast_kernel(const float* src, int64_t lds_, int rows, int cols, unsigned short* dst, int64_t ldd,
                            int dst_rows, int dst_cols, int transpose, int64_t sbs, int64_t dbs, int perm_h) {
  // one thread per destination element of the [dst_rows, dst_cols] window (row stride ldd)
  src += (int64_t)blockIdx.y * sbs;
  dst += (int64_t)blockIdx.y * dbs;
  const int64_t total = (int64_t)dst_rows * dst_cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int dr = (int)(i / dst_cols), dc = (int)(i % dst_cols);
    const int sr = transpose ? dc : dr;
    int sc = transpose ? dr : dc;
    float v = 0.f;
    if (sr < rows && sc < cols) {
      if (perm_h > 0) sc = (sc & 3) * perm_h + (sc >> 2);   // logical gate-interleaved column u*4+g <- TF column g*H+u
      v = src[(int64_t)sr * lds_ + sc];
    }
    dst[(int64_t)dr * ldd + dc] = las_f2bf(v);
  }
}

// All bf16 / re-ordered images of the fp32 master weights in ONE launch (they are rebuilt after every optimiser step:
// ~40 tiny launches otherwise).  blockIdx.y selects the job; the job table lives in device memory.
__global__ __launch_bounds__(256) void refresh_images_kernel(const las_image_job* jobs) {
  const las_image_job jb = jobs[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t first = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (jb.kind == LAS_IMAGE_CAST) {
    unsigned short* dst = static_cast<unsigned short*>(jb.dst);
    const int64_t total = (int64_t)jb.dst_rows * jb.dst_cols;
    for (int64_t i = first; i < total; i += stride) {
      const int dr = (int)(i / jb.dst_cols), dc = (int)(i % jb.dst_cols);
      const int sr = jb.transpose ? dc : dr;
      int sc = jb.transpose ? dr : dc;
      float v = 0.f;
      if (sr < jb.rows && sc < jb.cols) {
        if (jb.perm_h > 0) sc = (sc & 3) * jb.perm_h + (sc >> 2);
        v = jb.src[(int64_t)sr * jb.lds + sc];
      }
      dst[(int64_t)dr * jb.ldd + dc] = las_f2bf(v);
    }
  } else if (jb.kind == LAS_IMAGE_PACK_RECURRENT) {
    unsigned short* dst = static_cast<unsigned short*>(jb.dst);
    const int H = jb.rows;
    const int64_t total = (int64_t)H * 4 * H;
    for (int64_t i = first; i < total; i += stride) dst[i] = las_f2bf(jb.src[las_pack_recurrent_src(i, H)]);
  } else if (jb.kind == LAS_IMAGE_BIAS_INTERLEAVE) {
    float* dst = static_cast<float*>(jb.dst);
    const int H = jb.rows;
    for (int64_t i = first; i < 4 * (int64_t)H; i += stride) dst[i] = jb.src[(i & 3) * H + (i >> 2)];     // [u*4+g] <- [g*H+u]
  } else if (jb.kind == LAS_IMAGE_PACK_MFMA_B) {
    // [dst_rows / 16 column tiles][dst_cols / 32 K chunks][64 lanes][8]: lane l of a fragment holds row tile * 16 + (l & 15),
    // columns chunk * 32 + (l >> 4) * 8 + 0..7 of the source -- the B operand of v_mfma_f32_16x16x32_bf16 as one contiguous KB
    // (round 6, the weight images of las_gemm_nt_bimg: this window may be a K RANGE of a wider image -- ldd = the image's whole K,
    //  reserved = the window's first k --, and perm_h = H reads the source's COLUMN axis through the gate interleaving, logical
    //  index u * 4 + g <- TF column g * H + u)
    unsigned short* dst = static_cast<unsigned short*>(jb.dst);
    const int KC = jb.dst_cols / 32;
    const int KCT = jb.ldd > 0 ? (int)(jb.ldd / 32) : KC, kc0 = jb.reserved / 32;
    const int64_t total = (int64_t)jb.dst_rows * jb.dst_cols;
    for (int64_t i = first; i < total; i += stride) {
      const int e = (int)(i & 7), lane = (int)(i >> 3) & 63;
      const int64_t frag = i >> 9;
      const int nt = (int)(frag / KC), kc = (int)(frag % KC);
      const int n = nt * 16 + (lane & 15), k = kc * 32 + (lane >> 4) * 8 + e;
      // (transpose: the image of the source's transpose -- row n of the product's B operand is COLUMN n of src [cols, rows])
      int sr = jb.transpose ? k : n, sc = jb.transpose ? n : k;
      if (jb.perm_h > 0) sc = (sc & 3) * jb.perm_h + (sc >> 2);
      dst[(((int64_t)nt * KCT + kc0 + kc) << 9) + (lane << 3) + e] = las_f2bf(n < jb.rows && k < jb.cols ? jb.src[(int64_t)sr * jb.lds + sc] : 0.f);
    }
  } else if (jb.kind == LAS_IMAGE_PACK_INPUT) {
    unsigned short* dst = static_cast<unsigned short*>(jb.dst);
    const int D = jb.rows, H = jb.cols, chunks = jb.dst_rows;
    const int64_t total = (int64_t)(H / 16) * chunks * 4 * 512;
    for (int64_t i = first; i < total; i += stride) {
      const int64_t src = las_pack_input_src(i, D, H, chunks);
      dst[i] = las_f2bf(src >= 0 ? jb.src[src] : 0.f);
    }
  } else {                                                  // LAS_IMAGE_COPY_F32
    float* dst = static_cast<float*>(jb.dst);
    for (int64_t i = first; i < jb.cols; i += stride) dst[i] = jb.src[i];
  }
}

// holds its stream for about `ticks` cycles of the 100 MHz wall clock (las_stream_delay)
__global__ void delay_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

// las_stream_concurrency_probe: the waiter spins (bounded by the wall clock) until the setter's word arrives
__global__ void probe_wait_kernel(int* words, long long ticks) {
  const long long t0 = wall_clock64();
  int seen = 0;
  while (wall_clock64() - t0 < ticks) {
    if (__hip_atomic_load(words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { seen = 1; break; }
    __builtin_amdgcn_s_sleep(16);
  }
  words[1] = seen ? 1 : 2;
}
__global__ void probe_set_kernel(int* words) { __hip_atomic_store(words, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// out = bf16(a + b), b nullable (las_add_cast_bf16)
__global__ void add_cast_kernel(const float* a, int64_t lda, const float* b, int64_t ldb, unsigned short* out, int64_t ldo, int rows, int cols) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    float v = a[(int64_t)r * lda + c];
    if (b) v += b[(int64_t)r * ldb + c];
    out[(int64_t)r * ldo + c] = las_f2bf(v);
  }
}

struct FillTable { las_fill_job job[LAS_FILL_MAX_JOBS]; };

// blockIdx.y = job; the table is a kernel argument (see las_fill_many)
__global__ __launch_bounds__(256) void fill_many_kernel(FillTable t) {
  const las_fill_job jb = t.job[blockIdx.y];
  const int64_t total = jb.rows * jb.cols;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / jb.cols, c = i - r * jb.cols;
    if (jb.kind == LAS_FILL_ZERO32) static_cast<uint32_t*>(jb.dst)[r * jb.ldd + c] = 0u;
    else if (jb.kind == LAS_FILL_ZERO16) static_cast<unsigned short*>(jb.dst)[r * jb.ldd + c] = 0;
    else {
      const float v = jb.src ? static_cast<const float*>(jb.src)[r * jb.lds + c] : 0.f;
      if (jb.kind == LAS_FILL_COPY_F32) static_cast<float*>(jb.dst)[r * jb.ldd + c] = v;
      else static_cast<unsigned short*>(jb.dst)[r * jb.ldd + c] = las_f2bf(v);
    }
  }
}

__global__ void colsum_kernel(const unsigned short* X, int64_t ldx, int M, int N, float* out, int perm_h) {
  // blockDim = (64 columns, 4 row lanes); grid.x over column groups, grid.y over row chunks
  const int col = blockIdx.x * 64 + threadIdx.x;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per;
  const int r1 = min(M, r0 + rows_per);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + threadIdx.y; r < r1; r += blockDim.y) s += las_bf2f(X[(int64_t)r * ldx + col]);
  __shared__ float red[4][64];
  red[threadIdx.y][threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.y == 0 && col < N) {
    const int oc = perm_h > 0 ? (col & 3) * perm_h + (col >> 2) : col;
    atomicAdd(out + oc, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// The same sums with a result that does not depend on scheduling: every (column group, row chunk) block stores its 64 partial
// sums in the workspace, and the block that finishes LAST for a column group (a counter per group, reset by that block for the
// next launch) adds the chunks' partials in chunk order.  Workspace: [1024 counters][chunks][N rounded up to 64] floats.
__global__ void colsum_ws_kernel(const unsigned short* X, int64_t ldx, int M, int N, float* out, int perm_h, unsigned* counters,
                                 float* partial) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per;
  const int r1 = min(M, r0 + rows_per);
  const int N64 = gridDim.x * 64;
  float s = 0.f;
  if (col < N)
    for (int r = r0 + threadIdx.y; r < r1; r += blockDim.y) s += las_bf2f(X[(int64_t)r * ldx + col]);
  __shared__ float red[4][64];
  __shared__ unsigned last;
  red[threadIdx.y][threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.y == 0) {
    __hip_atomic_store(partial + (int64_t)blockIdx.y * N64 + col, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's write-through partial is acknowledged (no agent-scope fence:
  __syncthreads();                                       // it would write back the XCD's dirty L2 lines)
  if (threadIdx.x == 0 && threadIdx.y == 0)
    last = (__hip_atomic_fetch_add(counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.y - 1) ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  if (threadIdx.y == 0 && col < N) {
    float t = 0.f;
    for (unsigned y = 0; y < gridDim.y; ++y)
      t += __hip_atomic_load(partial + (int64_t)y * N64 + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int oc = perm_h > 0 ? (col & 3) * perm_h + (col >> 2) : col;
    out[oc] += t;
  }
  if (threadIdx.x == 0 && threadIdx.y == 0) counters[blockIdx.x] = 0u;
}

// The wide form of colsum_ws_kernel (N and ldx multiples of 8, X 16-byte aligned): a thread sums 8 columns with 16-byte loads,
// 32 column groups x 8 row lanes per workgroup = 256 columns; the two-byte loads of the narrow form ran a [51200 x 1024] bias
// sum at 0.4 TB/s (280 us, exposed behind the last recurrence of a step whose feature count rules the fused product out).
__global__ __launch_bounds__(256) void colsum_ws8_kernel(const unsigned short* X, int64_t ldx, int M, int N, float* out, int perm_h,
                                                         unsigned* counters, float* partial) {
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + cg * 8;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  const int N256 = gridDim.x * 256;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < N) {
    int r = r0 + rl;
    for (; r + 24 < r1; r += 32) {            // four rows in flight
      uint4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const uint4*>(X + (int64_t)(r + 8 * i) * ldx + col);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned short* e = reinterpret_cast<const unsigned short*>(&v[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += las_bf2f(e[j]);
      }
    }
    for (; r < r1; r += 8) {
      const uint4 v = *reinterpret_cast<const uint4*>(X + (int64_t)r * ldx + col);
      const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += las_bf2f(e[j]);
    }
  }
  __shared__ float red[8][256];
  __shared__ unsigned last;
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = acc[j];
  __syncthreads();
  {
    const int c = threadIdx.x;                 // one column of the 256 per thread
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][c];
    __hip_atomic_store(partial + (int64_t)blockIdx.y * N256 + blockIdx.x * 256 + c, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's write-through partial is acknowledged (no agent-scope fence:
  __syncthreads();                                       // it would write back the XCD's dirty L2 lines)
  if (threadIdx.x == 0)
    last = (__hip_atomic_fetch_add(counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.y - 1) ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < N) {
    float t = 0.f;
    for (unsigned y = 0; y < gridDim.y; ++y)
      t += __hip_atomic_load(partial + (int64_t)y * N256 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[perm_h > 0 ? (c & 3) * perm_h + (c >> 2) : c] += t;
  }
  if (threadIdx.x == 0) counters[blockIdx.x] = 0u;
}

__global__ void pyramid_len_kernel(const int32_t* a, int32_t* b, int B, int levels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  int v = a[i];
  for (int l = 0; l < levels; ++l) {          // level l: l + 1 stackings
    v = v / 2 + v % 2;
    b[(int64_t)l * B + i] = v;
  }
}

}  // namespace

extern "C" int las_cast_bf16(const float* src, int64_t lds_, int rows, int cols, las_bf16* dst, int64_t ldd,
                             int dst_rows, int dst_cols, int transpose, int batch, int64_t src_bstride,
                             int64_t dst_bstride, int src_col_perm_h, void* stream) {
  LAS_REQUIRE(rows >= 0 && cols >= 0 && dst_rows > 0 && dst_cols > 0 && ldd >= dst_cols && batch > 0,
              "las_cast_bf16: bad shape");
  LAS_REQUIRE(src_col_perm_h == 0 || cols == 4 * src_col_perm_h, "las_cast_bf16: src_col_perm_h needs cols == 4*H");
  const int64_t total = (int64_t)dst_rows * dst_cols;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cast_kernel, dim3(blocks, batch), dim3(256), 0, (hipStream_t)stream, src, lds_, rows, cols, dst,
                     ldd, dst_rows, dst_cols, transpose, src_bstride, dst_bstride, src_col_perm_h);
  LAS_LAUNCH_CHECK("cast launch");
  return LAS_OK;
}

extern "C" int las_refresh_images(const las_image_job* jobs_dev, int njobs, void* stream) {
  LAS_REQUIRE(jobs_dev != nullptr && njobs > 0 && njobs <= 65535, "las_refresh_images: bad job table (njobs=%d)", njobs);
  hipLaunchKernelGGL(refresh_images_kernel, dim3(96, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  LAS_LAUNCH_CHECK("refresh images launch");
  return LAS_OK;
}

extern "C" int las_add_cast_bf16(const float* a, int64_t lda, const float* b, int64_t ldb, las_bf16* out, int64_t ldo, int rows, int cols,
                                 void* stream) {
  LAS_REQUIRE(a != nullptr && out != nullptr && rows > 0 && cols > 0, "las_add_cast_bf16: bad arguments");
  int blocks = (int)(((int64_t)rows * cols + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(add_cast_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out, ldo, rows, cols);
  LAS_LAUNCH_CHECK("add cast launch");
  return LAS_OK;
}

extern "C" int las_stream_delay(int microseconds, void* stream) {
  LAS_REQUIRE(microseconds >= 0 && microseconds <= 1000, "las_stream_delay: 0..1000 us (got %d)", microseconds);
  if (microseconds == 0) return LAS_OK;
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100);
  LAS_LAUNCH_CHECK("delay launch");
  return LAS_OK;
}

// ---- a census of the dispatcher's placement, and a way to take CUs away from a launch (tests) ----
namespace {
__global__ void xcd_histogram_kernel(unsigned* counts, int spin) {
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    atomicAdd(counts + (xcc & 7), 1u);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)spin * 100) __builtin_amdgcn_s_sleep(8);      // keep the workgroup resident a little
  }
}
}  // namespace

// diagnostics / tests: `workgroups` one-wave workgroups that each note the XCD they run on (counts: 8 device words, added to) and
// then stay resident for spin_us microseconds holding lds_bytes of LDS -- a census of the dispatcher's placement, and a way to
// take CUs away from a launch that needs them (tests of the persistent kernels' bounded waits under CU pressure)
extern "C" int las_xcd_histogram(uint32_t* counts, int workgroups, int spin_us, int lds_bytes, void* stream) {
  LAS_REQUIRE(counts != nullptr && workgroups > 0 && spin_us >= 0 && spin_us <= 5000000 && lds_bytes >= 0 && lds_bytes <= 160 * 1024,
              "las_xcd_histogram: bad arguments");
  if (lds_bytes > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xcd_histogram_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(xcd_histogram_kernel, dim3(workgroups), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, counts, spin_us);
  LAS_LAUNCH_CHECK("xcd histogram launch");
  return LAS_OK;
}

extern "C" int las_stream_concurrency_probe(void* waiter_stream, void* setter_stream, int32_t* words, int wait_us) {
  LAS_REQUIRE(words != nullptr && wait_us > 0 && wait_us <= 100000, "las_stream_concurrency_probe: two device words, a wait of 1 .. 100000 us");
  int rc = las_check_hip(hipMemsetAsync(words, 0, 2 * sizeof(int32_t), (hipStream_t)waiter_stream), "probe memset");
  if (rc) return rc;
  hipEvent_t ev;
  rc = las_check_hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "probe event");
  if (rc) return rc;
  rc = las_check_hip(hipEventRecord(ev, (hipStream_t)waiter_stream), "probe event record");
  if (!rc) rc = las_check_hip(hipStreamWaitEvent((hipStream_t)setter_stream, ev, 0), "probe stream wait");
  (void)hipEventDestroy(ev);
  if (rc) return rc;
  hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)waiter_stream, words, (long long)wait_us * 100);
  LAS_LAUNCH_CHECK("probe waiter launch");
  hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)setter_stream, words);
  LAS_LAUNCH_CHECK("probe setter launch");
  return LAS_OK;
}

extern "C" int las_fill_many(const las_fill_job* jobs_host, int njobs, void* stream) {
  LAS_REQUIRE(jobs_host != nullptr && njobs > 0 && njobs <= LAS_FILL_MAX_JOBS, "las_fill_many: 1..%d jobs (got %d)", (int)LAS_FILL_MAX_JOBS, njobs);
  FillTable t;
  int64_t most = 0;
  for (int i = 0; i < njobs; ++i) {
    const las_fill_job& j = jobs_host[i];
    LAS_REQUIRE(j.dst != nullptr && j.rows >= 0 && j.cols >= 0 && j.kind >= LAS_FILL_ZERO32 && j.kind <= LAS_FILL_CAST_BF16,
                "las_fill_many: bad job %d", i);
    t.job[i] = j;
    if (j.cols == 0) t.job[i].cols = 1, t.job[i].rows = 0;
    if (j.rows * j.cols > most) most = j.rows * j.cols;
  }
  if (most == 0) return LAS_OK;
  int blocks = (int)((most + 1023) / 1024);          // about four elements per thread
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fill_many_kernel, dim3(blocks, njobs), dim3(256), 0, (hipStream_t)stream, t);
  LAS_LAUNCH_CHECK("fill launch");
  return LAS_OK;
}

static int colsum_chunks(int M) {
  int chunks = (M + 511) / 512;
  return chunks > 256 ? 256 : chunks;
}

extern "C" int las_colsum_bf16(const las_bf16* X, int64_t ldx, int M, int N, float* out, int out_perm_h, void* stream) {
  LAS_REQUIRE(M > 0 && N > 0, "las_colsum_bf16: empty");
  const int chunks = colsum_chunks(M);
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, chunks), dim3(64, 4), 0, (hipStream_t)stream, X, ldx, M, N, out, out_perm_h);
  LAS_LAUNCH_CHECK("colsum launch");
  return LAS_OK;
}

extern "C" size_t las_colsum_ws_bytes(int M, int N) {
  return 4096 + sizeof(float) * (size_t)colsum_chunks(M) * (size_t)((N + 255) / 256 * 256);
}

extern "C" int las_colsum_bf16_ws(const las_bf16* X, int64_t ldx, int M, int N, float* out, int out_perm_h, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  LAS_REQUIRE(M > 0 && N > 0 && N <= 64 * 1024, "las_colsum_bf16_ws: bad shape");
  LAS_REQUIRE(workspace && workspace_bytes >= las_colsum_ws_bytes(M, N) && ((uintptr_t)workspace % 16 == 0),
              "las_colsum_bf16_ws: workspace of las_colsum_ws_bytes(M, N) bytes needed (its first 4096 bytes zero before the first use)");
  if (N % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)X % 16 == 0)) {
    // (the partial rows are [chunks][N rounded up to 256] here: covered by las_colsum_ws_bytes, which rounds N up to 256)
    hipLaunchKernelGGL(colsum_ws8_kernel, dim3((N + 255) / 256, colsum_chunks(M)), dim3(256), 0, (hipStream_t)stream, X, ldx, M, N, out,
                       out_perm_h, static_cast<unsigned*>(workspace), reinterpret_cast<float*>(static_cast<char*>(workspace) + 4096));
    LAS_LAUNCH_CHECK("colsum (workspace, wide) launch");
    return LAS_OK;
  }
  hipLaunchKernelGGL(colsum_ws_kernel, dim3((N + 63) / 64, colsum_chunks(M)), dim3(64, 4), 0, (hipStream_t)stream, X, ldx, M, N, out,
                     out_perm_h, static_cast<unsigned*>(workspace), reinterpret_cast<float*>(static_cast<char*>(workspace) + 4096));
  LAS_LAUNCH_CHECK("colsum (workspace) launch");
  return LAS_OK;
}

extern "C" int las_pyramid_lengths(const int32_t* len_in, int32_t* len_out, int B, void* stream) {
  LAS_REQUIRE(B > 0, "las_pyramid_lengths: B must be positive");
  hipLaunchKernelGGL(pyramid_len_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, len_in, len_out, B, 1);
  LAS_LAUNCH_CHECK("pyramid launch");
  return LAS_OK;
}

extern "C" int las_pyramid_lengths_multi(const int32_t* len_in, int32_t* len_out, int B, int levels, void* stream) {
  LAS_REQUIRE(B > 0 && levels > 0 && levels <= 16, "las_pyramid_lengths_multi: B > 0, 1..16 levels");
  hipLaunchKernelGGL(pyramid_len_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, len_in, len_out, B, levels);
  LAS_LAUNCH_CHECK("pyramid launch");
  return LAS_OK;
}

