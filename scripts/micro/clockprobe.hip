// Micro-benchmark: what clock does a LOW-OCCUPANCY persistent kernel (32 workgroups x 4 waves, like the recurrent
// kernels) actually run at, and what do an MFMA, a transcendental and an LDS round trip cost there?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256) void probe(float* out, long long* res, int iters) {
  __shared__ float lds[1024];
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
  f32x4 acc[4] = {};
  lds[threadIdx.x] = lane;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[g], 0, 0, 0);
  asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float x = acc[0][0] * 1e-30f + 0.5f;
  for (int i = 0; i < iters; ++i) x = __builtin_amdgcn_rcpf(1.0f + __expf(-x));     // dependent exp+rcp chain
  asm volatile("s_nop 0" :: "v"(x));
  long long t2 = __builtin_amdgcn_s_memtime();
  int idx = lane;
  for (int i = 0; i < iters; ++i) idx = (int)lds[idx & 1023];                      // dependent LDS reads
  asm volatile("s_nop 0" :: "v"(idx));
  long long t3 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier();
  long long t4 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { res[0] = t1 - t0; res[1] = r1 - r0; res[2] = t2 - t1; res[3] = t3 - t2; res[4] = t4 - t3; }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + x + idx;
}
int main() {
  float* out; long long* res;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&res, 64);
  const int iters = 20000;
  for (int blocks : {32, 32, 256, 1024}) {
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, res, iters);
    hipDeviceSynchronize();
    long long h[5]; hipMemcpy(h, res, 40, hipMemcpyDeviceToHost);
    double clk = (double)h[0] / (double)h[1] * 100e6;
    printf("blocks %4d: s_memtime/s_memrealtime -> %.0f MHz; MFMA 16x16x32 %.1f ticks each; exp+rcp dependent pair %.1f ticks; LDS dependent read %.1f ticks; 4-wave s_barrier %.1f ticks\n",
           blocks, clk / 1e6, (double)h[0] / (4.0 * iters), (double)h[2] / iters, (double)h[3] / iters, (double)h[4] / iters);
  }
  return 0;
}
