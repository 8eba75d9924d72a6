// Micro-benchmark: one-way latency of the tagged-granule hand-off between two workgroups.
// Build: hipcc -O3 --offload-arch=gfx950 pingpong.hip -o pingpong ; run: ./pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
__device__ __forceinline__ u64 gl(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// mode bit0: plain store (else agent-scope atomic store); bit1: s_sleep between polls; lanes: granules per side
__global__ void pingpong(u64* buf, int iters, int partner_block, int mode, int lanes, unsigned* xcc_out, long long* cycles) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner_block ? 1 : -1);
  if (me < 0) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) xcc_out[me] = xcc & 0xf;
  u64* mine = buf + me * 4096;
  const u64* theirs = buf + (1 - me) * 4096;
  const int lane = threadIdx.x;
  const bool on = lane < lanes;
  bool failed = false;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 1; i <= iters; ++i) {
    if (me == 0) {
      if (on) { u64 x = ((u64)i << 32) | lane; if (mode & 1) mine[lane] = x; else __hip_atomic_store(mine + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    // wait for the partner's tag i (side 1 waits for side 0's i, then answers i; side 0 waits for the answer)
    bool ok;
    unsigned spins = 0;
    do {
      ok = !on || (unsigned)(gl(theirs + lane) >> 32) == (unsigned)i;
      ok = __all(ok);
      if (!ok && (mode & 2)) __builtin_amdgcn_s_sleep(1);
      if (!ok && ++spins > 200000u) { failed = true; break; }     // never hang the box (plain stores across XCDs may never land)
    } while (!ok);
    if (failed) break;
    if (me == 1) {
      if (on) { u64 x = ((u64)i << 32) | lane; if (mode & 1) mine[lane] = x; else __hip_atomic_store(mine + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[me] = failed ? -1 : t1 - t0;
}
int main() {
  u64* buf; unsigned* xcc; long long* cyc;
  hipMalloc(&buf, 8192 * 8); hipMalloc(&xcc, 8); hipMalloc(&cyc, 16);
  const int iters = 20000;
  for (int partner : {8, 1, 4}) for (int lanes : {1, 64}) for (int mode = 0; mode < 4; ++mode) {
    hipMemset(buf, 0, 8192 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(pingpong, dim3(16), dim3(64), 0, 0, buf, iters, partner, mode, lanes, xcc, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned hx[2]; hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost);
    long long hc[2]; hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
    printf("partner_block=%d xcc=(%u,%u) lanes=%d store=%s sleep=%d : one-way hop %.0f ns%s\n", partner, hx[0], hx[1], lanes,
           (mode & 1) ? "plain" : "atomic", (mode >> 1) & 1, ms * 1e6 / (2.0 * iters), (hc[0] < 0 || hc[1] < 0) ? "  (TIMED OUT)" : "");
  }
  return 0;
}
