// (1) does a 16-byte raw buffer load (sc1) return the two 8-byte granules a peer stored (across launches)?
// (2) does it see a peer's store WHILE polling inside one launch (plain store by the writer = the `local` flavour of lstm.hip, and the
//     write-through atomic store), or does it keep hitting a stale L1 line?
// hipcc --offload-arch=gfx950 -O3 wideload.hip -o wideload && ./wideload
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void writer(u64* p, int n, unsigned tag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) __hip_atomic_store(p + i, ((u64)tag << 32) | (unsigned)(i * 7 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void reader(const u64* p, int n, unsigned tag, unsigned* bad, unsigned* sample) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  int i = blockIdx.x * blockDim.x + threadIdx.x;       // pair index
  if (2 * i + 1 < n) {
    u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)i * 16u, 0, 16);
    u64 a = __hip_atomic_load(p + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(p + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = w.x == (unsigned)a && w.y == (unsigned)(a >> 32) && w.z == (unsigned)b && w.w == (unsigned)(b >> 32) && w.y == tag && w.w == tag;
    if (!ok) atomicAdd(bad, 1u);
    if (i == 5) { sample[0] = w.x; sample[1] = w.y; sample[2] = w.z; sample[3] = w.w; sample[4] = (unsigned)a; sample[5] = (unsigned)(a >> 32); }
  }
}
// block 0: waits, then stores two granules (mode 0: plain stores, 1: agent-scope atomic stores); block 8 (same XCD under round-robin)
// and block 1 (another XCD) poll them with the given aux bits; out[b] = iterations until both tags were seen (0: never)
template <int AUX>
__global__ void poll_test(u64* p, int mode, unsigned* out) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  if (threadIdx.x != 0) return;
  if (blockIdx.x == 0) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 5000) __builtin_amdgcn_s_sleep(8);       // 50 us at 100 MHz
    if (mode == 0) { p[0] = ((u64)5 << 32) | 11u; p[1] = ((u64)5 << 32) | 12u; }
    else { __hip_atomic_store(p, ((u64)5 << 32) | 11u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p + 1, ((u64)5 << 32) | 12u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  } else {
    volatile u64 warm = *(volatile u64*)p;      // the line is in this CU's L1 now
    (void)warm;
    unsigned it = 0;
    for (unsigned k = 1; k < 2000000; ++k) {
      u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, AUX);
      if (w.y == 5u && w.w == 5u) { it = k; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    out[blockIdx.x] = it;
  }
}
int main() {
  const int n = 1 << 16;
  u64* p; unsigned *bad, *sample;
  hipMalloc(&p, n * 8); hipMalloc(&bad, 4); hipMalloc(&sample, 128);
  hipMemset(p, 0, n * 8); hipMemset(bad, 0, 4); hipMemset(sample, 0, 128);
  writer<<<n / 256, 256>>>(p, n, 77u);
  reader<<<n / 512, 256>>>(p, n, 77u, bad, sample);
  unsigned hb = 0, hs[32];
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hs, sample, 32, hipMemcpyDeviceToHost);
  printf("mismatching pairs: %u of %d; pair 5: wide %u %u %u %u, narrow %u %u\n", hb, n / 2, hs[0], hs[1], hs[2], hs[3], hs[4], hs[5]);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(p, 0, 64); hipMemset(sample, 0, 128); hipDeviceSynchronize();
    poll_test<16><<<16, 64>>>(p, mode, sample); hipDeviceSynchronize();
    hipMemcpy(hs, sample, 128, hipMemcpyDeviceToHost);
    printf("in-launch polling, writer %s, aux sc1:           iterations until seen, blocks 1..15:", mode ? "atomic store" : "plain store ");
    for (int b = 1; b < 16; ++b) printf(" %u", hs[b]);
    printf("\n");
    hipMemset(p, 0, 64); hipMemset(sample, 0, 128); hipDeviceSynchronize();
    poll_test<(int)0x80000010u><<<16, 64>>>(p, mode, sample); hipDeviceSynchronize();
    hipMemcpy(hs, sample, 128, hipMemcpyDeviceToHost);
    printf("in-launch polling, writer %s, aux volatile|sc1:  iterations until seen, blocks 1..15:", mode ? "atomic store" : "plain store ");
    for (int b = 1; b < 16; ++b) printf(" %u", hs[b]);
    printf("\n");
  }
  return hb != 0;
}
