// Shared device/host helpers for liblas_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include "las_host.h"

// One target, stated to the compiler: the fence-free "last workgroup adds up" reductions (optim.hip, decoder.hip)
// order their relaxed stores with s_waitcnt vmcnt(0) because gfx950 acknowledges sc1 write-through stores at the coherence
// point, and the XCD-local exchanges rely on one L2 per XCD; neither is a property of the HIP memory model.  Another target
// must not compile this code silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "liblas_hip is written for gfx950 (MI355X) only"
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;

#define LAS_WAVE 64

int las_check_hip(hipError_t e, const char* what);
// Diagnostics / A-B knobs of the host code: integer environment variables READ ONCE, at their first use in the process (core.hip);
// las_set_knob (the C-ABI's test hook) overrides a value afterwards -- the tests switch slice heights and member counts that way.
int las_knob(const char* name, int default_value);

#define LAS_LAUNCH_CHECK(what)                                   \
  do {                                                           \
    int _rc = las_check_hip(hipGetLastError(), what);            \
    if (_rc) return _rc;                                         \
  } while (0)

// v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE divide is a ten-instruction sequence, and the gate math sits on the
// critical path of every recurrent step.  Limits stay exact: rcp(inf) = 0, rcp(1) = 1.
__device__ __forceinline__ float las_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float las_tanh(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); abs error ~2e-7
  float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ unsigned short las_f2bf(float x) {
  __bf16 b = (__bf16)x;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float las_bf2f(unsigned short u) {
  return __uint_as_float(((unsigned)u) << 16);
}
__device__ __forceinline__ float las_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float las_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// The same reductions on the DPP path (no LDS permutes: a shuffle is a ds_bpermute round trip of ~100 cycles, a DPP operand
// costs nothing): two quad_perm steps, row_half_mirror, row_mirror give every lane its 16-lane row's result, the four rows
// meet through readlane.  Sums add in a different order than las_wave_sum.
template <int CTRL>
__device__ __forceinline__ float las_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float las_quad_sum(float v) {       // every lane: the sum over its quad (lanes 4k .. 4k+3)
  v += las_dpp<0xB1>(v);      // quad_perm:[1,0,3,2]
  v += las_dpp<0x4E>(v);      // quad_perm:[2,3,0,1]
  return v;
}
__device__ __forceinline__ float las_wave_sum_dpp(float v) {
  v = las_quad_sum(v);
  v += las_dpp<0x141>(v);     // row_half_mirror
  v += las_dpp<0x140>(v);     // row_mirror
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
}
__device__ __forceinline__ float las_wave_max_dpp(float v) {
  v = fmaxf(v, las_dpp<0xB1>(v));
  v = fmaxf(v, las_dpp<0x4E>(v));
  v = fmaxf(v, las_dpp<0x141>(v));
  v = fmaxf(v, las_dpp<0x140>(v));
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48))));
}

// counter-based uniform generator shared by the dropout / sampling kernels (rng.hip, decoder.hip)
__device__ __forceinline__ unsigned las_mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// uniform in [0,1) from (seed, stream, 64-bit element index)
__device__ __forceinline__ float las_uniform(unsigned seed, unsigned stream, unsigned long long idx) {
  unsigned h = las_mix32(seed ^ (stream * 0x9E3779B9u));
  h = las_mix32(h ^ (unsigned)idx);
  h = las_mix32(h + (unsigned)(idx >> 32) * 0x85EBCA6Bu + 0x632BE5ABu);
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// N(0,1) draw i of (seed, stream): Box-Muller on the uniforms 2i and 2i+1
__device__ __forceinline__ float las_normal(unsigned seed, unsigned stream, unsigned long long i) {
  const float u1 = fmaxf(las_uniform(seed, stream, i * 2), 1e-12f);
  const float u2 = las_uniform(seed, stream, i * 2 + 1);
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
}

// Fragment-major image of K_h [H, 4H] (las_lstm_pack_recurrent): element i of the packed image comes from
//   K_h[kc*32 + 8*(lane>>4) + j][g*H + ublk*16 + (lane&15)],  i = (((ublk*KC + kc)*4 + g)*64 + lane)*8 + j,  KC = H/32.
__device__ __forceinline__ int64_t las_pack_recurrent_src(int64_t i, int H) {
  const int KC = H / 32;
  int64_t r = i;
  const int j = (int)(r % 8); r /= 8;
  const int lane = (int)(r % 64); r /= 64;
  const int g = (int)(r % 4); r /= 4;
  const int kc = (int)(r % KC); r /= KC;
  const int ublk = (int)r;
  const int k = kc * 32 + 8 * (lane >> 4) + j;
  const int col = g * H + ublk * 16 + (lane & 15);
  return (int64_t)k * 4 * H + col;
}

// Fragment-major image of K_x = kernel[0:D, :] over `chunks` 32-deep K chunks (las_lstm_pack_input): element i comes from
//   K_x[k][g*H + ublk*16 + (lane&15)],  k = kc*32 + 8*(lane>>4) + j,  i = (((ublk*chunks + kc)*4 + g)*64 + lane)*8 + j,
// and is zero for k >= D.  Returns the source index, or -1 for the zero padding.
__device__ __forceinline__ int64_t las_pack_input_src(int64_t i, int D, int H, int chunks) {
  int64_t r = i;
  const int j = (int)(r % 8); r /= 8;
  const int lane = (int)(r % 64); r /= 64;
  const int g = (int)(r % 4); r /= 4;
  const int kc = (int)(r % chunks); r /= chunks;
  const int ublk = (int)r;
  const int k = kc * 32 + 8 * (lane >> 4) + j;
  return k < D ? (int64_t)k * 4 * H + g * H + ublk * 16 + (lane & 15) : -1;
}
