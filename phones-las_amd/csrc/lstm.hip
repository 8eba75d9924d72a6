// Listener recurrence: tf.nn.(bidirectional_)dynamic_rnn over LSTMCell (las/ops.py:10-46) as two
// persistent kernels (forward, backward-in-time) on gfx950.
//
// Work decomposition: the batch is cut into slices of 16 utterances (the M of
// v_mfma_f32_16x16x32_bf16); one 256-thread workgroup owns one (slice, direction) chain for the whole
// sequence, so there is NO inter-workgroup communication.  Wave w owns hidden units
// [w*H/4, (w+1)*H/4) for all four gates, so i, j, f, o of one (utterance, unit) sit in the same lane's
// accumulators (C/D layout: col = lane&15 -> unit, row = (lane>>4)*4 + reg -> utterance) and the gate
// math needs no cross-lane traffic.  h_t crosses waves once per step through a double-buffered 16 x H
// bf16 LDS tile (one barrier per step) from which every wave reads its A fragments.  K_h is streamed
// from L2 each step in MFMA-fragment-major order (forward) / natural order (backward).
#include "las_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// pack K_h [H,4H] fp32 -> fragment-major bf16:
//   packed[((((w*KC + kc)*4 + g)*UB + ub)*64 + lane)*8 + j] =
//       K_h[kc*32 + 8*(lane>>4) + j][g*H + w*(H/4) + ub*16 + (lane&15)]
// ------------------------------------------------------------------------------------------------
__global__ void pack_recurrent_kernel(const float* kh, int H, unsigned short* packed) {
  const int UB = H / 64, KC = H / 32;
  const int64_t total = (int64_t)H * 4 * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int j = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int ub = (int)(r % UB); r /= UB;
    const int g = (int)(r % 4); r /= 4;
    const int kc = (int)(r % KC); r /= KC;
    const int w = (int)r;
    const int k = kc * 32 + 8 * (lane >> 4) + j;
    const int col = g * H + w * (H / 4) + ub * 16 + (lane & 15);
    packed[i] = las_f2bf(kh[(int64_t)k * 4 * H + col]);
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256, 1) void lstm_fwd_kernel(float* __restrict__ xproj, const unsigned short* __restrict__ wpacked,
                                                          const int32_t* __restrict__ length, unsigned short* __restrict__ y,
                                                          float* __restrict__ cbuf, float* __restrict__ c_last,
                                                          float* __restrict__ h_last, int B, int T, int ndir) {
  constexpr int UB = H / 64, KC = H / 32, HS = H + 8;
  __shared__ __attribute__((aligned(16))) unsigned short hlds[2][16][HS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slice = blockIdx.x, dir = blockIdx.y;
  const int l15 = lane & 15, lq = lane >> 4;

  int len[4], bidx[4];
  int smax = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bidx[r] = slice * 16 + lq * 4 + r;
    len[r] = (bidx[r] < B) ? min(length[bidx[r]], T) : 0;
    smax = max(smax, len[r]);
  }
  smax = (int)las_wave_max((float)smax);

  float c[UB][4], h[UB][4];
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int r = 0; r < 4; ++r) { c[ub][r] = 0.f; h[ub][r] = 0.f; }

  for (int i = tid; i < 16 * HS; i += 256) (&hlds[0][0][0])[i] = 0;
  __syncthreads();

  const unsigned short* wp = wpacked + (int64_t)dir * H * 4 * H + (int64_t)wave * (KC * 4 * UB * 512) + lane * 8;
  const int64_t xrow = (int64_t)ndir * 4 * H;   // xproj row stride
  const int64_t yrow = (int64_t)ndir * H;
  const int unit0 = wave * (H / 4) + l15;

  int cur = 0;
  for (int s = 0; s < smax; ++s) {
    bool act[4];
    int64_t rowoff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      act[r] = s < len[r];
      const int pos = dir == 0 ? s : len[r] - 1 - s;
      rowoff[r] = act[r] ? ((int64_t)bidx[r] * T + pos) : 0;
    }
    // x_t K_x + b for this step: issued now, consumed after the MFMA loop
    float xp[4][UB][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ub = 0; ub < UB; ++ub)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          xp[g][ub][r] = act[r] ? xproj[rowoff[r] * xrow + dir * 4 * H + g * H + unit0 + ub * 16] : 0.f;

    f32x4 acc[4][UB];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ub = 0; ub < UB; ++ub) acc[g][ub] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 2
    for (int kc = 0; kc < KC; ++kc) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(&hlds[cur][l15][kc * 32 + 8 * lq]);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(wp + (int64_t)((kc * 4 + g) * UB + ub) * 512);
          acc[g][ub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[g][ub], 0, 0, 0);
        }
    }

#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      const int unit = unit0 + ub * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gi = las_sigmoid(acc[0][ub][r] + xp[0][ub][r]);
        const float gj = las_tanh(acc[1][ub][r] + xp[1][ub][r]);
        const float gf = las_sigmoid(acc[2][ub][r] + xp[2][ub][r] + 1.0f);
        const float go = las_sigmoid(acc[3][ub][r] + xp[3][ub][r]);
        const float cn = gf * c[ub][r] + gi * gj;
        const unsigned short hb = las_f2bf(go * las_tanh(cn));
        if (act[r]) {
          float* gp = xproj + rowoff[r] * xrow + dir * 4 * H + unit;
          gp[0] = gi; gp[H] = gj; gp[2 * H] = gf; gp[3 * H] = go;
          cbuf[rowoff[r] * yrow + dir * H + unit] = cn;
          y[rowoff[r] * yrow + dir * H + unit] = hb;
          c[ub][r] = cn;
          h[ub][r] = las_bf2f(hb);
        }
        hlds[cur ^ 1][lq * 4 + r][unit] = act[r] ? hb : las_f2bf(h[ub][r]);
      }
    }
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (bidx[r] < B) {
        const int64_t o = ((int64_t)dir * B + bidx[r]) * H + unit0 + ub * 16;
        c_last[o] = c[ub][r];
        h_last[o] = h[ub][r];
      }
}

// ------------------------------------------------------------------------------------------------
// backward in time
// ------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256, 1) void lstm_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ cbuf,
                                                          const float* __restrict__ dy, const float* __restrict__ dc_last,
                                                          const float* __restrict__ dh_last, const unsigned short* __restrict__ kh,
                                                          const int32_t* __restrict__ length, unsigned short* __restrict__ dz,
                                                          int B, int T, int ndir) {
  constexpr int UB = H / 64, KC = (4 * H) / 32, ZS = 4 * H + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* dzl = reinterpret_cast<unsigned short*>(smem);   // [2][16][ZS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slice = blockIdx.x, dir = blockIdx.y;
  const int l15 = lane & 15, lq = lane >> 4;

  int len[4], bidx[4];
  int smax = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bidx[r] = slice * 16 + lq * 4 + r;
    len[r] = (bidx[r] < B) ? min(length[bidx[r]], T) : 0;
    smax = max(smax, len[r]);
  }
  smax = (int)las_wave_max((float)smax);

  const int unit0 = wave * (H / 4) + l15;
  float dc[UB][4], dh[UB][4];
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = bidx[r] < B;
      const int64_t o = ((int64_t)dir * B + (ok ? bidx[r] : 0)) * H + unit0 + ub * 16;
      dc[ub][r] = (ok && dc_last) ? dc_last[o] : 0.f;
      dh[ub][r] = (ok && dh_last) ? dh_last[o] : 0.f;
    }

  const int64_t grow = (int64_t)ndir * 4 * H;
  const int64_t yrow = (int64_t)ndir * H;
  const unsigned short* khd = kh + (int64_t)dir * H * 4 * H;

  int cur = 0;
  for (int s = smax - 1; s >= 0; --s) {
    unsigned short* zl = dzl + cur * 16 * ZS;
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      const int unit = unit0 + ub * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool act = s < len[r];
        float di = 0.f, dj = 0.f, df = 0.f, dov = 0.f;
        if (act) {
          const int pos = dir == 0 ? s : len[r] - 1 - s;
          const int64_t ro = (int64_t)bidx[r] * T + pos;
          const float* gp = gates + ro * grow + dir * 4 * H + unit;
          const float gi = gp[0], gj = gp[H], gf = gp[2 * H], go = gp[3 * H];
          const float ct = cbuf[ro * yrow + dir * H + unit];
          float cp = 0.f;
          if (s > 0) {
            const int64_t rp = (int64_t)bidx[r] * T + (dir == 0 ? pos - 1 : pos + 1);
            cp = cbuf[rp * yrow + dir * H + unit];
          }
          const float dht = dy[ro * yrow + dir * H + unit] + dh[ub][r];
          const float tc = las_tanh(ct);
          dov = dht * tc * go * (1.f - go);
          const float dct = dc[ub][r] + dht * go * (1.f - tc * tc);
          di = dct * gj * gi * (1.f - gi);
          dj = dct * gi * (1.f - gj * gj);
          df = dct * cp * gf * (1.f - gf);
          dc[ub][r] = dct * gf;
          unsigned short* zp = dz + ro * grow + dir * 4 * H + unit;
          zp[0] = las_f2bf(di); zp[H] = las_f2bf(dj); zp[2 * H] = las_f2bf(df); zp[3 * H] = las_f2bf(dov);
        }
        unsigned short* zr = zl + (lq * 4 + r) * ZS + unit;
        zr[0] = las_f2bf(di); zr[H] = las_f2bf(dj); zr[2 * H] = las_f2bf(df); zr[3 * H] = las_f2bf(dov);
      }
    }
    __syncthreads();

    // dh_{t-1} = dz_t * K_h^T : A = dz tile [16, 4H] from LDS, B[k][n] = K_h[n][k] (natural rows)
    f32x4 acc[UB];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) acc[ub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int kc = 0; kc < KC; ++kc) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(zl + l15 * ZS + kc * 32 + 8 * lq);
#pragma unroll
      for (int ub = 0; ub < UB; ++ub) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(khd + (int64_t)(unit0 + ub * 16) * 4 * H + kc * 32 + 8 * lq);
        acc[ub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[ub], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (s < len[r]) dh[ub][r] = acc[ub][r];
    cur ^= 1;
  }
}

template <int H>
int launch_fwd(float* xproj, const las_bf16* wp, const int32_t* length, las_bf16* y, float* cbuf, float* c_last,
               float* h_last, int B, int T, int ndir, hipStream_t st) {
  dim3 grid((B + 15) / 16, ndir);
  hipLaunchKernelGGL((lstm_fwd_kernel<H>), grid, dim3(256), 0, st, xproj, wp, length, y, cbuf, c_last, h_last, B, T, ndir);
  LAS_LAUNCH_CHECK("lstm fwd launch");
  return LAS_OK;
}

template <int H>
int launch_bwd(const float* gates, const float* cbuf, const float* dy, const float* dc_last, const float* dh_last,
               const las_bf16* kh, const int32_t* length, las_bf16* dz, int B, int T, int ndir, hipStream_t st) {
  dim3 grid((B + 15) / 16, ndir);
  const size_t lds = (size_t)2 * 16 * (4 * H + 8) * sizeof(unsigned short);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((lstm_bwd_kernel<H>), grid, dim3(256), lds, st, gates, cbuf, dy, dc_last, dh_last, kh, length, dz, B, T, ndir);
  LAS_LAUNCH_CHECK("lstm bwd launch");
  return LAS_OK;
}

}  // namespace

extern "C" int las_lstm_pack_recurrent(const float* kernel_h, int H, las_bf16* packed, void* stream) {
  LAS_REQUIRE(H >= 64 && H % 64 == 0, "las_lstm_pack_recurrent: num_units must be a multiple of 64 (got %d)", H);
  hipLaunchKernelGGL(pack_recurrent_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, kernel_h, H, packed);
  LAS_LAUNCH_CHECK("pack launch");
  return LAS_OK;
}

extern "C" int las_lstm_recurrent_fwd(float* xproj, const las_bf16* wpacked, const int32_t* length, las_bf16* y,
                                      float* cbuf, float* c_last, float* h_last, int B, int T, int H, int ndir,
                                      void* stream) {
  LAS_REQUIRE(B > 0 && T > 0 && (ndir == 1 || ndir == 2), "las_lstm_recurrent_fwd: bad shape B=%d T=%d ndir=%d", B, T, ndir);
  hipStream_t st = (hipStream_t)stream;
  int rc = las_check_hip(hipMemsetAsync(y, 0, (size_t)B * T * ndir * H * sizeof(las_bf16), st), "memset y");
  if (rc) return rc;
  switch (H) {
    case 64: return launch_fwd<64>(xproj, wpacked, length, y, cbuf, c_last, h_last, B, T, ndir, st);
    case 128: return launch_fwd<128>(xproj, wpacked, length, y, cbuf, c_last, h_last, B, T, ndir, st);
    case 256: return launch_fwd<256>(xproj, wpacked, length, y, cbuf, c_last, h_last, B, T, ndir, st);
    case 512: return launch_fwd<512>(xproj, wpacked, length, y, cbuf, c_last, h_last, B, T, ndir, st);
    default: break;
  }
  las_set_error("las_lstm_recurrent_fwd: num_units %d not in {64,128,256,512}", H);
  return LAS_ERR_ARG;
}

extern "C" int las_lstm_recurrent_bwd(const float* gates, const float* cbuf, const float* dy, const float* dc_last,
                                      const float* dh_last, const las_bf16* kh_bf16, const int32_t* length,
                                      las_bf16* dz, int B, int T, int H, int ndir, void* stream) {
  LAS_REQUIRE(B > 0 && T > 0 && (ndir == 1 || ndir == 2), "las_lstm_recurrent_bwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  int rc = las_check_hip(hipMemsetAsync(dz, 0, (size_t)B * T * ndir * 4 * H * sizeof(las_bf16), st), "memset dz");
  if (rc) return rc;
  switch (H) {
    case 64: return launch_bwd<64>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, B, T, ndir, st);
    case 128: return launch_bwd<128>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, B, T, ndir, st);
    case 256: return launch_bwd<256>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, B, T, ndir, st);
    case 512: return launch_bwd<512>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, B, T, ndir, st);
    default: break;
  }
  las_set_error("las_lstm_recurrent_bwd: num_units %d not in {64,128,256,512}", H);
  return LAS_ERR_ARG;
}
