// Output projection + sequence cross-entropy + its backward through the projection in ONE launch (round 4):
//   logits   = ctx W_proj + b                      (las/model.py:251-257 Dense(V); ctx = the decoder outputs, [B*U, M] bf16)
//   loss     = sum_{b, t < len_b} CE(logits[b,t], targets[b,t]) / (sum_b len_b + 1e-12)      (model_helper.py:24-30)
//   dlogits  = (softmax - onehot) * grad_scale / sum(w) on the unmasked steps, 0 elsewhere    (bf16: operand of the weight gradients)
//   dctx     = dlogits W_proj^T                                                              (fp32: the decoder backward's input)
// Between the decoder's forward launch and its backward launch the train step used to issue five small launches (projection
// product, a fill, the loss kernel, the product back through the projection, another fill): 0.1 ms of the critical path for
// 1.3 GFLOP.  Here a workgroup owns 16 rows of the [B*U] decoder outputs: the K = M reduction of the projection is split over
// its four waves (operand pieces straight from global memory: W_proj is 128 KB and L2-resident), the partial tiles meet in LDS,
// every wave then normalises four rows (wave reductions), and the 16 x Vp tile of dlogits is the A operand -- from LDS -- of the
// product back through W_proj, 16 column tiles of 16 per wave.  The loss needs no zeroed output in front of the launch: the
// workgroups leave their partial sums in slots of a small persistent workspace and the LAST one to arrive (a counter it resets)
// adds them in a fixed order (no fp32 atomics: the loss is bit-reproducible).
#include "las_common.h"

namespace {

constexpr int PL_ROWS = 16;

template <int VT>          // 16-column tiles of the (padded) vocabulary: Vp = 16 VT <= 128
__global__ __launch_bounds__(256) void proj_ce_kernel(const unsigned short* __restrict__ ctx, int64_t ldc_, const unsigned short* __restrict__ wT,
                                                      const float* __restrict__ bias, const unsigned short* __restrict__ wn,
                                                      const int32_t* __restrict__ targets, int64_t tstride, const int32_t* __restrict__ target_len,
                                                      int B, int U, int V, int M, float grad_scale, float* __restrict__ logits,
                                                      unsigned short* __restrict__ dlogits, float* __restrict__ dctx, int64_t ldd,
                                                      float* __restrict__ loss_out, float* __restrict__ partial, unsigned* __restrict__ counter) {
  constexpr int Vp = 16 * VT;
  constexpr int KA = (Vp + 31) / 32;                 // 32-deep chunks of the product back through the projection
  constexpr int LDL = Vp + 4;
  __shared__ __attribute__((aligned(16))) float red[4][PL_ROWS][LDL];
  __shared__ __attribute__((aligned(16))) unsigned short dl[PL_ROWS][KA * 32 + 8];
  __shared__ float wl[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int BU = B * U, row0 = blockIdx.x * PL_ROWS;

  // ---- logits tile: K = M split over the waves in 32-deep chunks (wave w: chunks w, w + 4, ...) ----
  const int KC = M / 32;
  f32x4 acc[VT];
#pragma unroll
  for (int v = 0; v < VT; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned short* arow = ctx + (int64_t)min(row0 + l15, BU - 1) * ldc_ + 8 * lq;
  for (int kc0 = wave; kc0 < KC; kc0 += 16) {        // four chunks of this wave in flight at a time
    uint4 av[4], bv[4][VT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kc = min(kc0 + 4 * i, KC - 1);
      av[i] = *reinterpret_cast<const uint4*>(arow + kc * 32);
#pragma unroll
      for (int v = 0; v < VT; ++v) bv[i][v] = *reinterpret_cast<const uint4*>(wT + (int64_t)(v * 16 + l15) * M + kc * 32 + 8 * lq);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (kc0 + 4 * i < KC) {
#pragma unroll
        for (int v = 0; v < VT; ++v)
          acc[v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[i]), __builtin_bit_cast(bf16x8, bv[i][v]), acc[v], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VT; ++v)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lq * 4 + r][v * 16 + l15] = acc[v][r];
  for (int i = tid; i < PL_ROWS * (KA * 32 + 8); i += 256) (&dl[0][0])[i] = 0;       // the K padding of the second product
  __syncthreads();

  // ---- loss and d(logits): wave w owns rows 4 w .. 4 w + 3 ----
  float total = 0.f;
  for (int i = lane; i < B; i += 64) total += (float)min(target_len[i], U);
  total = las_wave_sum(total) + 1e-12f;
  const float inv_total = 1.0f / total;
  float local_loss = 0.f;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int rl = wave * 4 + rr, row = row0 + rl;
    if (row >= BU) continue;                         // (wave-uniform)
    const int b = row / U, t = row % U;
    const bool on = t < target_len[b];
    float lg[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = lane + 64 * j;
      lg[j] = -INFINITY;
      if (v < Vp) {
        const float x = red[0][rl][v] + red[1][rl][v] + red[2][rl][v] + red[3][rl][v] + bias[v];
        logits[(int64_t)row * Vp + v] = x;
        if (v < V) lg[j] = x;
      }
    }
    const float mx = las_wave_max(fmaxf(lg[0], lg[1]));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) if (lane + 64 * j < V) sum += __expf(lg[j] - mx);
    sum = las_wave_sum(sum);
    const float lse = mx + __logf(sum);
    const int tgt = targets[(int64_t)b * tstride + t];
    // the target's logit: it sits in lane tgt & 63, slot tgt >> 6
    const float mine = (tgt >> 6) ? lg[1] : lg[0];
    const float lt = __shfl(mine, tgt & 63, 64);
    if (on && lane == 0) local_loss += (lse - lt) * inv_total;
    const float sc = grad_scale * inv_total;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = lane + 64 * j;
      if (v < Vp) {
        const float d = (on && v < V) ? (__expf(lg[j] - lse) - (v == tgt ? 1.f : 0.f)) * sc : 0.f;
        const unsigned short db = las_f2bf(d);
        dlogits[(int64_t)row * Vp + v] = db;
        dl[rl][v] = db;
      }
    }
  }
  if (lane == 0) wl[wave] = local_loss;
  __syncthreads();

  // ---- d(ctx) tile = dlogits [16, Vp] W_proj^T: wave w forms column tiles w, w + 4, ... of the M columns ----
  bf16x8 af[KA];
#pragma unroll
  for (int kc = 0; kc < KA; ++kc) af[kc] = *reinterpret_cast<const bf16x8*>(&dl[l15][kc * 32 + 8 * lq]);
  const int NT = M / 16;
  for (int nt0 = wave; nt0 < NT; nt0 += 16) {
    uint4 bw[4][KA];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nt = min(nt0 + 4 * i, NT - 1);
#pragma unroll
      for (int kc = 0; kc < KA; ++kc)      // (a chunk past the end of a Vp that is no multiple of 32 re-reads inside the row: its A columns are zero)
        bw[i][kc] = *reinterpret_cast<const uint4*>(wn + (int64_t)(nt * 16 + l15) * Vp + min(kc * 32 + 8 * lq, Vp - 8));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nt = nt0 + 4 * i;
      if (nt >= NT) break;                 // (wave-uniform)
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KA; ++kc)
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kc], __builtin_bit_cast(bf16x8, bw[i][kc]), o, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + lq * 4 + r;
        if (row < BU) dctx[(int64_t)row * ldd + nt * 16 + l15] = o[r];
      }
    }
  }

  // ---- the loss: every workgroup leaves its partial sum in ITS slot of the workspace (write-through) and arrives at a counter;
  // the LAST one to arrive adds the slots in a fixed order (lane i: slots i, i + 64, ...; then the wave butterfly) -- the reported
  // loss is the same bits from run to run, as every other sum of the train op -- and resets the counter.  No zeroed output, no
  // fill launch in front of the kernel. ----
  __shared__ int last_one;
  if (tid == 0) {
    const float mine = wl[0] + wl[1] + wl[2] + wl[3];
    __hip_atomic_store(partial + blockIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the slot is at the coherence point before the arrival
    const unsigned n = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_one = (n == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (last_one && wave == 0) {
    float sum = 0.f;
    for (int i = lane; i < (int)gridDim.x; i += 64) sum += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sum = las_wave_sum(sum);
    if (lane == 0) {
      *loss_out = sum;
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

extern "C" int las_proj_ce_supported(int V, int Vp, int M) {
  return V > 0 && Vp >= V && Vp % 16 == 0 && Vp <= 128 && M % 128 == 0 && M >= 128;
}

extern "C" size_t las_proj_ce_workspace_bytes(int B, int U) {
  // arrival counter (zero before the first use; every launch leaves it zero) + one partial-sum slot per workgroup
  return 64 + sizeof(float) * (size_t)(((size_t)B * U + PL_ROWS - 1) / PL_ROWS);
}

extern "C" int las_proj_ce(const las_bf16* ctx, int64_t ld_ctx, const las_bf16* wprojT, const float* bproj, const las_bf16* wproj,
                           const int32_t* targets, int64_t target_stride, const int32_t* target_len, int B, int U, int V, int Vp, int M,
                           float grad_scale, float* logits, las_bf16* dlogits, float* dctx, int64_t ld_dctx, float* loss_out,
                           void* workspace, void* stream) {
  LAS_REQUIRE(ctx && wprojT && bproj && wproj && targets && target_len && logits && dlogits && dctx && loss_out && workspace,
              "las_proj_ce: null argument");
  LAS_REQUIRE(B > 0 && U > 0 && las_proj_ce_supported(V, Vp, M), "las_proj_ce: unsupported shape B=%d U=%d V=%d Vp=%d M=%d "
              "(Vp a multiple of 16 up to 128, M a multiple of 128)", B, U, V, Vp, M);
  LAS_REQUIRE(ld_ctx % 8 == 0 && ld_ctx >= M && ld_dctx >= M && ((uintptr_t)ctx % 16 == 0) && ((uintptr_t)wprojT % 16 == 0) &&
                  ((uintptr_t)wproj % 16 == 0) && ((uintptr_t)workspace % 16 == 0),
              "las_proj_ce: operands must be 16-byte aligned, ld_ctx a multiple of 8");
  const int blocks = (B * U + PL_ROWS - 1) / PL_ROWS;
  unsigned* counter = static_cast<unsigned*>(workspace);
  float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + 64);
  hipStream_t st = (hipStream_t)stream;
#define LAS_PL(VT)                                                                                                            \
  hipLaunchKernelGGL(proj_ce_kernel<VT>, dim3(blocks), dim3(256), 0, st, ctx, ld_ctx, wprojT, bproj, wproj, targets,        \
                     target_stride, target_len, B, U, V, M, grad_scale, logits, dlogits, dctx, ld_dctx, loss_out, partial, counter)
  switch (Vp / 16) {
    case 1: LAS_PL(1); break;
    case 2: LAS_PL(2); break;
    case 3: LAS_PL(3); break;
    case 4: LAS_PL(4); break;
    case 5: LAS_PL(5); break;
    case 6: LAS_PL(6); break;
    case 7: LAS_PL(7); break;
    default: LAS_PL(8); break;
  }
#undef LAS_PL
  LAS_LAUNCH_CHECK("proj ce launch");
  return LAS_OK;
}
