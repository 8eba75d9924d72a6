// Listener recurrence: tf.nn.(bidirectional_)dynamic_rnn over LSTMCell (las/ops.py:10-46) as two
// persistent, weight-stationary kernels (forward, backward-in-time) on gfx950.
//
// Decomposition.  The batch is cut into slices of 16 utterances (the M of v_mfma_f32_16x16x32_bf16).
// One (slice, direction) pair is a serial chain of T dependent steps; it is run by a GROUP of G
// co-resident 256-thread workgroups ("members").  Member m owns hidden units [m*H/G, (m+1)*H/G): each of
// its 4 waves owns H/(4G) units for all four gates and keeps the matching columns of K_h (forward) /
// rows of K_h (backward) in REGISTERS for the whole sequence as ready-made MFMA B fragments — the
// recurrent weights are read from memory once per launch, not once per step.  i, j, f, o of one
// (utterance, unit) land in the same lane (C/D layout col = lane&15 -> unit, row = (lane>>4)*4+reg ->
// utterance), so the gate math needs no cross-lane traffic.
//
// Per step the members all-gather what the next product needs (h_t: 16 x H bf16 forward; dz_t:
// 16 x 4H bf16 backward).  Inside a member it goes through a double-buffered LDS tile (one barrier per
// step); between members through 8-byte {tag, 2 x bf16} granules in a global exchange buffer, stored
// and polled with agent-scope relaxed atomics (write-through, L1-bypassing): the data is its own flag
// (cdna_hip_programming.md Guideline 16, form R2), tag = step epoch, buffer zeroed by a memset node at
// every launch, two parity slots so a fast member cannot overwrite what a slow one still reads.
// Results never depend on workgroup placement; blockIdx = group + member*ngroups only makes the members
// of a group share an XCD under round-robin dispatch when ngroups % 8 == 0 (speed).  Every spin is
// bounded: on timeout the kernel sets a status word and returns.
#include "las_common.h"

namespace {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

constexpr unsigned SPIN_LIMIT = 1u << 20;

// `local`: every member of the group runs on the SAME XCD (established at kernel start, see xcd_colocated): the
// XCD's L2 is then the coherence point, so a plain store (line stays in L2) plus the peers' L1-bypassing loads is
// enough and a poll costs an L2 hit instead of a fabric round trip.  Otherwise the write-through agent-scope store.
__device__ __forceinline__ void granule_store(u64* p, unsigned tag, unsigned value, bool local) {
  const u64 x = ((u64)tag << 32) | value;
  if (local) *p = x;
  else __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 granule_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Do all G members of this group sit on one XCD?  Each member publishes its XCC id with the placement-independent
// write-through store, then reads everybody's.  Purely a speed decision: the result only selects the store flavour.
// Call from all threads; returns the same value in every thread of the workgroup (fail -> false).
template <int G>
__device__ bool xcd_colocated(u64* table, int member, int* lds_flag, unsigned* status) {
  if (G == 1) return false;
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    __hip_atomic_store(table + member, ((u64)1 << 32) | (xcc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool same = true;
    for (int m = 0; m < G; ++m) {
      u64 v = 0;
      unsigned spins = 0;
      do {
        v = __hip_atomic_load(table + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 32) == 1) break;
        __builtin_amdgcn_s_sleep(2);
      } while (++spins < (1u << 20));
      same = same && ((v >> 32) == 1) && ((unsigned)v == xcc + 1);
    }
    *lds_flag = same ? 1 : 0;
    atomicAdd(status + (same ? 1 : 2), 1u);      // diagnostics: members that took the local / the fabric flavour
  }
  __syncthreads();
  return *lds_flag != 0;
}

// number of cooperating workgroups per chain
// H <= 128: one workgroup; H = 256: 4 x 64 units; H = 512: 16 x 32 units with the K dimension split over wave pairs
// (a wave's register-resident weight block is 16 units x 4 gates x K/KS: 128 VGPRs in every case)
__host__ __device__ constexpr int coop_members(int H) { return H <= 128 ? 1 : (H == 256 ? 4 : H / 32); }
__host__ __device__ constexpr int k_split(int H) { return H > 256 ? 2 : 1; }

// ------------------------------------------------------------------------------------------------
// pack K_h [H,4H] fp32 -> MFMA-B-fragment-major bf16, grouped by 16-unit block:
//   packed[(((ublk*KC + kc)*4 + g)*64 + lane)*8 + j] = K_h[kc*32 + 8*(lane>>4) + j][g*H + ublk*16 + (lane&15)]
// ------------------------------------------------------------------------------------------------
__global__ void pack_recurrent_kernel(const float* kh, int H, unsigned short* packed) {
  const int KC = H / 32;
  const int64_t total = (int64_t)H * 4 * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int j = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int g = (int)(r % 4); r /= 4;
    const int kc = (int)(r % KC); r /= KC;
    const int ublk = (int)r;
    const int k = kc * 32 + 8 * (lane >> 4) + j;
    const int col = g * H + ublk * 16 + (lane & 15);
    packed[i] = las_f2bf(kh[(int64_t)k * 4 * H + col]);
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Workgroup size: 4 compute waves, plus one PREFETCH wave when the chain is shared by several workgroups.  The
// prefetch wave only touches (loads and discards) the cache lines the compute waves will need PF_DIST steps
// later, so their own loads hit L2 instead of HBM: a wave's vmcnt retires in order, so a slow HBM load issued by
// a compute wave would delay the return of its next inter-workgroup poll.
__host__ __device__ constexpr int rec_threads(int H) { return coop_members(H) > 1 ? 320 : 256; }
constexpr int PF_DIST = 3;

template <int H>
__global__ __launch_bounds__(rec_threads(H)) void lstm_fwd_kernel(float* __restrict__ xproj, const unsigned short* __restrict__ wpacked,
                                                          const int32_t* __restrict__ length, unsigned short* __restrict__ y,
                                                          float* __restrict__ cbuf, float* __restrict__ c_last,
                                                          float* __restrict__ h_last, u64* __restrict__ exch,
                                                          unsigned* __restrict__ status, int B, int T, int ndir, int ngroups) {
  constexpr int G = coop_members(H);
  constexpr int HS = H / G;            // units per member
  constexpr int KS = k_split(H);       // ways the K dimension is split over waves
  constexpr int NUB = HS / 16;         // 16-unit blocks of this member
  constexpr int UB = (NUB * KS) / 4 > 0 ? (NUB * KS) / 4 : 1;   // blocks per wave (4 compute waves)
  constexpr int KC = H / 32;
  constexpr int KCW = KC / KS;         // k-chunks a wave owns
  constexpr int LS = H + 8;            // LDS row stride (elements)
  constexpr int NGRAN = 8 * HS;        // granules a member publishes per step (2 rows x 1 unit each)
  __shared__ __attribute__((aligned(16))) unsigned short hlds[2][16][LS];
  __shared__ int fail_flag;
  __shared__ __attribute__((aligned(16))) float pf_scratch[256];   // 1 KiB sink of the prefetch wave's LDS-DMAs
  __shared__ __attribute__((aligned(16))) float red[KS > 1 ? 2 * 4 * 64 * 4 : 4];   // partial sums of the upper K half

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gstride = (ngroups + 7) & ~7;        // members of a group are 8k blocks apart: one XCD under round-robin
  const int group = blockIdx.x % gstride, member = blockIdx.x / gstride;
  if (group >= ngroups) return;
  const int nslices = ngroups / ndir;
  const int slice = group % nslices, dir = group / nslices;
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

  // register-resident B fragments of this wave's K_h columns
  // wave -> (first unit block, K half): KS = 1: 4 waves x UB blocks, all of K; KS = 2: wave&1 = block, wave>>1 = K half
  const int wblk = KS > 1 ? (wave & 1) : wave * UB;
  const int kh = KS > 1 ? (wave >> 1) & 1 : 0;
  const bool lead = (kh == 0);          // the wave that finishes the step for its units
  bf16x8 wf[UB][KCW][4];
  if (wave < 4)
#pragma unroll
  for (int ub = 0; ub < UB; ++ub) {
    const int ublk = member * NUB + wblk + ub;
#pragma unroll
    for (int kc = 0; kc < KCW; ++kc)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        wf[ub][kc][g] = *reinterpret_cast<const bf16x8*>(wpacked + (int64_t)dir * H * 4 * H +
                                                          ((int64_t)((ublk * KC + kh * KCW + kc) * 4 + g) * 64 + lane) * 8);
  }

  float c[UB][4], h[UB][4];
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int r = 0; r < 4; ++r) { c[ub][r] = 0.f; h[ub][r] = 0.f; }

  for (int i = tid; i < 2 * 16 * LS; i += rec_threads(H)) (&hlds[0][0][0])[i] = 0;
  if (tid == 0) fail_flag = 0;
  __syncthreads();
  __shared__ int colo_flag;
  const bool local = xcd_colocated<G>(exch + (int64_t)2 * ngroups * G * NGRAN + (int64_t)group * G, member, &colo_flag, status);

  const int64_t xrow = (int64_t)ndir * 4 * H;
  const int64_t yrow = (int64_t)ndir * H;
  const int unit0 = member * HS + wblk * 16 + l15;            // + ub*16
  u64* ex_group = exch + (int64_t)group * G * NGRAN;           // + parity*ngroups*G*NGRAN + member*NGRAN

  if constexpr (G > 1) {
    if (wave == 4) {
      // prefetch wave: per step and utterance one 1-KiB LDS-DMA (global_load_lds, 16 B per lane) of this member's
      // gate-interleaved xproj span into a scratch tile nobody reads: no VGPR destination, nothing to wait for.
      // It uses the raw s_barrier (a __syncthreads() would drain the DMAs first).
      const int mylen = (slice * 16 + l15 < B) ? min(length[slice * 16 + l15], T) : 0;
      for (int s = 0; s < smax; ++s) {
        const int sp = s + PF_DIST;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int ll = __builtin_amdgcn_readlane(mylen, rr);
          if (sp < ll) {
            const int pos = dir == 0 ? sp : ll - 1 - sp;
            const float* src = xproj + ((int64_t)(slice * 16 + rr) * T + pos) * xrow + dir * 4 * H + member * HS * 4 +
                               (lane % (HS > 64 ? 64 : HS)) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                             (__attribute__((address_space(3))) void*)(pf_scratch), 16, 0, 0);
          }
        }
        __builtin_amdgcn_s_barrier();
        if (*reinterpret_cast<volatile int*>(&fail_flag)) break;
        if constexpr (KS > 1) __builtin_amdgcn_s_barrier();      // the compute waves' partial-sum hand-off barrier
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
  }

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
    // x_t K_x + b of this step: issued now, consumed after the MFMAs
    float xp[4][UB][4];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act[r] && lead) v = *reinterpret_cast<const float4*>(xproj + rowoff[r] * xrow + dir * 4 * H + (unit0 + ub * 16) * 4);
        xp[0][ub][r] = v.x; xp[1][ub][r] = v.y; xp[2][ub][r] = v.z; xp[3][ub][r] = v.w;
      }

    // all-gather h_{s-1}: peers' slices arrive as granules tagged with epoch s
    if constexpr (G > 1) if (s > 0) {
      const u64* src = ex_group + (int64_t)((s - 1) & 1) * ngroups * G * NGRAN;
      constexpr int PER = (G - 1) * NGRAN / 256;     // granules per thread
      constexpr int CH = PER > 6 ? 5 : PER;          // per polling round (bounds the registers held)
      static_assert(PER % CH == 0, "sweep chunking");
      for (int c0 = 0; c0 < PER; c0 += CH) {
        u64 v[CH];
        unsigned spins = 0;
        bool ok;
        do {
          ok = true;
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const int q = tid + (c0 + i) * 256;
            const int pi = q / NGRAN, gi = q % NGRAN;
            const int peer = pi + (pi >= member ? 1 : 0);
            v[i] = granule_load(src + (int64_t)peer * NGRAN + gi);
            ok = ok && ((unsigned)(v[i] >> 32) == (unsigned)s);
          }
          if (!ok) {
            if (++spins > SPIN_LIMIT) { fail_flag = 1; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        } while (!ok);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int q = tid + (c0 + i) * 256;
          const int pi = q / NGRAN, gi = q % NGRAN;
          const int peer = pi + (pi >= member ? 1 : 0);
          const int rp = gi / HS, ul = gi % HS;
          const unsigned val = (unsigned)v[i];
          hlds[cur][rp * 2][peer * HS + ul] = (unsigned short)(val & 0xffffu);
          hlds[cur][rp * 2 + 1][peer * HS + ul] = (unsigned short)(val >> 16);
        }
      }
    }
    __syncthreads();
    if (fail_flag) break;

    f32x4 acc[4][UB];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ub = 0; ub < UB; ++ub) acc[g][ub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KCW; ++kc) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(&hlds[cur][l15][(kh * KCW + kc) * 32 + 8 * lq]);
#pragma unroll
      for (int ub = 0; ub < UB; ++ub)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g][ub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[ub][kc][g], acc[g][ub], 0, 0, 0);
    }
    if constexpr (KS > 1) {
      // the upper-K wave hands its partial sums to the lead wave of the same unit block through LDS
      f32x4* rbuf = reinterpret_cast<f32x4*>(red);
      if (!lead) {
#pragma unroll
        for (int g = 0; g < 4; ++g) rbuf[((wave & 1) * 4 + g) * 64 + lane] = acc[g][0];
      }
      __syncthreads();
      if (lead) {
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g][0] += rbuf[((wave & 1) * 4 + g) * 64 + lane];
      }
    }
    if (lead) {

    u64* dst = ex_group + (int64_t)(s & 1) * ngroups * G * NGRAN + (int64_t)member * NGRAN;
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      const int unit = unit0 + ub * 16;
      const int ul = unit - member * HS;
      unsigned short hb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gi = las_sigmoid(acc[0][ub][r] + xp[0][ub][r]);
        const float gj = las_tanh(acc[1][ub][r] + xp[1][ub][r]);
        const float gf = las_sigmoid(acc[2][ub][r] + xp[2][ub][r] + 1.0f);
        const float go = las_sigmoid(acc[3][ub][r] + xp[3][ub][r]);
        const float cn = gf * c[ub][r] + gi * gj;
        const unsigned short hn = las_f2bf(go * las_tanh(cn));
        if (act[r]) {
          *reinterpret_cast<float4*>(xproj + rowoff[r] * xrow + dir * 4 * H + unit * 4) = make_float4(gi, gj, gf, go);
          cbuf[rowoff[r] * yrow + dir * H + unit] = cn;
          y[rowoff[r] * yrow + dir * H + unit] = hn;
          c[ub][r] = cn;
          h[ub][r] = las_bf2f(hn);
        }
        hb[r] = act[r] ? hn : las_f2bf(h[ub][r]);
        hlds[cur ^ 1][lq * 4 + r][unit] = hb[r];
      }
      if constexpr (G > 1) {
        granule_store(dst + (lq * 2) * HS + ul, (unsigned)(s + 1), (unsigned)hb[0] | ((unsigned)hb[1] << 16), local);
        granule_store(dst + (lq * 2 + 1) * HS + ul, (unsigned)(s + 1), (unsigned)hb[2] | ((unsigned)hb[3] << 16), local);
      }
    }
    }   // lead
    cur ^= 1;
  }

  if (fail_flag) {
    if (tid == 0) atomicOr(status, 1u);
    return;
  }
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (bidx[r] < B && lead) {
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
                                                          u64* __restrict__ exch, unsigned* __restrict__ status,
                                                          int B, int T, int ndir, int ngroups) {
  constexpr int G = coop_members(H);
  constexpr int HS = H / G;
  constexpr int KS = k_split(H);
  constexpr int NUB = HS / 16;
  constexpr int UB = (NUB * KS) / 4 > 0 ? (NUB * KS) / 4 : 1;
  constexpr int KC = (4 * H) / 32;
  constexpr int KCW = KC / KS;
  constexpr int ZS = 4 * H + 8;
  constexpr int NGRAN = 8 * 4 * HS;    // granules per member per step: 8 row pairs x 4 gates x HS units
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* dzl = reinterpret_cast<unsigned short*>(smem);   // [2][16][ZS]
  int& fail_flag = *reinterpret_cast<int*>(smem + (size_t)2 * 16 * ZS * sizeof(unsigned short));   // keeps the dynamic base 16-B aligned
  f32x4* rbuf = reinterpret_cast<f32x4*>(smem + (size_t)2 * 16 * ZS * sizeof(unsigned short) + 16);   // [2][64] (KS > 1)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gstride = (ngroups + 7) & ~7;
  const int group = blockIdx.x % gstride, member = blockIdx.x / gstride;
  if (group >= ngroups) return;
  const int nslices = ngroups / ndir;
  const int slice = group % nslices, dir = group / nslices;
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

  const int wblk = KS > 1 ? (wave & 1) : wave * UB;
  const int khalf = KS > 1 ? (wave >> 1) & 1 : 0;
  const bool lead = (khalf == 0);
  const int unit0 = member * HS + wblk * 16 + l15;
  const unsigned short* khd = kh + (int64_t)dir * H * 4 * H;
  // register-resident B fragments: B[k][n] = K_h[n = unit][k = gate column]  (natural rows of K_h)
  bf16x8 wf[UB][KCW];
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int kc = 0; kc < KCW; ++kc)
      wf[ub][kc] = *reinterpret_cast<const bf16x8*>(khd + (int64_t)(unit0 + ub * 16) * 4 * H + (khalf * KCW + kc) * 32 + 8 * lq);

  float dc[UB][4], dh[UB][4];
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = bidx[r] < B && lead;
      const int64_t o = ((int64_t)dir * B + (ok ? bidx[r] : 0)) * H + unit0 + ub * 16;
      dc[ub][r] = (ok && dc_last) ? dc_last[o] : 0.f;
      dh[ub][r] = (ok && dh_last) ? dh_last[o] : 0.f;
    }
  if (tid == 0) fail_flag = 0;
  __syncthreads();
  int* colo_flag = reinterpret_cast<int*>(smem + (size_t)2 * 16 * ZS * sizeof(unsigned short)) + 1;
  const bool local = xcd_colocated<G>(exch + (int64_t)2 * ngroups * G * NGRAN + (int64_t)group * G, member, colo_flag, status);

  const int64_t grow = (int64_t)ndir * 4 * H;
  const int64_t yrow = (int64_t)ndir * H;
  u64* ex_group = exch + (int64_t)group * G * NGRAN;

  // saved forward values of one step, loaded one step ahead so their latency hides behind the all-gather
  struct Saved { float gi, gj, gf, go, ct, cp, dyv; };
  auto load_saved = [&](int s, Saved (&sv)[UB][4]) {
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      const int unit = unit0 + ub * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Saved v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (s >= 0 && s < len[r] && lead) {
          const int pos = dir == 0 ? s : len[r] - 1 - s;
          const int64_t ro = (int64_t)bidx[r] * T + pos;
          const float4 gv = *reinterpret_cast<const float4*>(gates + ro * grow + dir * 4 * H + unit * 4);
          v.gi = gv.x; v.gj = gv.y; v.gf = gv.z; v.go = gv.w;
          v.ct = cbuf[ro * yrow + dir * H + unit];
          if (s > 0) {
            const int64_t rp = (int64_t)bidx[r] * T + (dir == 0 ? pos - 1 : pos + 1);
            v.cp = cbuf[rp * yrow + dir * H + unit];
          }
          v.dyv = dy[ro * yrow + dir * H + unit];
        }
        sv[ub][r] = v;
      }
    }
  };

  int cur = 0;
  unsigned epoch = 0;
  Saved sv[UB][4];
  load_saved(smax - 1, sv);
  for (int s = smax - 1; s >= 0; --s) {
    ++epoch;
    unsigned short* zl = dzl + cur * 16 * ZS;
    u64* dst = ex_group + (int64_t)(epoch & 1) * ngroups * G * NGRAN + (int64_t)member * NGRAN;
    if (lead)
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      const int unit = unit0 + ub * 16;
      const int ul = unit - member * HS;
      unsigned short zb[4][4];       // [gate][row]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool act = s < len[r];
        float di = 0.f, dj = 0.f, df = 0.f, dov = 0.f;
        if (act) {
          const Saved v = sv[ub][r];
          const int pos = dir == 0 ? s : len[r] - 1 - s;
          const int64_t ro = (int64_t)bidx[r] * T + pos;
          const float dht = v.dyv + dh[ub][r];
          const float tc = las_tanh(v.ct);
          dov = dht * tc * v.go * (1.f - v.go);
          const float dct = dc[ub][r] + dht * v.go * (1.f - tc * tc);
          di = dct * v.gj * v.gi * (1.f - v.gi);
          dj = dct * v.gi * (1.f - v.gj * v.gj);
          df = dct * v.cp * v.gf * (1.f - v.gf);
          dc[ub][r] = dct * v.gf;
          uint2 zv;
          zv.x = (unsigned)las_f2bf(di) | ((unsigned)las_f2bf(dj) << 16);
          zv.y = (unsigned)las_f2bf(df) | ((unsigned)las_f2bf(dov) << 16);
          *reinterpret_cast<uint2*>(dz + ro * grow + dir * 4 * H + unit * 4) = zv;   // gate-interleaved [unit][i,j,f,o]
        }
        zb[0][r] = las_f2bf(di); zb[1][r] = las_f2bf(dj); zb[2][r] = las_f2bf(df); zb[3][r] = las_f2bf(dov);
        unsigned short* zr = zl + (lq * 4 + r) * ZS + unit;
        zr[0] = zb[0][r]; zr[H] = zb[1][r]; zr[2 * H] = zb[2][r]; zr[3 * H] = zb[3][r];
      }
      if constexpr (G > 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          granule_store(dst + ((lq * 2) * 4 + g) * HS + ul, epoch, (unsigned)zb[g][0] | ((unsigned)zb[g][1] << 16), local);
          granule_store(dst + ((lq * 2 + 1) * 4 + g) * HS + ul, epoch, (unsigned)zb[g][2] | ((unsigned)zb[g][3] << 16), local);
        }
      }
    }
    load_saved(s - 1, sv);      // next step's operands: in flight during the all-gather and the MFMAs

    // all-gather dz_t from the peers
    if constexpr (G > 1) {
      const u64* src = ex_group + (int64_t)(epoch & 1) * ngroups * G * NGRAN;
      constexpr int TOTAL = (G - 1) * NGRAN;
      constexpr int CH = 12;                         // granules per thread per chunk
      for (int base = 0; base < TOTAL; base += 256 * CH) {
        u64 v[CH];
        unsigned spins = 0;
        bool ok;
        do {
          ok = true;
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const int q = base + tid + i * 256;
            const int pi = q / NGRAN, gi = q % NGRAN;
            const int peer = pi + (pi >= member ? 1 : 0);
            v[i] = granule_load(src + (int64_t)peer * NGRAN + gi);
            ok = ok && ((unsigned)(v[i] >> 32) == epoch);
          }
          if (!ok) {
            if (++spins > SPIN_LIMIT) { fail_flag = 1; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        } while (!ok);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int q = base + tid + i * 256;
          const int pi = q / NGRAN, gi = q % NGRAN;
          const int peer = pi + (pi >= member ? 1 : 0);
          const int rp = gi / (4 * HS), g = (gi / HS) % 4, ul = gi % HS;
          const unsigned val = (unsigned)v[i];
          zl[(rp * 2) * ZS + g * H + peer * HS + ul] = (unsigned short)(val & 0xffffu);
          zl[(rp * 2 + 1) * ZS + g * H + peer * HS + ul] = (unsigned short)(val >> 16);
        }
      }
    }
    __syncthreads();
    if (fail_flag) break;

    // dh_{t-1}[own units] = dz_t [16,4H] * K_h^T
    f32x4 acc[UB];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) acc[ub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KCW; ++kc) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(zl + l15 * ZS + (khalf * KCW + kc) * 32 + 8 * lq);
#pragma unroll
      for (int ub = 0; ub < UB; ++ub) acc[ub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[ub][kc], acc[ub], 0, 0, 0);
    }
    if constexpr (KS > 1) {
      if (!lead) rbuf[(wave & 1) * 64 + lane] = acc[0];
      __syncthreads();
      if (lead) acc[0] += rbuf[(wave & 1) * 64 + lane];
    }
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (s < len[r] && lead) dh[ub][r] = acc[ub][r];
    cur ^= 1;
  }
  if (fail_flag && tid == 0) atomicOr(status, 2u);
}

struct CoopGeom { int nslices, ngroups, G, blocks; size_t exch_bytes; };

CoopGeom geom(int B, int H, int ndir, bool bwd) {
  CoopGeom g;
  g.G = coop_members(H);
  g.nslices = (B + 15) / 16;
  g.ngroups = g.nslices * ndir;
  g.blocks = ((g.ngroups + 7) & ~7) * g.G;        // group stride rounded up to 8 (idle blocks exit at once)
  const size_t ngran = (size_t)8 * (H / g.G) * (bwd ? 4 : 1);
  g.exch_bytes = g.G > 1 ? ((size_t)2 * g.ngroups * g.G * ngran + (size_t)g.ngroups * g.G) * sizeof(u64) : 0;   // + XCC-id table
  return g;
}

template <int H>
int launch_fwd(float* xproj, const las_bf16* wp, const int32_t* length, las_bf16* y, float* cbuf, float* c_last,
               float* h_last, void* ws, int B, int T, int ndir, hipStream_t st) {
  const CoopGeom g = geom(B, H, ndir, false);
  unsigned* status = reinterpret_cast<unsigned*>(ws);
  u64* exch = reinterpret_cast<u64*>(reinterpret_cast<char*>(ws) + 64);
  hipLaunchKernelGGL((lstm_fwd_kernel<H>), dim3(g.blocks), dim3(rec_threads(H)), 0, st, xproj, wp, length, y, cbuf, c_last, h_last,
                     exch, status, B, T, ndir, g.ngroups);
  LAS_LAUNCH_CHECK("lstm fwd launch");
  return LAS_OK;
}

template <int H>
int launch_bwd(const float* gates, const float* cbuf, const float* dy, const float* dc_last, const float* dh_last,
               const las_bf16* kh, const int32_t* length, las_bf16* dz, void* ws, int B, int T, int ndir, hipStream_t st) {
  const CoopGeom g = geom(B, H, ndir, true);
  unsigned* status = reinterpret_cast<unsigned*>(ws);
  u64* exch = reinterpret_cast<u64*>(reinterpret_cast<char*>(ws) + 64);
  const size_t lds = (size_t)2 * 16 * (4 * H + 8) * sizeof(unsigned short) + 16 + 2 * 64 * 16;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((lstm_bwd_kernel<H>), dim3(g.blocks), dim3(256), lds, st, gates, cbuf, dy, dc_last, dh_last, kh, length,
                     dz, exch, status, B, T, ndir, g.ngroups);
  LAS_LAUNCH_CHECK("lstm bwd launch");
  return LAS_OK;
}

bool supported_units(int H) { return H == 64 || H == 128 || H == 256 || H == 512; }

}  // namespace

extern "C" size_t las_lstm_workspace_bytes(int B, int H, int ndir) {
  if (!supported_units(H) || B <= 0) return 0;
  return 64 + geom(B, H, ndir, true).exch_bytes;
}

extern "C" int las_lstm_pack_recurrent(const float* kernel_h, int H, las_bf16* packed, void* stream) {
  LAS_REQUIRE(supported_units(H), "las_lstm_pack_recurrent: num_units must be 64, 128, 256 or 512 (got %d)", H);
  hipLaunchKernelGGL(pack_recurrent_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, kernel_h, H, packed);
  LAS_LAUNCH_CHECK("pack launch");
  return LAS_OK;
}

extern "C" int las_lstm_recurrent_fwd(float* xproj, const las_bf16* wpacked, const int32_t* length, las_bf16* y,
                                      float* cbuf, float* c_last, float* h_last, void* workspace, int B, int T, int H,
                                      int ndir, void* stream) {
  LAS_REQUIRE(B > 0 && T > 0 && (ndir == 1 || ndir == 2), "las_lstm_recurrent_fwd: bad shape B=%d T=%d ndir=%d", B, T, ndir);
  LAS_REQUIRE(supported_units(H), "las_lstm_recurrent_fwd: num_units %d not in {64,128,256,512}", H);
  LAS_REQUIRE(workspace != nullptr && ((uintptr_t)workspace % 16 == 0), "las_lstm_recurrent_fwd: workspace missing or misaligned");
  hipStream_t st = (hipStream_t)stream;
  int rc = las_check_hip(hipMemsetAsync(y, 0, (size_t)B * T * ndir * H * sizeof(las_bf16), st), "memset y");
  if (rc) return rc;
  rc = las_check_hip(hipMemsetAsync(workspace, 0, 64 + geom(B, H, ndir, false).exch_bytes, st), "memset workspace");
  if (rc) return rc;
  switch (H) {
    case 64: return launch_fwd<64>(xproj, wpacked, length, y, cbuf, c_last, h_last, workspace, B, T, ndir, st);
    case 128: return launch_fwd<128>(xproj, wpacked, length, y, cbuf, c_last, h_last, workspace, B, T, ndir, st);
    case 512: return launch_fwd<512>(xproj, wpacked, length, y, cbuf, c_last, h_last, workspace, B, T, ndir, st);
    default: return launch_fwd<256>(xproj, wpacked, length, y, cbuf, c_last, h_last, workspace, B, T, ndir, st);
  }
}

extern "C" int las_lstm_recurrent_bwd(const float* gates, const float* cbuf, const float* dy, const float* dc_last,
                                      const float* dh_last, const las_bf16* kh_bf16, const int32_t* length,
                                      las_bf16* dz, void* workspace, int B, int T, int H, int ndir, void* stream) {
  LAS_REQUIRE(B > 0 && T > 0 && (ndir == 1 || ndir == 2), "las_lstm_recurrent_bwd: bad shape");
  LAS_REQUIRE(supported_units(H), "las_lstm_recurrent_bwd: num_units %d not in {64,128,256,512}", H);
  LAS_REQUIRE(workspace != nullptr && ((uintptr_t)workspace % 16 == 0), "las_lstm_recurrent_bwd: workspace missing or misaligned");
  hipStream_t st = (hipStream_t)stream;
  int rc = las_check_hip(hipMemsetAsync(dz, 0, (size_t)B * T * ndir * 4 * H * sizeof(las_bf16), st), "memset dz");
  if (rc) return rc;
  rc = las_check_hip(hipMemsetAsync(workspace, 0, 64 + geom(B, H, ndir, true).exch_bytes, st), "memset workspace");
  if (rc) return rc;
  switch (H) {
    case 64: return launch_bwd<64>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, workspace, B, T, ndir, st);
    case 128: return launch_bwd<128>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, workspace, B, T, ndir, st);
    case 512: return launch_bwd<512>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, workspace, B, T, ndir, st);
    default: return launch_bwd<256>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, workspace, B, T, ndir, st);
  }
}
