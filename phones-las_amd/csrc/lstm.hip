// Listener recurrence: tf.nn.(bidirectional_)dynamic_rnn over LSTMCell (las/ops.py:10-46) as two
// persistent, weight-stationary kernels (forward, backward-in-time) on gfx950.
//
// Decomposition.  The batch is cut into slices of 16 utterances (the M of v_mfma_f32_16x16x32_bf16) -- or of 8 / 4
// (rows 0..1 / row 0 of every lane's quad; the rest of the tile stays zero) while every chain still finds CUs of its
// own: a step costs what one lane has to do in it, and shorter slices give every lane fewer (utterance, unit) pairs.
// One (slice, direction) pair is a serial chain of T dependent steps; it is run by a GROUP of G
// co-resident 256-thread workgroups ("members", one wave per SIMD).  Member m owns hidden units
// [m*H/G, (m+1)*H/G) and keeps its part of K_h in REGISTERS for the whole sequence as ready-made MFMA B
// fragments - the recurrent weights are read from memory once per launch, not once per step.  i, j, f, o of
// one (utterance, unit) land in the same lane (C/D layout col = lane&15 -> unit, row = (lane>>4)*4+reg ->
// utterance), so the gate math needs no cross-lane traffic.
//
// Exchange.  Forward: the members all-gather h_t (16 x H bf16).  Backward: each member multiplies ITS dz
// columns with its rows of K_h^T and the partial dh tiles are reduce-scattered to their owners.  Inside a
// member data goes through a double-buffered LDS tile (one LDS-only barrier per step); between members
// through 8-byte {tag, payload} granules in a global exchange buffer: the data is its own flag
// (cdna_hip_programming.md Guideline 16, form R2), tag = LAUNCH BASE + step epoch, two parity slots so a fast member
// cannot overwrite what a slow one still reads.  The launch base lives in the workspace header and is advanced past every
// tag of the launch by the LAST workgroup to leave (launch_arrive below): whatever earlier launches left in the buffer
// carries smaller tags and can never satisfy a poll, so the buffer is zeroed once, by the caller, when it is allocated --
// not by a memset node in front of every launch (round 4: six nodes and their launch gaps per train step).  Loads are
// agent-scope relaxed atomics (L1-bypassing); stores are write-through agent-scope atomics, or plain stores
// once the members have established that they share an XCD (then its L2 is the coherence point).
// Results never depend on workgroup placement; blocks in chunks of 8 groups (block = chunk*8G + member*8 + group%8) only make the members of a
// group share an XCD under round-robin dispatch (speed).  Every spin is bounded: on timeout the kernel sets
// a status word and returns.  Each group has a PREFETCH COMPANION workgroup per four members on another CU of the
// same XCD that pulls their HBM operands into the shared L2 a few steps ahead.
#include "las_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

constexpr unsigned SPIN_LIMIT = 1u << 20;

// Padding (bf16 elements) of a row of the LDS tiles the MFMA A fragments are read from (h_t forward, dz_t backward).  A lane
// (l15 = row, lq = k group) reads 16 bytes at row * stride + 64 kc + 16 lq; `ds_read_b128` serves a wave in four groups of 16
// lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS table), i.e. rows {0-3, 12-15} of one k group
// with rows {4-11} of the NEXT -- and a group is conflict-free when its sixteen 16-byte slots (address / 16 mod 16) differ.
// With 8 elements of padding the stride is 1 slot (mod 16) and row 12 / k group q meets row 11 / k group q + 1 in every group: two
// LDS cycles per group instead of one.  16 elements make the stride 2 slots: rows {0-3, 12-15} take the even slots, rows {4-11} + 1
// the odd ones.  Round 6 (profiles/r06_lds_pad_ab.txt, the library built with -DLAS_LSTM_LDS_PAD=8 against this one, bit-identical
// results): forward 0.86 -> 0.835 us per step at 256 units, 1.57 -> 1.49 at 512; backward 0.88 -> 0.865 at 256 (it reads half as
// many fragments per step); metric-M 5.78 -> 5.73 ms, metric-L 16.69 -> 16.58.  The single-workgroup chains (64 / 128 units) measured
// no better with it (128 units backward 0.955 -> 0.965): they keep 8.
#ifdef LAS_LSTM_LDS_PAD
constexpr int lds_pad(int) { return LAS_LSTM_LDS_PAD; }
#else
constexpr int lds_pad(int G) { return G > 1 ? 16 : 8; }
#endif

#ifdef LAS_STAMPS
// diagnostics build (LAS_CXXFLAGS=-DLAS_STAMPS): wall-clock (100 MHz) stamps of the phases of the first 256 steps of workgroup 0
// (forward launch: [0, 2048), backward launch: [2048, 4096)); scripts/gpu_lstm_stamps.py
__device__ unsigned long long las_lstm_stamps[2 * 256 * 8];
#define LSTM_STAMP(off, step, k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (step) < 256) las_lstm_stamps[(off) + (step) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define LSTM_STAMP(off, step, k) do { } while (0)
#endif

// Workgroup barrier that only orders LDS traffic.  __syncthreads() also drains vmcnt, i.e. waits until every global
// store of the step has been acknowledged, which puts that round trip on the critical path of every step.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) only (vmcnt, expcnt untouched)
  __builtin_amdgcn_s_barrier();
}

// Re-read a flag word in LDS as a ds_read_b32 (a cast to a generic `volatile int*` is compiled into a FLAT load followed by
// `s_waitcnt vmcnt(0)`; scripts/check_isa.py lists the kernels that have flat instructions).
__device__ __forceinline__ int lds_flag(int* p) {
  return *reinterpret_cast<volatile __attribute__((address_space(3))) int*>((__attribute__((address_space(3))) int*)p);
}

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

// Launch epochs (round 4).  Workspace header (the 64 bytes in front of the exchange buffer), as 32-bit words: [0] sticky
// timeout status, [1], [2] diagnostics counters, [4] EPOCH BASE of the next launch, [5] arrival counter.  Every tag a
// launch writes is base + k with 1 <= k <= T + 1; every workgroup of the grid -- members, companions, the idle blocks of a
// rounded-up grid -- calls launch_arrive() on its way out, and the last one moves the base past this launch's tags.  All of
// them have read the base by then (they read it first and only arrive when they are done), so the members of a group
// always agree on it.  Before the 32-bit tags could wrap, the last workgroup clears the exchange buffer and starts the base
// over from zero (once in about two million launches).
constexpr int HDR_BASE = 4, HDR_ARRIVE = 5;
__device__ __forceinline__ unsigned launch_base(const unsigned* status) {
  return __hip_atomic_load(status + HDR_BASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void launch_arrive(unsigned* status, unsigned base, int T, u64* exch, long long exch_words) {
  __shared__ int last_one;
  __syncthreads();                                   // every thread of this workgroup is done with the exchange
  if (threadIdx.x == 0) {
    const unsigned n = atomicAdd(status + HDR_ARRIVE, 1u);
    last_one = (n == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!last_one) return;
  const unsigned next = base + (unsigned)T + 2u;
  const bool wrap = next >= 0x7ff00000u;
  if (wrap)
    for (long long i = threadIdx.x; i < exch_words; i += blockDim.x) exch[i] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(status + HDR_ARRIVE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(status + HDR_BASE, wrap ? 0u : next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Do all G members of this group sit on one XCD?  Each member publishes its XCC id with the placement-independent
// write-through store, then reads everybody's.  Purely a speed decision: the result only selects the store flavour.
// Call from all threads; returns the same value in every thread of the workgroup (fail -> false).
template <int G>
__device__ bool xcd_colocated(u64* table, int member, int* lds_flag, unsigned* status, unsigned base) {
  if (G == 1) return false;
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    __hip_atomic_store(table + member, ((u64)(base + 1u) << 32) | (xcc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool same = true;
    for (int m = 0; m < G; ++m) {
      u64 v = 0;
      unsigned spins = 0;
      do {
        v = __hip_atomic_load(table + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == base + 1u) break;
        __builtin_amdgcn_s_sleep(2);
      } while (++spins < (1u << 20));
      same = same && ((unsigned)(v >> 32) == base + 1u) && ((unsigned)v == xcc + 1);
    }
    *lds_flag = same ? 1 : 0;
    atomicAdd(status + (same ? 1 : 2), 1u);      // diagnostics: members that took the local / the fabric flavour
  }
  __syncthreads();
  return *lds_flag != 0;
}

// number of cooperating workgroups per chain
// H <= 128: one workgroup; H = 256: 4 x 64 units; H = 512: 8 x 64 units (round 3: one 16-unit block per wave over the WHOLE K, 256
// weight registers of the wave's 512, pinned in AccVGPRs -- the H = 256 code path at twice the K; the round-2 form, 16 x 32 units
// with the K / row split, measured 2.46 against 2.06 us per backward step and was removed in round 6); H = 1024 (round 5): 32 x 32
// units with the K dimension split over wave pairs (forward) and the rows of a quad split over wave pairs (backward) -- a chain is
// then a whole XCD (32 CUs), B = 64 fills the chip with its 8 chains: built for completeness (the reference takes any num_units,
// las/ops.py:10-12), not for speed
__host__ __device__ constexpr int coop_members(int H) { return H <= 128 ? 1 : (H == 256 ? 4 : (H == 512 ? 8 : H / 32)); }
// The kernels take the member count as a template parameter (default: coop_members(H)).
__host__ __device__ constexpr int k_split_g(int H, int G) { return H / G < 64 ? 2 : 1; }
// prefetch companions per group: one serves four members' columns
__host__ __device__ constexpr int group_companions_g(int G) { return G >= 4 ? G / 4 : 1; }

// ------------------------------------------------------------------------------------------------
// pack K_h [H,4H] fp32 -> MFMA-B-fragment-major bf16, grouped by 16-unit block:
//   packed[(((ublk*KC + kc)*4 + g)*64 + lane)*8 + j] = K_h[kc*32 + 8*(lane>>4) + j][g*H + ublk*16 + (lane&15)]
// ------------------------------------------------------------------------------------------------
__global__ void pack_recurrent_kernel(const float* kh, int H, unsigned short* packed) {
  const int64_t total = (int64_t)H * 4 * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    packed[i] = las_f2bf(kh[las_pack_recurrent_src(i, H)]);
}

// K_x = kernel[0:D, :] (fp32, ld 4H) -> the same fragment-major image over `chunks` 32-deep K chunks, zero from row D on:
//   packed[(((ublk*chunks + kc)*4 + g)*64 + lane)*8 + j] = K_x[kc*32 + 8*(lane>>4) + j][g*H + ublk*16 + (lane&15)]
__global__ void pack_input_kernel(const float* kx, int D, int H, int chunks, unsigned short* packed) {
  const int64_t total = (int64_t)(H / 16) * chunks * 4 * 512;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t src = las_pack_input_src(i, D, H, chunks);
    packed[i] = las_f2bf(src >= 0 ? kx[src] : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Workgroup size: 4 compute waves, one per SIMD (512-register budget).  When the chain is shared by several workgroups
// the group has PREFETCH COMPANION workgroups (see the kernel) that keep its HBM operands L2-resident.

// ROWS = utterances per slice: 16 fills the MFMA tile; 8 (rows 0,1 of every quad; the other two stay zero) halves the
// element-wise work, the exchange and the HBM accesses of every lane at the same MFMA cost -- the per-step latency is
// mostly that work, so small batches run twice as many (half-filled) chains on otherwise idle CUs.
// KX > 0 (round 4): the INPUT projection x_t K_x + b is formed inside the step as well -- KX 32-deep chunks of the input width,
// K_x register-resident as B fragments (las_lstm_pack_input), x_t as A fragments straight from the [B, T, Dp] bf16 features
// (loaded one step ahead; the companion keeps the rows L2-resident).  These products do not depend on h_{t-1}: they are issued
// at the top of the step and run while the peers' granules are on their way, so the bottom listener layer needs no x K_x GEMM,
// no 419 MB fp32 round trip of its result, and the chain no xproj load.  `xproj` is then only the saved-gates OUTPUT.
struct FusedInput {
  const unsigned short* x;       // [B, T, ldx] bf16 (per direction: + dir * xdir elements)
  int64_t ldx, xdir;
  const unsigned short* kxp;     // ndir images of las_lstm_pack_input
  const float* bias;             // [ndir * 4H] gate-interleaved
  int Dp;                        // valid input columns (multiple of 8, <= 32 * KX)
  // streamed input projection (round 4): xproj is being written by a las_gemm_nt_stream launch that runs BESIDE this kernel.
  // `ready` is the buffer shared with it (layout: gemm.hip, las_stream_flags_offset): this kernel writes, per chain group, the
  // XCD the group runs on (word 16 + group), the product's workgroups on that XCD then produce the group's rows into the L2
  // both sides share and count finished column tiles in word flags + group * nsb + step block; `want` of them make a block
  // of 256 / ROWS steps complete.  nullptr: xproj is complete on entry.
  unsigned* ready;
  int nsb, flags;
  unsigned want;
  int rows;                      // utterances per slice of this launch (0: slice_rows' choice)
};

// Wait until the streamed rows of step block sb of this group are there (wave-uniform; bounded).  Returns false on timeout.
__device__ __forceinline__ bool stream_wait(const FusedInput& fi, int group, int sb) {
  const unsigned* f = fi.ready + fi.flags + (int64_t)group * fi.nsb + sb;
  unsigned spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < fi.want) {
    if (++spins > SPIN_LIMIT) return false;
    __builtin_amdgcn_s_sleep(4);
  }
  asm volatile("" ::: "memory");        // the rows are read after the count (no acquire fence: nobody has touched these lines before)
  return true;
}

template <int H, int ROWS, int G, int KX>
__device__ __forceinline__ void lstm_fwd_body(float* __restrict__ xproj, const unsigned short* __restrict__ wpacked,
                                              const int32_t* __restrict__ length, unsigned short* __restrict__ y,
                                              float* __restrict__ cbuf, float* __restrict__ c_last,
                                              float* __restrict__ h_last, u64* __restrict__ exch,
                                              unsigned* __restrict__ status, int B, int T, int ndir, int ngroups,
                                              int companions, const FusedInput fi, const unsigned base) {
  constexpr int HS = H / G;            // units per member
  constexpr int KS = k_split_g(H, G);  // ways the K dimension is split over waves
  constexpr int NUB = HS / 16;         // 16-unit blocks of this member
  constexpr int UB = (NUB * KS) / 4 > 0 ? (NUB * KS) / 4 : 1;   // blocks per wave (4 compute waves)
  constexpr int KC = H / 32;
  constexpr int KCW = KC / KS;         // k-chunks a wave owns
  constexpr int LS = H + lds_pad(G);   // LDS row stride (elements)
  constexpr int RL = ROWS / 4;         // utterances per lane (rows lq*4 .. lq*4+RL-1 of the MFMA tile)
  static_assert(ROWS == 16 || ROWS == 8 || (ROWS == 4 && KS == 1), "4-row slices: only without the K split");
  constexpr int GV = RL >= 2 ? 2 : 1;      // bf16 values per granule: a row pair of one unit, or a single value (4-row slices)
  constexpr int NGRAN = ROWS * HS / GV;    // granules a member publishes per step
  constexpr int NPOLL = (G - 1) * NGRAN;   // granules a member collects per step
  // WIDE polls (round 5): a thread fetches two ADJACENT granules of a peer with one 16-byte load and checks both epochs -- each
  // 8-byte half is a granule of its own, written by one 8-byte store, so nothing is assumed about the 16 bytes arriving together.
  // The load is `global_load_dwordx4 ... sc1` in inline assembly, i.e. granule_load's instruction at twice the width (there is no
  // 16-byte atomic load to ask the compiler for).  NOT a raw buffer load: `buffer_load_dwordx4 ... sc0 sc1` through a descriptor
  // kept returning the previous step's granules from this CU's L1 while the 8-byte load of the same address saw the new ones
  // (scripts/micro/wideload.hip passes -- a lone poller -- the chains timed out: measured, round 5).
  // Half the polling instructions (256 units, 4 rows: 3 -> 2 per thread; 512 units, 8 rows: 7 -> 4); -DLAS_FWD_NARROW_POLL
  // builds the one-granule form (A/B).
#ifdef LAS_FWD_NARROW_POLL
  constexpr bool WIDEP = false;
#else
  constexpr bool WIDEP = G > 1 && NGRAN % 2 == 0 && HS % 2 == 0 && KX == 0;   // (KX > 0 issues x_{s+1} loads between a round and its check)
#endif
  constexpr int NPW = WIDEP ? NPOLL / 2 : NPOLL;           // polling units of a member per step (pairs / granules)
  constexpr int PER = G > 1 ? (NPW + 255) / 256 : 1;       // ... per thread (the threads past the end poll a unit a second time)
  __shared__ __attribute__((aligned(16))) unsigned short hlds[2][16][LS];
  __shared__ int fail_flag;
  __shared__ int colo_flag;
  __shared__ __attribute__((aligned(16))) float pf_scratch[256];   // 1 KiB sink of the companion's LDS-DMAs
  __shared__ __attribute__((aligned(16))) float red[KS > 1 ? 2 * 4 * 64 * 4 : 4];   // partial sums of the upper K half

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gstride = (ngroups + 7) & ~7;        // members of a group are 8k blocks apart: one XCD under round-robin
  const int nblk = gstride * G;                  // compute workgroups; the blocks beyond them: CPG prefetch companions per group
  constexpr int CPG = group_companions_g(G);
  const bool companion = (int)blockIdx.x >= nblk;
  const int cblk = companion ? blockIdx.x - nblk : blockIdx.x;
  // Blocks are laid out in chunks of 8 groups: block = chunk * 8 * G + member * 8 + group % 8.  The members of a group are
  // 8 blocks apart (one XCD under round-robin dispatch), and the dispatcher, which hands out blocks in order, completes
  // the groups of one chunk before it starts the next: when there are more workgroups than the device can hold at once,
  // the resident groups are whole and the others wait their turn (a member-major order would leave every group partial).
  const int per = 8 * (companion ? CPG : G), chunk = cblk / per, within = cblk % per;
  const int group = chunk * 8 + (within & 7), member = companion ? 0 : within >> 3;
  const int cm = companion ? within >> 3 : 0;          // which quarter (1/CPG) of the direction's columns a companion serves
  if (group >= ngroups) return;
  const int nslices = ngroups / ndir;
  const int slice = group % nslices, dir = group / nslices;
  const int l15 = lane & 15, lq = lane >> 4;
  const int64_t xrow = (int64_t)ndir * 4 * H;
  const int64_t yrow = (int64_t)ndir * H;
  const int64_t par_stride = (int64_t)ngroups * G * NGRAN;     // granules per parity slot
  u64* ex_group = exch + (int64_t)group * G * NGRAN;           // + parity*par_stride + member*NGRAN
  u64* done_word = exch + 2 * par_stride + (int64_t)ngroups * G + (int64_t)group * G + member;   // companions watch member 0's

  int len[RL], bidx[RL];
  int smax = 0, smin = 0x7fffffff;
#pragma unroll
  for (int r = 0; r < RL; ++r) {
    bidx[r] = slice * ROWS + lq * RL + r;
    len[r] = (bidx[r] < B) ? min(length[bidx[r]], T) : 0;
    smax = max(smax, len[r]);
    smin = min(smin, len[r]);
  }
  smax = (int)las_wave_max((float)smax);
  smin = -(int)las_wave_max((float)(-smin));       // steps every utterance of the slice is still running: the lean path
  if ((int64_t)B * T * xrow * 4 >= ((int64_t)1 << 32)) smin = 0;     // 32-bit byte offsets do not reach: general path only

  if (companion) {
    // PREFETCH COMPANION: one workgroup per group on another CU of the same XCD (block index = nblk + group, and nblk is
    // a multiple of 8) pulls the xproj lines of ALL members of the group into the shared L2 PF_AHEAD steps before they
    // are needed (1-KiB LDS-DMAs into a scratch tile nobody reads), so the compute CUs' in-order vector-memory queues
    // only ever see L2 hits.  It paces itself on the epoch tags of member 0's granules and leaves when member 0's done
    // word is set.
    constexpr int PF_AHEAD = 6;
    const int mylen = (l15 < ROWS && slice * ROWS + l15 < B) ? min(length[slice * ROWS + l15], T) : 0;   // lane = row of the slice
    // rows t >= length of the outputs are zero (dynamic_rnn): the companion clears the direction's columns of them, so a
    // dense batch needs no memset of y at all (the compute workgroups write every row they own)
    for (int rr = wave * RL; rr < wave * RL + RL; ++rr) {
      const int bb = slice * ROWS + rr;
      if (bb >= B) continue;
      const int ll = __builtin_amdgcn_readlane(mylen, rr);
      constexpr int LPR = H * 2 / 16 / CPG;                       // 16-byte pieces of this companion's share of a row
      for (int e = ll * LPR + lane; e < T * LPR; e += 64) {
        const int t = e / LPR, c = cm * LPR + e % LPR;
        *reinterpret_cast<uint4*>(y + ((int64_t)bb * T + t) * yrow + dir * H + c * 8) = make_uint4(0, 0, 0, 0);
      }
    }
    const u64* tag0 = ex_group;                                   // member 0 publishes here every step
    int seen = -1;
    for (int sp = 0; sp < smax; ++sp) {
      unsigned spins = 0;
      while (seen < sp - PF_AHEAD) {
        const u64 v0 = granule_load(tag0), v1 = granule_load(tag0 + par_stride), dn = granule_load(done_word);
        if ((unsigned)(dn >> 32) == base + 1u) return;            // the compute workgroups are done (or failed)
        seen = max(max((int)((unsigned)(v0 >> 32) - base), (int)((unsigned)(v1 >> 32) - base)), 0) - 1;   // (older launches' tags: <= base)
        if (seen < sp - PF_AHEAD) {
          if (++spins > SPIN_LIMIT) return;
          __builtin_amdgcn_s_sleep(8);
        }
      }
      if constexpr (KX == 0) {
        // streamed xproj: a line must not enter this XCD's L2 before the product has written it (it would stay stale there)
        if (fi.ready != nullptr && sp % (256 / ROWS) == 0 && !stream_wait(fi, group, sp / (256 / ROWS))) return;
      }
#pragma unroll
      for (int r4 = 0; r4 < RL; ++r4) {
        const int rr = wave * RL + r4;
        const int ll = __builtin_amdgcn_readlane(mylen, rr);
        if (sp < ll) {
          const int pos = dir == 0 ? sp : ll - 1 - sp;
          if constexpr (KX > 0) {
            // fused input projection: the step reads the feature row itself (Dp bf16: one dword per lane)
            const unsigned short* src = fi.x + dir * fi.xdir + ((int64_t)(slice * ROWS + rr) * T + pos) * fi.ldx + lane * 2;
            if (cm == 0 && lane * 2 < fi.Dp)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                               (__attribute__((address_space(3))) void*)(pf_scratch), 4, 0, 0);
          } else {
            const float* src = xproj + ((int64_t)(slice * ROWS + rr) * T + pos) * xrow + dir * 4 * H + cm * (4 * H / CPG) + lane * 4;
#pragma unroll
            for (int c4 = 0; c4 < H / 64 / CPG; ++c4)               // this companion's share of the 4H floats in 1-KiB pieces
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c4 * 256),
                                               (__attribute__((address_space(3))) void*)(pf_scratch), 16, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // The chain is latency-bound: when other kernels (weight-gradient GEMMs on the side stream) share this CU, these
  // waves go first in the issue arbitration.
  __builtin_amdgcn_s_setprio(3);

  // register-resident B fragments of this wave's K_h columns
  // wave -> (first unit block, K half): KS = 1: 4 waves x UB blocks, all of K; KS = 2: wave&1 = block, wave>>1 = K half
  const int wblk = KS > 1 ? (wave & 1) : wave * UB;
  const int kh = KS > 1 ? (wave >> 1) & 1 : 0;
  const bool lead = (kh == 0);          // the wave that finishes the step for its units
  bf16x8 wf[UB][KCW][4];
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
  // fused input projection: this wave's K_x columns as B fragments, the bias of this lane's unit(s)
  constexpr int KXA = KX > 0 ? KX : 1;
  bf16x8 wx[UB][KXA][4];
  float4 bz[UB];
  if constexpr (KX > 0) {
    static_assert(KS == 1, "fused input projection: not with the K split");
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      const int ublk = member * NUB + wblk + ub;
#pragma unroll
      for (int kc = 0; kc < KX; ++kc)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          wx[ub][kc][g] = *reinterpret_cast<const bf16x8*>(fi.kxp + (int64_t)dir * (H / 16) * KX * 4 * 512 +
                                                            ((int64_t)((ublk * KX + kc) * 4 + g) * 64 + lane) * 8);
      bz[ub] = *reinterpret_cast<const float4*>(fi.bias + dir * 4 * H + (member * HS + (wblk + ub) * 16 + l15) * 4);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the weights are in registers before the time loop, not waited for inside it
  // More than 128 weight registers (512 units as 8 members: 256) do not fit the 256 architectural VGPRs next to the step's
  // own values.  Left to itself the allocator keeps such weights in AccVGPRs as SPILLS and copies them back through
  // v_accvgpr_read before every product that uses them (four VALU instructions per MFMA: the product phase of a step took
  // twice the MFMAs' own time).  An MFMA reads its B operand from an AccVGPR directly, so the weights are pinned there.
  if constexpr (UB * KCW * 4 * 4 > 128) {
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
      for (int kc = 0; kc < KCW; ++kc)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm("" : "+a"(wf[ub][kc][g]));
  }

  float c[UB][RL], h[UB][RL];
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int r = 0; r < RL; ++r) { c[ub][r] = 0.f; h[ub][r] = 0.f; }

  for (int i = tid; i < 2 * 16 * LS; i += 256) (&hlds[0][0][0])[i] = 0;
  if (tid == 0) fail_flag = 0;
  __syncthreads();
  const bool local = xcd_colocated<G>(exch + 2 * par_stride + (int64_t)group * G, member, &colo_flag, status, base);
  if constexpr (KX == 0) {
    // streamed input product: tell its workgroups where this group runs (they produce its rows into that XCD's L2)
    if (fi.ready != nullptr && member == 0 && tid == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      __hip_atomic_store(fi.ready + 16 + group, (local || G == 1) ? (xcc & 7u) + 1u : 0x100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  const int unit0 = member * HS + wblk * 16 + l15;            // + ub*16

  // loop-invariant pieces of the exchange: which granules this thread polls and where their halves go in the LDS tile
  unsigned poll_off[PER], scat_off[PER];
  if constexpr (G > 1) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int q = (tid + i * 256) % NPW;
      const int pi = WIDEP ? q / (NGRAN / 2) : q / NGRAN, gi = WIDEP ? 2 * (q % (NGRAN / 2)) : q % NGRAN;   // (WIDEP: the first of the pair)
      const int peer = pi + (pi >= member ? 1 : 0);
      poll_off[i] = (unsigned)(peer * NGRAN + gi) * 8u;                              // bytes inside the group's parity slot
      scat_off[i] = (unsigned)((gi / HS) * (ROWS == 16 ? 2 : 4) * LS + peer * HS + gi % HS);   // (first) row of the granule in the tile
    }
  }
  // lean path: byte offsets of this lane's four rows at the current step (xproj; cbuf = /4, y = /8: same row index)
  unsigned xoff[RL];
#pragma unroll
  for (int r = 0; r < RL; ++r) {
    const int pos = dir == 0 ? 0 : len[r] - 1;
    xoff[r] = (unsigned)((((int64_t)bidx[r] * T + pos) * xrow + dir * 4 * H + unit0 * 4) * 4);
  }
  const int xstep = (dir == 0 ? 1 : -1) * (int)(xrow * 4);
  // fused input projection: A fragments of x_t.  A row a of the tile = C row a = utterance (a >> 2) * RL + (a & 3) of the slice
  // (rows with (a & 3) >= RL are padding: they read utterance 0 of the slice, their products are never looked at); lane
  // (l15 = a, lq) holds columns kc * 32 + 8 * lq .. + 8 -- clamped into the row: K_x's image is zero from row D on.
  bf16x8 xa[KXA];
  int xa_len = 0, xa_b = 0;
  unsigned xa_col[KXA];
  const char* xa_base = nullptr;
  int xa_step = 0;
  unsigned xa_off = 0;
  if constexpr (KX > 0) {
    const bool valid = (l15 & 3) < RL && slice * ROWS + (l15 >> 2) * RL + (l15 & 3) < B;
    xa_b = valid ? slice * ROWS + (l15 >> 2) * RL + (l15 & 3) : slice * ROWS;
    xa_len = valid ? min(length[xa_b], T) : 0;
#pragma unroll
    for (int kc = 0; kc < KX; ++kc) xa_col[kc] = (unsigned)min(kc * 32 + 8 * lq, fi.Dp - 8) * 2u;
    xa_base = reinterpret_cast<const char*>(fi.x + dir * fi.xdir);
    xa_step = (dir == 0 ? 1 : -1) * (int)(fi.ldx * 2);
    const int pos0 = (dir == 0 || xa_len == 0) ? 0 : xa_len - 1;
    xa_off = (unsigned)((((int64_t)xa_b * T + pos0) * fi.ldx) * 2);
    if ((int64_t)B * T * fi.ldx * 2 >= ((int64_t)1 << 32)) smin = 0;
#pragma unroll
    for (int kc = 0; kc < KX; ++kc) xa[kc] = *reinterpret_cast<const bf16x8*>(xa_base + xa_off + xa_col[kc]);
  }
  char* const xbase = reinterpret_cast<char*>(xproj);
  char* const cbase = reinterpret_cast<char*>(cbuf);
  char* const ybase = reinterpret_cast<char*>(y);

  int cur = 0;
  // Single-workgroup chains (G = 1: nothing to wait for between steps): lean steps keep their HBM stores (saved gates,
  // c, y) for one step and issue them behind the NEXT step's xproj load -- the vector-memory queue retires in order, and
  // that load, needed right after the MFMAs, would otherwise sit behind three stores (128 units: 1.07 -> 0.88 us per
  // step).  With G > 1 the stores already lie in the shadow of the exchange and moving them only lengthens the step.
  constexpr bool DEFER = (G == 1);
  float4 pend_g[UB][RL];
  float pend_c[UB][RL];
  unsigned short pend_h[UB][RL];
  unsigned pend_off[RL];
  bool pending = false;
  auto flush_pending = [&]() {
    if (pending && lead) {
#pragma unroll
      for (int ub = 0; ub < UB; ++ub)
#pragma unroll
        for (int r = 0; r < RL; ++r) {
          *reinterpret_cast<float4*>(xbase + pend_off[r] + ub * 256) = pend_g[ub][r];
          *reinterpret_cast<float*>(cbase + (pend_off[r] >> 2) + ub * 64) = pend_c[ub][r];
          *reinterpret_cast<unsigned short*>(ybase + (pend_off[r] >> 3) + ub * 32) = pend_h[ub][r];
        }
    }
    pending = false;
  };
  // fused input projection: accx = x_s K_x of the step about to run, formed from xa = x_s; then xa <- x_{s+1} is requested
  f32x4 accx[4][UB];
  auto input_products = [&](int s, auto lean_tag) {        // s: the step the products are for (xa holds x_s)
    constexpr bool LEAN = decltype(lean_tag)::value;
    if constexpr (KX > 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) accx[g][ub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KX; ++kc)
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
#pragma unroll
          for (int g = 0; g < 4; ++g) accx[g][ub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[kc], wx[ub][kc][g], accx[g][ub], 0, 0, 0);
      if constexpr (LEAN) {
        if (s + 1 < xa_len) xa_off += (unsigned)xa_step;      // (a row that ends here re-reads its last frame: never used)
#pragma unroll
        for (int kc = 0; kc < KX; ++kc) xa[kc] = *reinterpret_cast<const bf16x8*>(xa_base + xa_off + xa_col[kc]);
      } else {
        const int sn = s + 1;
        const int posn = sn < xa_len ? (dir == 0 ? sn : xa_len - 1 - sn) : 0;
        const int64_t o = (((int64_t)xa_b * T + posn) * fi.ldx) * 2;
#pragma unroll
        for (int kc = 0; kc < KX; ++kc) xa[kc] = *reinterpret_cast<const bf16x8*>(xa_base + o + xa_col[kc]);
      }
    }
  };
  auto step = [&](int s, auto lean_tag) {
    constexpr bool LEAN = decltype(lean_tag)::value;
    bool act[RL];
    int64_t rowoff[RL];
#pragma unroll
    for (int r = 0; r < RL; ++r) {
      act[r] = LEAN || s < len[r];
      if constexpr (!LEAN) {
        const int pos = dir == 0 ? s : len[r] - 1 - s;
        rowoff[r] = act[r] ? ((int64_t)bidx[r] * T + pos) : 0;
      }
    }
    LSTM_STAMP(0, s, 0);
    if constexpr (KX == 0) {
      // streamed xproj: the first step of a block of 256 / ROWS steps waits for the block's rows (normally long there: the product runs ahead)
      if (fi.ready != nullptr && s % (256 / ROWS) == 0 && !stream_wait(fi, group, s / (256 / ROWS))) fail_flag = 1;
    }
    // x_t K_x + b of this step: issued now, consumed after the MFMAs
    float4 xp[UB][RL];
    f32x4 acc[4][UB];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ub = 0; ub < UB; ++ub) acc[g][ub] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool did_x = (KX == 0);
    if constexpr (KX > 0) {
      // the input products of this step are issued inside the poll below, behind the first polling round's loads (eight MFMAs in
      // FRONT of the polls cost their issue time on every step: 0.78 against 0.70 ms per 800-step launch; behind the previous
      // step's granule stores likewise)
#pragma unroll
      for (int ub = 0; ub < UB; ++ub)
#pragma unroll
        for (int r = 0; r < RL; ++r) xp[ub][r] = bz[ub];
    } else {
#pragma unroll
      for (int ub = 0; ub < UB; ++ub)
#pragma unroll
        for (int r = 0; r < RL; ++r) {
          xp[ub][r] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (lead) {
            if constexpr (LEAN) xp[ub][r] = *reinterpret_cast<const float4*>(xbase + xoff[r] + ub * 256);
            else if (act[r]) xp[ub][r] = *reinterpret_cast<const float4*>(xproj + rowoff[r] * xrow + dir * 4 * H + (unit0 + ub * 16) * 4);
          }
        }
    }

    // all-gather h_{s-1}: peers' slices arrive as granules tagged with epoch s
    if constexpr (G > 1) if (s > 0) {
      const char* src = reinterpret_cast<const char*>(ex_group + (int64_t)((s - 1) & 1) * par_stride);
      unsigned short* hl = &hlds[cur][0][0];
      constexpr int CH = PER;                        // all of them in one polling round (512 units: 15 granules, 30 registers)
      static_assert(PER % CH == 0, "sweep chunking");
#pragma unroll
      for (int c0 = 0; c0 < PER; c0 += CH) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u64 v[WIDEP ? 1 : CH];
        u32x4 w[WIDEP ? CH : 1];
        // a round of wide polls: the loads, then ONE wait for all of them (inline assembly: the compiler does not know these loads
        // are in flight; nothing else is issued in between -- WIDEP implies KX == 0 -- and older loads have returned by then)
        auto wide_round = [&]() {
          // The compiler does not know that the destination registers are written later, by the memory system: nothing it emits
          // between a load and the wait may touch them -- phones-las_amd/build.py checks the generated assembly of every
          // instantiation for that ON EVERY BUILD, with the flags of that build (scripts/check_wide_polls.py; round 6),
          // and fails the build on a violation.  (The loads and the wait as ONE asm statement per round close the window by
          // construction; measured 0.86 -> 0.88 us per step at 256 units -- profiles/r06_fwd_helper_waves_ab.txt, fhw=0 against
          // r05 -- because nothing can be scheduled into the round any more: not taken.)
          // (a 64-bit VGPR address per load.  The scalar-base form -- voffset + an "s" operand made uniform with readfirstlane -- ran the
          // dense steps 0.01 us faster and FAULTED on address 0 in the steps of ragged batches, whose loop the compiler treats as
          // divergent; the same polls written as two 8-byte atomic loads ran there, so the defect is in how that operand is formed.)
#pragma unroll
          for (int i = 0; i < CH; ++i)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(w[i]) : "v"(src + poll_off[c0 + i]) : "memory");
#pragma unroll
          for (int i = 0; i < CH; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[i]) : : "memory");
        };
        unsigned spins = 0;
        bool ok = true;
        const unsigned want_tag = base + (unsigned)s;
        // the first polling round is issued, THEN the input products of this step (fused input projection) and the request
        // for x_{s+1}: the MFMAs run while the round is in flight.  One site, unconditional: with the products under the
        // "round came back empty" branch the compiler merged two copies of them through 16 accumulator moves and a
        // vmcnt(0) wait for x_{s+1} INSIDE the polling loop (0.80 against 0.70 ms per 800-step launch).
        auto poll_round = [&]() {
          bool all_in = true;
          if constexpr (WIDEP) {
            wide_round();
#pragma unroll
            for (int i = 0; i < CH; ++i) all_in = all_in && (w[i].y == want_tag) && (w[i].w == want_tag);
          } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) v[i] = granule_load(reinterpret_cast<const u64*>(src + poll_off[c0 + i]));
#pragma unroll
            for (int i = 0; i < CH; ++i) all_in = all_in && ((unsigned)(v[i] >> 32) == want_tag);
          }
          return all_in;
        };
        if constexpr (WIDEP) {
          wide_round();
        } else {
#pragma unroll
          for (int i = 0; i < CH; ++i) v[i] = granule_load(reinterpret_cast<const u64*>(src + poll_off[c0 + i]));
        }
        if constexpr (KX > 0) {
          if (c0 == 0) {
            __builtin_amdgcn_sched_barrier(0);
            input_products(s, lean_tag);
            did_x = true;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if constexpr (WIDEP) {
#pragma unroll
          for (int i = 0; i < CH; ++i) ok = ok && (w[i].y == want_tag) && (w[i].w == want_tag);
        } else {
#pragma unroll
          for (int i = 0; i < CH; ++i) ok = ok && ((unsigned)(v[i] >> 32) == want_tag);
        }
        while (!__all(ok)) {                         // wave-uniform loop: every lane re-polls until the whole wave is served
          if (++spins > SPIN_LIMIT) { fail_flag = 1; break; }
          __builtin_amdgcn_s_sleep(1);
          ok = poll_round();
        }
        if constexpr (WIDEP) {
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            // two adjacent units of the same row(s): one 32-bit LDS store per row
            *reinterpret_cast<unsigned*>(&hl[scat_off[c0 + i]]) = (w[i].x & 0xffffu) | (w[i].z << 16);
            if constexpr (GV == 2) *reinterpret_cast<unsigned*>(&hl[scat_off[c0 + i] + LS]) = (w[i].x >> 16) | (w[i].z & 0xffff0000u);
          }
        } else {
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const unsigned val = (unsigned)v[i];
            hl[scat_off[c0 + i]] = (unsigned short)(val & 0xffffu);
            if constexpr (GV == 2) hl[scat_off[c0 + i] + LS] = (unsigned short)(val >> 16);
          }
        }
      }
    }
    if constexpr (KX > 0) {
      if (!did_x) input_products(s, lean_tag);          // (the granules were there at once, the first step, single-workgroup chains)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) acc[g][ub] = accx[g][ub];
    }
    LSTM_STAMP(0, s, 1);
    flush_pending();                          // the previous step's stores, now that this step's polls are served
    lds_barrier();
    LSTM_STAMP(0, s, 2);
    if (fail_flag) return false;

    // all of the step's A fragments are requested before the first product (left to itself the compiler kept two in
    // flight and waited for the next pair right behind the products of the last: the LDS latency three more times per step)
    bf16x8 afr[KCW];
#pragma unroll
    for (int kc = 0; kc < KCW; ++kc) afr[kc] = *reinterpret_cast<const bf16x8*>(&hlds[cur][l15][(kh * KCW + kc) * 32 + 8 * lq]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kc = 0; kc < KCW; ++kc) {
#pragma unroll
      for (int ub = 0; ub < UB; ++ub)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g][ub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kc], wf[ub][kc][g], acc[g][ub], 0, 0, 0);
    }
    if constexpr (KS > 1) {
      // the upper-K wave hands its partial sums to the lead wave of the same unit block through LDS
      f32x4* rbuf = reinterpret_cast<f32x4*>(red);
      if (!lead) {
#pragma unroll
        for (int g = 0; g < 4; ++g) rbuf[((wave & 1) * 4 + g) * 64 + lane] = acc[g][0];
      }
      lds_barrier();
      if (lead) {
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g][0] += rbuf[((wave & 1) * 4 + g) * 64 + lane];
      }
    }
    LSTM_STAMP(0, s, 3);
    if (lead) {
      u64* dst = ex_group + (int64_t)(s & 1) * par_stride + (int64_t)member * NGRAN;
      float4 gsave[UB][RL];
      float csave[UB][RL];
#pragma unroll
      for (int ub = 0; ub < UB; ++ub) {
        const int unit = unit0 + ub * 16;
        const int ul = unit - member * HS;
        unsigned short hb[RL];
#pragma unroll
        for (int r = 0; r < RL; ++r) {
          const float gi = las_sigmoid(acc[0][ub][r] + xp[ub][r].x);
          const float gj = las_tanh(acc[1][ub][r] + xp[ub][r].y);
          const float gf = las_sigmoid(acc[2][ub][r] + xp[ub][r].z + 1.0f);
          const float go = las_sigmoid(acc[3][ub][r] + xp[ub][r].w);
          const float cn = gf * c[ub][r] + gi * gj;
          const unsigned short hn = las_f2bf(go * las_tanh(cn));
          gsave[ub][r] = make_float4(gi, gj, gf, go);
          csave[ub][r] = cn;
          if (act[r]) {
            c[ub][r] = cn;
            h[ub][r] = las_bf2f(hn);
          }
          hb[r] = act[r] ? hn : las_f2bf(h[ub][r]);
          hlds[cur ^ 1][lq * 4 + r][unit] = hb[r];
        }
        // the peers wait for these: they go out before the step's own HBM stores
        if constexpr (G > 1) {
#pragma unroll
          for (int rp = 0; rp < (RL + 1) / 2; ++rp)
            granule_store(dst + (lq * ((RL + 1) / 2) + rp) * HS + ul, base + (unsigned)(s + 1),
                          (unsigned)hb[2 * rp] | (GV == 2 ? (unsigned)hb[(2 * rp + 1) % RL] << 16 : 0u), local);
        }
      }
      LSTM_STAMP(0, s, 4);
#pragma unroll
      for (int ub = 0; ub < UB; ++ub) {
        const int unit = unit0 + ub * 16;
#pragma unroll
        for (int r = 0; r < RL; ++r) {
          if constexpr (LEAN && DEFER) {
            pend_g[ub][r] = gsave[ub][r];
            pend_c[ub][r] = csave[ub][r];
            pend_h[ub][r] = las_f2bf(h[ub][r]);
            pend_off[r] = xoff[r];
          } else if constexpr (LEAN) {
            *reinterpret_cast<float4*>(xbase + xoff[r] + ub * 256) = gsave[ub][r];
            *reinterpret_cast<float*>(cbase + (xoff[r] >> 2) + ub * 64) = csave[ub][r];
            *reinterpret_cast<unsigned short*>(ybase + (xoff[r] >> 3) + ub * 32) = las_f2bf(h[ub][r]);
          } else if (act[r]) {
            *reinterpret_cast<float4*>(xproj + rowoff[r] * xrow + dir * 4 * H + unit * 4) = gsave[ub][r];
            cbuf[rowoff[r] * yrow + dir * H + unit] = csave[ub][r];
            y[rowoff[r] * yrow + dir * H + unit] = las_f2bf(h[ub][r]);
          }
        }
      }
    }
    if constexpr (LEAN) {
      pending = DEFER;
#pragma unroll
      for (int r = 0; r < RL; ++r) xoff[r] += (unsigned)xstep;
    }
    if constexpr (G == 1) {
      // single-workgroup chains exchange nothing: one granule per step tells the companion how far the chain is
      // (every fourth step: the write-through store sits in the in-order memory queue of the wave that also loads)
      if (companions && tid == 0 && (s & 3) == 3) granule_store(ex_group + (int64_t)(s & 1) * par_stride, base + (unsigned)(s + 1), 0u, false);
    }
    cur ^= 1;
    LSTM_STAMP(0, s, 5);
    return true;
  };

  bool ok = true;
  int s = 0;
  for (; s < smin && ok; ++s) ok = step(s, std::true_type{});
  for (; s < smax && ok; ++s) ok = step(s, std::false_type{});
  flush_pending();

  if (tid == 0) __hip_atomic_store(done_word, ((u64)(base + 1u) << 32) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // releases the companion
  if (!ok) {
    if (tid == 0) atomicOr(status, 1u);
    return;
  }
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int r = 0; r < RL; ++r)
      if (bidx[r] < B && lead) {
        const int64_t o = ((int64_t)dir * B + bidx[r]) * H + unit0 + ub * 16;
        c_last[o] = c[ub][r];
        h_last[o] = h[ub][r];
      }
}

template <int H, int ROWS, int G = coop_members(H), int KX = 0>
__global__ __launch_bounds__(256) void lstm_fwd_kernel(float* __restrict__ xproj, const unsigned short* __restrict__ wpacked,
                                                       const int32_t* __restrict__ length, unsigned short* __restrict__ y,
                                                       float* __restrict__ cbuf, float* __restrict__ c_last,
                                                       float* __restrict__ h_last, u64* __restrict__ exch,
                                                       unsigned* __restrict__ status, int B, int T, int ndir, int ngroups,
                                                       int companions, long long exch_words, const FusedInput fi) {
  const unsigned base = launch_base(status);
  lstm_fwd_body<H, ROWS, G, KX>(xproj, wpacked, length, y, cbuf, c_last, h_last, exch, status, B, T, ndir, ngroups, companions, fi, base);
  launch_arrive(status, base, T, exch, exch_words);
}

// ------------------------------------------------------------------------------------------------
// backward in time
//
// Member m owns hidden units [m*HS, (m+1)*HS) and therefore the gate columns dz_t[:, own units x 4 gates]: it
// computes them from the saved gates (one step ahead in registers), keeps them in a small LDS tile and multiplies
// that tile with ITS rows of K_h^T: a partial dh_{t-1} for ALL H units (K = 4*HS).  The partial sums are
// reduce-scattered: the 16-unit tiles that belong to other members travel as fp32 granules {epoch, value} to the
// lane that will use them (same wave, same lane of the owner), own tiles stay in registers.  Per lane and step that
// is 4 x (G-1) granules each way, against 8 x 4 x (G-1) for an all-gather of dz_t.
// kh is K_h [H, 4H] in bf16 with GATE-INTERLEAVED columns (u*4+g): a member's K range is contiguous.
// ------------------------------------------------------------------------------------------------
// PACK (round 5; 8-row slices without the row split, i.e. two rows per lane): the two rows' partial sums of a tile travel as ONE
// granule {epoch, bf16 | bf16 << 16} instead of two fp32 granules -- half the exchange instructions on either side; the sums
// are still formed in fp32 (own partial in fp32 + the peers' bf16-rounded ones).
// HW (round 6; "helper waves"): the workgroup has EIGHT waves.  Waves 0-3 walk the chain as before but issue no HBM access of their
// own any more: waves 4-7 (one beside each chain wave on its SIMD, default priority) load the saved values of the step AFTER
// next, fold them into the gate-derivative coefficients and leave those in LDS (two slots), and copy the dz rows the chain
// waves put into the LDS tile out to HBM.  A chain wave's step is then: poll -> seven multiply-adds -> LDS store -> barrier ->
// products + sends; its vector-memory queue holds granules only.  Same arithmetic in the same order: results are bit-identical
// to the four-wave form.  The per-step barrier counts all eight waves; the helpers arrive a step's length early.
template <int H, int ROWS, int G, bool PACK = false, bool HW = false>     // ROWS: utterances per slice, G: members, as in lstm_fwd_kernel
__device__ __forceinline__ void lstm_bwd_body(const float* __restrict__ gates, const float* __restrict__ cbuf,
                                              const float* __restrict__ dy, const float* __restrict__ dc_last,
                                              const float* __restrict__ dh_last, const unsigned short* __restrict__ kh,
                                              const int32_t* __restrict__ length, unsigned short* __restrict__ dz,
                                              u64* __restrict__ exch, unsigned* __restrict__ status,
                                              int B, int T, int ndir, int ngroups, const unsigned base) {
  constexpr int HS = H / G;
  constexpr int NUB = HS / 16;                    // 16-unit blocks of a member
  constexpr bool SPLIT = NUB < 4;                 // H = 512: two waves share a unit block, two rows of every quad each
  static_assert(!SPLIT || NUB == 2, "row split is written for two unit blocks per member");
  constexpr int UBW = SPLIT ? 1 : NUB / 4;        // unit blocks a wave owns
  static_assert(ROWS == 16 || ROWS == 8 || (ROWS == 4 && !SPLIT), "4-row slices: only without the row split");
  constexpr int RPL = SPLIT ? ROWS / 8 : ROWS / 4;   // rows of its quad a lane owns (8-row slices: rows 0,1 of every quad; SPLIT:
                                                     // the two waves of a unit block take half of them each)
  constexpr int KCW = HS / 8;                     // k-chunks of the member's 4*HS gate columns
  constexpr int NT = SPLIT ? G / 2 : G * UBW;     // 16-unit output tiles a wave computes
  constexpr int OWN = SPLIT ? 0 : UBW;            // ... of which stay in registers (own units)
  constexpr int ZS = 4 * HS + lds_pad(G);         // LDS row stride of the dz tile (elements)
  constexpr int PAIR = NUB * 256;                 // granules per (destination, sender) pair: [block][row][lane]
  static_assert(!PACK || (!SPLIT && RPL == 2 && G > 1), "packed partial sums: two rows per lane, no row split");
  constexpr int RPG = PACK ? 1 : RPL;             // granules per (tile, lane)
  constexpr int PER = G > 1 ? (SPLIT ? G * RPL : (G - 1) * UBW * RPG) : 1;   // granules a lane polls per step
  static_assert(!HW || (!SPLIT && UBW * RPL <= 4), "helper waves: chains without the row split, at most four (row, unit) pairs per lane");
  __shared__ __attribute__((aligned(16))) unsigned short ztile[2][16][ZS];
  __shared__ int fail_flag;
  __shared__ int colo_flag;
  __shared__ __attribute__((aligned(16))) float pf_scratch[256];
  // HW: coefficients of a step, [slot][half][row of the lane][chain thread]: {ao, bc, ci, cj} and {cf, gf, dy, -}
  __shared__ __attribute__((aligned(16))) float4 coef_lds[HW ? 2 : 1][2][HW ? UBW * RPL : 1][HW ? 256 : 1];

  const int tid = HW ? (threadIdx.x & 255) : threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool helper = HW && threadIdx.x >= 256;          // waves 4-7: wave w + 4 serves chain wave w (same lanes, same rows and units)
  const int gstride = (ngroups + 7) & ~7;
  const int nblk = gstride * G;
  constexpr int CPG = group_companions_g(G);
  const bool companion = (int)blockIdx.x >= nblk;                // CPG per group (see lstm_fwd_kernel)
  const int cblk = companion ? blockIdx.x - nblk : blockIdx.x;
  const int per = 8 * (companion ? CPG : G), chunk = cblk / per, within = cblk % per;     // chunks of 8 groups (lstm_fwd_kernel)
  const int group = chunk * 8 + (within & 7), member = companion ? 0 : within >> 3;
  const int cm = companion ? within >> 3 : 0;
  if (group >= ngroups) return;
  const int nslices = ngroups / ndir;
  const int slice = group % nslices, dir = group / nslices;
  const int l15 = lane & 15, lq = lane >> 4;
  const int64_t grow = (int64_t)ndir * 4 * H;
  const int64_t yrow = (int64_t)ndir * H;
  const int64_t grp_gran = (int64_t)G * G * PAIR;               // granules of one group in one parity slot
  const int64_t par_stride = (int64_t)ngroups * grp_gran;
  u64* ex_group = exch + (int64_t)group * grp_gran;             // + parity*par_stride + (dest*G + sender)*PAIR + ...
  u64* done_word = exch + 2 * par_stride + (int64_t)ngroups * G + (int64_t)group * G + member;

  const int blk = SPLIT ? (wave & 1) : wave * UBW;              // first unit block of this wave
  const int hh = SPLIT ? (wave >> 1) : 0;                       // which row pair of every quad (SPLIT)
  int len[RPL], bidx[RPL];
  int smax = 0, smin = 0x7fffffff;
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    bidx[r] = slice * ROWS + lq * (ROWS / 4) + hh * RPL + r;
    len[r] = (bidx[r] < B) ? min(length[bidx[r]], T) : 0;
  }
  {
    const int ll = (l15 < ROWS && slice * ROWS + l15 < B) ? min(length[slice * ROWS + l15], T) : 0;     // all rows of the slice
    smax = (int)las_wave_max((float)ll);
    smin = -(int)las_wave_max((float)(-ll));
  }
  if ((int64_t)B * T * grow * 4 >= ((int64_t)1 << 32)) smin = 0;     // 32-bit byte offsets do not reach: general path only
  const int s_start = smax - 1;                  // this launch's steps: s_start down to 0

  if (companion) {
    // PREFETCH COMPANION (see lstm_fwd_kernel): gates, c and dy lines of the whole group, PF_AHEAD steps ahead.
    if (HW && threadIdx.x >= 256) return;        // (the companion workgroup of the eight-wave form: four waves do the work)
    constexpr int PF_AHEAD = 6;
    const int mylen = (l15 < ROWS && slice * ROWS + l15 < B) ? min(length[slice * ROWS + l15], T) : 0;
    // dz rows t >= length are zero: cleared here (the direction's gate columns), so dense batches need no memset of dz
    for (int rr = wave * (ROWS / 4); rr < (wave + 1) * (ROWS / 4); ++rr) {
      const int bb = slice * ROWS + rr;
      if (bb >= B) continue;
      const int ll = __builtin_amdgcn_readlane(mylen, rr);
      constexpr int LPR = 4 * H * 2 / 16 / CPG;
      for (int e = ll * LPR + lane; e < T * LPR; e += 64) {
        const int t = e / LPR, c = cm * LPR + e % LPR;
        *reinterpret_cast<uint4*>(dz + ((int64_t)bb * T + t) * grow + dir * 4 * H + c * 8) = make_uint4(0, 0, 0, 0);
      }
    }
    const u64* tag0 = ex_group + (G > 1 ? (int64_t)(1 * G + 0) * PAIR : 0);   // (destination 1, sender 0): member 0 writes it every step (G = 1: a progress granule)
    int seen = -1;
    for (int it = 0; it <= s_start; ++it) {
      const int sp = s_start - it;
      unsigned spins = 0;
      while (seen < it - PF_AHEAD) {
        const u64 v0 = granule_load(tag0), v1 = granule_load(tag0 + par_stride), dn = granule_load(done_word);
        if ((unsigned)(dn >> 32) == base + 1u) return;
        seen = max(max((int)((unsigned)(v0 >> 32) - base), (int)((unsigned)(v1 >> 32) - base)), 0) - 1;
        if (seen < it - PF_AHEAD) {
          if (++spins > SPIN_LIMIT) return;
          __builtin_amdgcn_s_sleep(8);
        }
      }
#pragma unroll
      for (int r4 = 0; r4 < ROWS / 4; ++r4) {
        const int rr = wave * (ROWS / 4) + r4;
        const int ll = __builtin_amdgcn_readlane(mylen, rr);
        if (sp < ll) {
          const int pos = dir == 0 ? sp : ll - 1 - sp;
          const int64_t R = (int64_t)(slice * ROWS + rr) * T + pos;
          const float* src = gates + R * grow + dir * 4 * H + cm * (4 * H / CPG) + lane * 4;
#pragma unroll
          for (int c4 = 0; c4 < H / 64 / CPG; ++c4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c4 * 256),
                                             (__attribute__((address_space(3))) void*)(pf_scratch), 16, 0, 0);
          constexpr int CW = H / CPG;                              // c and dy: this companion's H/CPG floats of each
#pragma unroll
          for (int c4 = 0; c4 < (CW + 255) / 256; ++c4) {
            const int col = cm * CW + (c4 * 256 + lane * 4) % CW;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cbuf + R * yrow + dir * H + col),
                                             (__attribute__((address_space(3))) void*)(pf_scratch), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dy + R * yrow + dir * H + col),
                                             (__attribute__((address_space(3))) void*)(pf_scratch), 16, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // latency-bound chain: ahead of co-resident GEMM waves (and of its own helper wave) in the issue arbitration
  if (!helper) __builtin_amdgcn_s_setprio(3);

  // register-resident B fragments: B[k][n] = K_h[n][member's gate columns k]: rows of K_h, contiguous 16-byte pieces
  const unsigned short* khd = kh + (int64_t)dir * H * 4 * H;
  int tile_dest[NT];
  bf16x8 wf[NT][KCW];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int dest = SPLIT ? hh * (G / 2) + j : (member + j / UBW) % G;      // own tiles first (not SPLIT)
    const int ublk = SPLIT ? blk : blk + j % UBW;
    tile_dest[j] = dest;
    const int n = dest * HS + ublk * 16 + l15;
    if (!helper) {
#pragma unroll
      for (int kc = 0; kc < KCW; ++kc)
        wf[j][kc] = *reinterpret_cast<const bf16x8*>(khd + (int64_t)n * 4 * H + member * 4 * HS + kc * 32 + 8 * lq);
    }
  }
  constexpr bool W_AGPR = NT * KCW * 4 > 128;       // (see lstm_fwd_kernel: weights beyond 128 registers live in AccVGPRs)

  const int unit0 = member * HS + blk * 16 + l15;                // + ub*16
  float dc[UBW][RPL], dh[UBW][RPL], part[UBW][RPL];
#pragma unroll
  for (int ub = 0; ub < UBW; ++ub)
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const bool ok = bidx[r] < B;
      const int64_t o = ((int64_t)dir * B + (ok ? bidx[r] : 0)) * H + unit0 + ub * 16;
      dc[ub][r] = (ok && dc_last) ? dc_last[o] : 0.f;
      dh[ub][r] = (ok && dh_last) ? dh_last[o] : 0.f;
      part[ub][r] = 0.f;
    }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): weights and initial state are in registers before the time loop
  if constexpr (W_AGPR) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int kc = 0; kc < KCW; ++kc) asm("" : "+a"(wf[j][kc]));
  }
  if (tid == 0) fail_flag = 0;
  __syncthreads();
  const bool local = xcd_colocated<G>(exch + 2 * par_stride + (int64_t)group * G, member, &colo_flag, status, base);

  // loop-invariant pieces of the exchange (byte offsets inside the group's parity slot)
  unsigned poll_off[PER], send_off[NT > OWN ? NT - OWN : 1];
  if constexpr (G > 1) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      int sender, ub, r;
      if constexpr (SPLIT) { sender = e / RPL; ub = 0; r = hh * RPL + e % RPL; }
      else { sender = (member + 1 + e / (UBW * RPG)) % G; ub = (e / RPG) % UBW; r = e % RPG; }
      poll_off[e] = (unsigned)((((member * G + sender) * NUB + blk + ub) * 4 + r) * 64 + lane) * 8u;
    }
#pragma unroll
    for (int j = OWN; j < NT; ++j) {
      const int ublk = SPLIT ? blk : blk + j % UBW;
      send_off[j - OWN] = (unsigned)((((tile_dest[j] * G + member) * NUB + ublk) * 4) * 64 + lane) * 8u;   // + r*512
    }
  }

  // lean path: byte offset of this lane's rows into gates (cbuf, dy = /4; dz = /2: same element index)
  unsigned goff[RPL];
  const int gstep = (dir == 0 ? 1 : -1) * (int)(grow * 4);
  const char* const gbase = reinterpret_cast<const char*>(gates);
  const char* const cbase = reinterpret_cast<const char*>(cbuf);
  const char* const dbase = reinterpret_cast<const char*>(dy);
  char* const zbase = reinterpret_cast<char*>(dz);

  struct Saved { float4 g; float ct, cp, dyv; };
  Saved sv[UBW][RPL];
  // general loader (any row may be finished); also used for the first lean step, which needs c_t from memory
  auto load_general = [&](int s) {
#pragma unroll
    for (int ub = 0; ub < UBW; ++ub) {
      const int unit = unit0 + ub * 16;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        Saved v{make_float4(0.f, 0.f, 0.f, 0.f), 0.f, 0.f, 0.f};
        if (s >= 0 && s < len[r]) {
          const int pos = dir == 0 ? s : len[r] - 1 - s;
          const int64_t ro = (int64_t)bidx[r] * T + pos;
          v.g = *reinterpret_cast<const float4*>(gates + ro * grow + dir * 4 * H + unit * 4);
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
  // lean loader for step s <= smin-2: every row is running, c_t is the c_{t-1} of the step before.  No branch in it: a
  // conditional load is a branch, the compiler put a vmcnt wait at its join and copied the loaded c_{t-1} into its home
  // register at once -- 0.25 us of stall per step, on the chain's critical path (phase stamps).  Step 0 reads c_0 in place
  // of the c_{-1} that does not exist; prepare() replaces it by 0.
  auto load_lean = [&](int s) {
    const unsigned cback = s > 0 ? (unsigned)gstep : 0u;
#pragma unroll
    for (int ub = 0; ub < UBW; ++ub)
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        sv[ub][r].ct = sv[ub][r].cp;
        sv[ub][r].g = *reinterpret_cast<const float4*>(gbase + goff[r] + ub * 256);
        sv[ub][r].cp = *reinterpret_cast<const float*>(cbase + ((goff[r] - cback) >> 2) + ub * 64);
        sv[ub][r].dyv = *reinterpret_cast<const float*>(dbase + (goff[r] >> 2) + ub * 64);
      }
  };
  auto set_goff = [&](int s) {       // offsets of step s (valid while every row is running)
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int pos = dir == 0 ? s : len[r] - 1 - s;
      goff[r] = (unsigned)((((int64_t)bidx[r] * T + pos) * grow + dir * 4 * H + unit0 * 4) * 4);
    }
  };

  // gate-derivative coefficients of a step: dz = f(dh) is then seven multiply-adds once dh arrives
  //   do = dht*ao, dct = dc + dht*bc, di = dct*ci, dj = dct*cj, df = dct*cf, dc' = dct*gf   (dht = dy + dh)
  struct Coef { float ao, bc, ci, cj, cf, gf, dyv; };
  Coef cf[UBW][RPL];
  auto prepare = [&](bool first_step = false) {   // from sv (waits for its loads): runs while the partial sums are in flight
#pragma unroll                                     // first_step: the coefficients of time step 0 (c_{-1} = 0)
    for (int ub = 0; ub < UBW; ++ub)
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const Saved v = sv[ub][r];
        const float tc = las_tanh(v.ct);
        Coef k;
        k.ao = tc * v.g.w * (1.f - v.g.w);
        k.bc = v.g.w * (1.f - tc * tc);
        k.ci = v.g.y * v.g.x * (1.f - v.g.x);
        k.cj = v.g.x * (1.f - v.g.y * v.g.y);
        k.cf = (first_step ? 0.f : v.cp) * v.g.z * (1.f - v.g.z);
        k.gf = v.g.z;
        k.dyv = v.dyv;
        cf[ub][r] = k;
      }
  };

  int cur = 0;
  unsigned epoch = 0;          // = iterations done; partial sums sent in iteration i carry tag i+1 in parity slot i&1
  bool ok = true;
  if constexpr (HW) {
    if (helper) {
      // HELPER WAVE.  Iteration s runs between the chain's barriers of steps s + 1 and s: coefficients of step s - 1 into slot
      // (s - 1) & 1 (the chain read that slot's previous content, step s + 1's, before the barrier of step s + 1), the dz rows of
      // step s + 1 from the tile the chain filled before that barrier (it refills this buffer in step s - 1, behind the barrier
      // of step s) out to HBM, then the request for step s - 2's saved values, which have the whole step to arrive.
      auto put_coefs = [&](int slot) {
#pragma unroll
        for (int ub = 0; ub < UBW; ++ub)
#pragma unroll
          for (int r = 0; r < RPL; ++r) {
            const Coef k = cf[ub][r];
            coef_lds[slot][0][ub * RPL + r][tid] = make_float4(k.ao, k.bc, k.ci, k.cj);
            coef_lds[slot][1][ub * RPL + r][tid] = make_float4(k.cf, k.gf, k.dyv, 0.f);
          }
      };
      auto store_dz = [&](int x) {                   // time step x of this lane's rows: LDS tile -> HBM
        const unsigned short* zt = &ztile[(s_start - x) & 1][0][0];
#pragma unroll
        for (int ub = 0; ub < UBW; ++ub) {
          const int unit = unit0 + ub * 16;
#pragma unroll
          for (int r = 0; r < RPL; ++r)
            if (x < len[r]) {
              const uint2 zv = *reinterpret_cast<const uint2*>(zt + (lq * 4 + hh * RPL + r) * ZS + (unit - member * HS) * 4);
              const int pos = dir == 0 ? x : len[r] - 1 - x;
              *reinterpret_cast<uint2*>(dz + ((int64_t)bidx[r] * T + pos) * grow + dir * 4 * H + unit * 4) = zv;
            }
        }
      };
      load_general(s_start);
      prepare();
      put_coefs(s_start & 1);
      load_general(s_start - 1);
      __syncthreads();
      for (int s = s_start; s >= 0; --s) {
        if (s >= 1) {
          prepare();                                 // from the values requested one iteration ago
          put_coefs((s - 1) & 1);
        }
        if (s < s_start) store_dz(s + 1);
        if (s >= 2) load_general(s - 2);
        lds_barrier();
        if (lds_flag(&fail_flag)) return;
      }
      store_dz(0);
      return;
    }
    __syncthreads();                                 // the first step's coefficients are in their slot
  } else {
    set_goff(smin > 0 ? smin - 1 : 0);
    load_general(s_start);
    prepare();
  }
  // one time step; LEAN (compile time): every row of the slice is running at step s (and at s - 1: the lean loader).
  // Two instantiations instead of a run-time flag: with both loaders in one body the values they load met in phi copies,
  // i.e. in waits for loads that are not needed before the end of the step.  Returns false when the chain is over.
  auto iter = [&](const int s, auto lean_tag) -> bool {
    constexpr bool lean = decltype(lean_tag)::value;
    LSTM_STAMP(2048, smax - 1 - s, 0);
    if constexpr (HW) {
      // this step's coefficients: requested from LDS now, in registers long before the peers' granules are
#pragma unroll
      for (int ub = 0; ub < UBW; ++ub)
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const float4 a = coef_lds[s & 1][0][ub * RPL + r][tid], b = coef_lds[s & 1][1][ub * RPL + r][tid];
          cf[ub][r] = Coef{a.x, a.y, a.z, a.w, b.x, b.y, b.z};
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- dh_s: own partial + the peers' (sent in the previous iteration, i.e. for time step s) ----
    if (epoch > 0) {
      float cand[UBW][RPL];
#pragma unroll
      for (int ub = 0; ub < UBW; ++ub)
#pragma unroll
        for (int r = 0; r < RPL; ++r) cand[ub][r] = part[ub][r];
      if constexpr (G > 1) {
        const char* src = reinterpret_cast<const char*>(ex_group + (int64_t)((epoch - 1) & 1) * par_stride);
        constexpr int CH = PER / ((PER + 15) / 16);      // polling rounds of at most 16 granules in flight (28 -> 2 x 14)
        static_assert(PER % CH == 0, "sweep chunking");
#pragma unroll
        for (int c0 = 0; c0 < PER; c0 += CH) {
          u64 v[CH];
          unsigned spins = 0;
          for (;;) {                                   // wave-uniform: every lane re-polls until the whole wave is served
            bool got = true;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
              v[i] = granule_load(reinterpret_cast<const u64*>(src + poll_off[c0 + i]));
              got = got && ((unsigned)(v[i] >> 32) == base + epoch);
            }
            if (__all(got)) break;
            if (++spins > SPIN_LIMIT) { fail_flag = 1; ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
          }
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const int e = c0 + i;
            if constexpr (PACK) {
              const int ub = e % UBW;
              cand[ub][0] += __uint_as_float((unsigned)v[i] << 16);
              cand[ub][1] += __uint_as_float((unsigned)v[i] & 0xffff0000u);
            } else {
              const int ub = SPLIT ? 0 : (e / RPL) % UBW, r = e % RPL;
              cand[ub][r] += __uint_as_float((unsigned)v[i]);
            }
          }
        }
      }
#pragma unroll
      for (int ub = 0; ub < UBW; ++ub)
#pragma unroll
        for (int r = 0; r < RPL; ++r)
          if (s + 1 < len[r]) dh[ub][r] = cand[ub][r];        // rows that were running at step s+1
    }

    LSTM_STAMP(2048, smax - 1 - s, 1);
    // next step's operands first: in flight during the gate math, the product and the exchange, and ahead of this
    // step's dz stores in the in-order vector-memory queue
    unsigned zoff[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) zoff[r] = HW ? 0u : goff[r];
    if constexpr (HW) {
      // (the helper wave has them)
    } else if constexpr (lean) {
      // unconditional (a block under `if (s > 0)` gave the loaded values a home register to be copied into, behind a wait for
      // the load: 0.2 us per step): the last step re-reads its own row, nobody uses what it gets
      const unsigned dec = s > 0 ? (unsigned)gstep : 0u;
#pragma unroll
      for (int r = 0; r < RPL; ++r) goff[r] -= dec;
      load_lean(s > 0 ? s - 1 : 0);
    } else {
      if (s > 0) load_general(s - 1);
    }
    LSTM_STAMP(2048, smax - 1 - s, 2);
    // ---- gate derivatives of step s -> dz (LDS tile for the product, HBM for the weight-gradient GEMMs) ----
    unsigned short* zl = &ztile[cur][0][0];
#pragma unroll
    for (int ub = 0; ub < UBW; ++ub) {
      const int unit = unit0 + ub * 16;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const bool act = lean || s < len[r];
        uint2 zv = make_uint2(0u, 0u);
        if (act) {
          const Coef k = cf[ub][r];              // everything that does not depend on dh was folded one step ahead
          const float dht = k.dyv + dh[ub][r];
          const float dov = dht * k.ao;
          const float dct = dc[ub][r] + dht * k.bc;
          const float di = dct * k.ci;
          const float dj = dct * k.cj;
          const float df = dct * k.cf;
          dc[ub][r] = dct * k.gf;
          zv.x = (unsigned)las_f2bf(di) | ((unsigned)las_f2bf(dj) << 16);
          zv.y = (unsigned)las_f2bf(df) | ((unsigned)las_f2bf(dov) << 16);
          if constexpr (HW) {
            // (the helper wave copies the row out of the LDS tile)
          } else if (lean) *reinterpret_cast<uint2*>(zbase + (zoff[r] >> 1) + ub * 128) = zv;
          else {
            const int pos = dir == 0 ? s : len[r] - 1 - s;
            *reinterpret_cast<uint2*>(dz + ((int64_t)bidx[r] * T + pos) * grow + dir * 4 * H + unit * 4) = zv;
          }
        }
        *reinterpret_cast<uint2*>(zl + (lq * 4 + hh * RPL + r) * ZS + (unit - member * HS) * 4) = zv;   // [row][u*4+g]
      }
    }
    LSTM_STAMP(2048, smax - 1 - s, 3);
    lds_barrier();
    LSTM_STAMP(2048, smax - 1 - s, 4);
    // Exchanging chains read the timeout flag together with the A fragments and test it when those arrive: a read-and-branch
    // right behind the barrier is an LDS round trip of its own on the critical path of every step (256 units: 1.10 -> 1.05 us
    // per step, 512 units: 2.60 -> 2.50).  The single-workgroup chains keep the early test (there the late one costs: 128
    // units 1.26 -> 1.42 us; so does it in the forward kernel, 0.87 -> 0.95: measured, left alone).
    int failed = 0;
    // (HW: a ds_read_b32.  The four-wave form keeps the generic volatile read it has had since round 3 -- compiled into a FLAT load
    // and a vmcnt(0) wait behind the barrier, which looks like a defect in the ISA and measures FASTER there: 1.02 us per step
    // against 1.09 with the LDS read, round 6, gpurun_out/r06_lstm_ab.log -- the wait happens to drain the queue at a cheap moment.)
    if constexpr (G > 1) failed = HW ? lds_flag(&fail_flag) : *reinterpret_cast<volatile int*>(&fail_flag);
    else if (fail_flag) { ok = false; return false; }
    if (s == 0) {                            // dh_{-1} is not needed
      if (failed) ok = false;
      return false;
    }

    // ---- partial dh_{s-1}[all units] = dz_s[16, own 4*HS] * K_h^T ----
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the tiles the peers are waiting for first (each sent as soon as it is complete), this member's own tiles last: their
    // MFMAs run while the granules are on their way
    bf16x8 afr[KCW];
#pragma unroll
    for (int kc = 0; kc < KCW; ++kc) afr[kc] = *reinterpret_cast<const bf16x8*>(zl + l15 * ZS + kc * 32 + 8 * lq);
    if (failed) { ok = false; return false; }
    if constexpr (G > 1) {
      char* dst = reinterpret_cast<char*>(ex_group + (int64_t)(epoch & 1) * par_stride);
#pragma unroll
      for (int j = OWN; j < NT; ++j) {
#pragma unroll
        for (int kc = 0; kc < KCW; ++kc) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kc], wf[j][kc], acc[j], 0, 0, 0);
        if constexpr (PACK) {
          granule_store(reinterpret_cast<u64*>(dst + send_off[j - OWN]), base + epoch + 1,
                        (unsigned)las_f2bf(acc[j][0]) | ((unsigned)las_f2bf(acc[j][1]) << 16), local);
        } else {
#pragma unroll
          for (int r = 0; r < (SPLIT ? ROWS / 4 : RPL); ++r)       // 8-row slices: rows 2, 3 of every quad carry nothing
            granule_store(reinterpret_cast<u64*>(dst + send_off[j - OWN] + r * 512), base + epoch + 1, __float_as_uint(acc[j][r]), local);
        }
      }
    }
    // own tiles: nobody waits for one of them in particular, so they advance together -- K chunk by K chunk -- instead of one
    // after the other (a tile alone is a chain of KCW dependent MFMAs: 128 units, single-workgroup chains, 16 in a row)
#pragma unroll
    for (int kc = 0; kc < KCW; ++kc)
#pragma unroll
      for (int j = 0; j < (G > 1 ? OWN : NT); ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kc], wf[j][kc], acc[j], 0, 0, 0);
    LSTM_STAMP(2048, smax - 1 - s, 5);
#pragma unroll
    for (int ub = 0; ub < OWN; ++ub)
#pragma unroll
      for (int r = 0; r < RPL; ++r) part[ub][r] = acc[ub][r];
    if constexpr (!HW) prepare(lean && s == 1);                 // coefficients of step s-1 (its operands were loaded above)
    LSTM_STAMP(2048, smax - 1 - s, 6);
    if constexpr (G == 1) {                  // (see lstm_fwd_kernel: the companion's pace)
      if (tid == 0 && (epoch & 3) == 3) granule_store(ex_group + (int64_t)(epoch & 1) * par_stride, base + epoch + 1, 0u, false);
    }
    ++epoch;
    cur ^= 1;
    return ok;
  };
  {
    int s = s_start;
    bool go = true;
    for (; s >= smin && go; --s) go = iter(s, std::false_type{});
    for (; s >= 0 && go; --s) go = iter(s, std::true_type{});
  }
  if (tid == 0) __hip_atomic_store(done_word, ((u64)(base + 1u) << 32) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!ok && tid == 0) atomicOr(status, 2u);
}

template <int H, int ROWS, int G = coop_members(H), bool PACK = false, bool HW = false>
__global__ __launch_bounds__(HW ? 512 : 256) void lstm_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ cbuf,
                                                       const float* __restrict__ dy, const float* __restrict__ dc_last,
                                                       const float* __restrict__ dh_last, const unsigned short* __restrict__ kh,
                                                       const int32_t* __restrict__ length, unsigned short* __restrict__ dz,
                                                       u64* __restrict__ exch, unsigned* __restrict__ status,
                                                       int B, int T, int ndir, int ngroups, long long exch_words) {
  const unsigned base = launch_base(status);
  lstm_bwd_body<H, ROWS, G, PACK, HW>(gates, cbuf, dy, dc_last, dh_last, kh, length, dz, exch, status, B, T, ndir, ngroups, base);
  launch_arrive(status, base, T, exch, exch_words);
}

struct CoopGeom { int nslices, ngroups, G, blocks, companions; size_t exch_bytes; };

int members(int H) { return coop_members(H); }

CoopGeom geom(int B, int H, int ndir, bool bwd, int rows = 16, int G = 0) {
  CoopGeom g;
  g.G = G > 0 ? G : members(H);
  g.nslices = (B + rows - 1) / rows;
  g.ngroups = g.nslices * ndir;
  g.blocks = ((g.ngroups + 7) & ~7) * g.G;        // group stride rounded up to 8 (idle blocks exit at once)
  // granules per parity slot: forward all-gather of h_t: ngroups x G members x (rows/2)*HS; backward reduce-scatter of
  // the partial dh: ngroups x G x G (destination, sender) pairs x NUB*256
  const size_t HS = H / g.G;
  const size_t per_parity = bwd ? (size_t)g.ngroups * g.G * g.G * (HS / 16) * 256 : (size_t)g.ngroups * g.G * (rows >= 8 ? rows / 2 : rows) * HS;
  g.exch_bytes = (2 * per_parity + (size_t)g.ngroups * g.G + g.blocks) * sizeof(u64);   // + XCC-id table + done words (G = 1: progress granules + done words)
  g.companions = ((g.ngroups + 7) & ~7) * group_companions_g(g.G);   // prefetch companions (blocks nblk ...)
  return g;
}

// Utterances per slice: the shortest (4, 8, 16; see lstm_fwd_kernel) whose chains, with their companions, still find
// a CU each (256 on MI355X); LAS_LSTM_ROWS=16 / 8 / 4 forces one (tests, diagnostics).
int slice_rows(int B, int H, int ndir, bool bwd) {
  const int forced = las_knob("LAS_LSTM_ROWS", 0);
  if (H > 256 && members(H) >= 16) {    // the 32-unit-member kernels (512 units as 16 members, 1024 as 32: K split, row split): 16 or 8 rows
    if (forced == 16 || forced == 8) return forced;
    return H > 512 ? 8 : 16;            // (1024 units on full tiles spill 200-500 bytes per lane: half tiles by default)
  }
  if (forced == 16 || forced == 8 || forced == 4) return forced;
  // every chain workgroup and every companion should find a CU of its own (256 on MI355X; fewer in a partitioned
  // mode); the backward leaves three eighths of them to the weight-gradient products that run beside it
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  const int budget = bwd ? cus * 5 / 8 : cus;
  for (int rows = 4; rows <= 8; rows *= 2) {
    const CoopGeom g = geom(B, H, ndir, bwd, rows);
    if (g.blocks + g.companions <= budget) return rows;
  }
  return 16;
}

// LAS_LSTM_PREFETCH=0 launches the recurrent kernels without their prefetch companions (diagnostics)
int prefetch_mode() { return las_knob("LAS_LSTM_PREFETCH", 1) != 0; }

// 64-bit words of the workspace behind its header: what the last workgroup clears before the launch tags would wrap (the
// whole exchange area of las_lstm_workspace_bytes, not only this launch's layout: forward and backward launches of every layer
// share it)
long long exch_words(int B, int H, int ndir) { return (long long)((las_lstm_workspace_bytes(B, H, ndir) - 64) / sizeof(u64)); }

template <int H, int ROWS, int G, int KX = 0>
int launch_fwd_as(float* xproj, const las_bf16* wp, const int32_t* length, las_bf16* y, float* cbuf, float* c_last, float* h_last,
                  void* ws, int B, int T, int ndir, hipStream_t st, const FusedInput& fi) {
  const CoopGeom g = geom(B, H, ndir, false, ROWS, G);
  unsigned* status = reinterpret_cast<unsigned*>(ws);
  u64* exch = reinterpret_cast<u64*>(reinterpret_cast<char*>(ws) + 64);
  const int pf = prefetch_mode();
  // A streamed input product runs beside this launch (fi.ready): asking for LDS the kernel does not use keeps its workgroups
  // (72 KiB each) off the CUs of the chain's workgroups, as the backward launch does for the weight-gradient products.
  size_t hog = 0;
  if (fi.ready != nullptr) {
    static int hog_kb = -1;
    if (hog_kb < 0) {
      constexpr int static_kb = (2 * 16 * (H + lds_pad(G)) * 2 + 1024 + 4096 + 1023) / 1024 + 2;
      hog_kb = las_knob("LAS_STREAM_HOG_KB", 160 - static_kb - 6);     // (diagnostics: 0 lets the product's workgroups share the chain's CUs)
      if (hog_kb > 160 - static_kb - 2) hog_kb = 160 - static_kb - 2;
      if (hog_kb > 0) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_kernel<H, ROWS, G, KX>), hipFuncAttributeMaxDynamicSharedMemorySize, hog_kb * 1024);
    }
    hog = (size_t)(hog_kb > 0 ? hog_kb : 0) * 1024;
  }
  hipLaunchKernelGGL((lstm_fwd_kernel<H, ROWS, G, KX>), dim3(pf ? g.blocks + g.companions : g.blocks), dim3(256), hog, st, xproj, wp, length, y,
                     cbuf, c_last, h_last, exch, status, B, T, ndir, g.ngroups, pf, exch_words(B, H, ndir), fi);
  LAS_LAUNCH_CHECK("lstm fwd launch");
  return LAS_OK;
}

// widths the fused input projection is built for: 32-deep chunks of the (padded) feature count
constexpr bool fused_input_units(int H) { return H == 128 || H == 256 || H == 512; }
int fused_input_chunks(int H, int Dp) {
  if (!fused_input_units(H) || Dp < 8 || Dp % 8 || Dp > 96) return 0;
  return Dp <= 64 ? 2 : 3;
}

template <int H>
int launch_fwd(float* xproj, const las_bf16* wp, const int32_t* length, las_bf16* y, float* cbuf, float* c_last,
               float* h_last, void* ws, int B, int T, int ndir, hipStream_t st, const FusedInput& fi) {
  const int rows = fi.rows > 0 ? fi.rows : slice_rows(B, H, ndir, false);
  const int kx = fi.x ? fused_input_chunks(H, fi.Dp) : 0;
#define LAS_FWD(R, GG) do {                                                                                                   \
    if constexpr (fused_input_units(H)) {                                                                                      \
      if (kx == 2) return launch_fwd_as<H, R, GG, 2>(xproj, wp, length, y, cbuf, c_last, h_last, ws, B, T, ndir, st, fi);      \
      if (kx == 3) return launch_fwd_as<H, R, GG, 3>(xproj, wp, length, y, cbuf, c_last, h_last, ws, B, T, ndir, st, fi);      \
    }                                                                                                                          \
    return launch_fwd_as<H, R, GG, 0>(xproj, wp, length, y, cbuf, c_last, h_last, ws, B, T, ndir, st, fi);                     \
  } while (0)
#define LAS_FWD0(R, GG) return launch_fwd_as<H, R, GG, 0>(xproj, wp, length, y, cbuf, c_last, h_last, ws, B, T, ndir, st, fi)
  constexpr int G0 = coop_members(H);
  if constexpr (H == 1024) {            // (32-unit members: K split over wave pairs, no fused input projection, 8- or 16-row slices)
    if (rows == 8) LAS_FWD0(8, G0);
    LAS_FWD0(16, G0);
  } else {
    if (rows == 8) LAS_FWD(8, G0);
    if (rows == 4) LAS_FWD(4, G0);
    LAS_FWD(16, G0);
  }
#undef LAS_FWD
#undef LAS_FWD0
}

template <int H, int ROWS, int G, bool PACK = false, bool HW = false>
int launch_bwd_as(const float* gates, const float* cbuf, const float* dy, const float* dc_last, const float* dh_last,
                  const las_bf16* kh, const int32_t* length, las_bf16* dz, void* ws, int B, int T, int ndir, hipStream_t st) {
  const CoopGeom g = geom(B, H, ndir, true, ROWS, G);
  unsigned* status = reinterpret_cast<unsigned*>(ws);
  u64* exch = reinterpret_cast<u64*>(reinterpret_cast<char*>(ws) + 64);
  const int pf = prefetch_mode();
  // The weight-gradient GEMMs of the layer above run beside this kernel on the second stream.  Asking for LDS the
  // kernel does not use keeps their workgroups (32 KiB of LDS each) off the CUs of the chain's workgroups: measured
  // 7.96 ms per metric-M step with them kept off, 8.5 ms when one fits beside a chain workgroup.  The request is what
  // the CU's 160 KiB leave after the kernel's own static LDS (dz tile 2 x 16 x (4*HS + 8) bf16, DMA scratch), minus a
  // margin smaller than any GEMM workgroup's need.
  static int hog_kb = -1;
  if (hog_kb < 0) {
    constexpr int HS_ = H / G;
    constexpr int coef_bytes = HW ? 2 * 2 * ((HS_ >= 64 ? HS_ / 64 : 1) * (ROWS / 4)) * 256 * 16 : 16;      // the helper waves' coefficient slots
    constexpr int static_kb = (2 * 16 * (4 * HS_ + lds_pad(G)) * 2 + 1024 + coef_bytes + 1023) / 1024 + 1;
    hog_kb = las_knob("LAS_LSTM_BWD_LDS_KB", 160 - static_kb - 6);
    if (hog_kb > 160 - static_kb - 2) hog_kb = 160 - static_kb - 2;
    if (hog_kb > 0)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_kernel<H, ROWS, G, PACK, HW>), hipFuncAttributeMaxDynamicSharedMemorySize, hog_kb * 1024);
  }
  hipLaunchKernelGGL((lstm_bwd_kernel<H, ROWS, G, PACK, HW>), dim3(pf ? g.blocks + g.companions : g.blocks), dim3(HW ? 512 : 256), (size_t)(hog_kb > 0 ? hog_kb : 0) * 1024, st,
                     gates, cbuf, dy, dc_last, dh_last, kh, length, dz, exch, status, B, T, ndir, g.ngroups, exch_words(B, H, ndir));
  LAS_LAUNCH_CHECK("lstm bwd launch");
  return LAS_OK;
}

template <int H>
int launch_bwd(const float* gates, const float* cbuf, const float* dy, const float* dc_last, const float* dh_last,
               const las_bf16* kh, const int32_t* length, las_bf16* dz, void* ws, int B, int T, int ndir, hipStream_t st) {
  const int rows = slice_rows(B, H, ndir, true);
  constexpr int G0 = coop_members(H);
#define LAS_BWD(R, PACK, HW) return launch_bwd_as<H, R, G0, PACK, HW>(gates, cbuf, dy, dc_last, dh_last, kh, length, dz, ws, B, T, ndir, st)
  // Eight-wave form (lstm_bwd_body, HW; round 6): helper waves own the HBM side of a step.  256 units 1.02 -> 0.88 us per step,
  // 128 units 1.05 -> 0.96 (profiles/r06_lstm_helper_waves_ab.txt, r06_hw128_ab.txt); not at 64 units (0.60 -> 0.66: a step is
  // too short for the eight-wave barrier), not at 512 / 1024 (a wave's 256 weight registers leave no room for a second wave on
  // its SIMD), not on 16-row slices (four (row, unit) pairs per lane: registers).  LAS_LSTM_BWD_HW=0: the four-wave form (A/B).
  const bool hw = (H == 256 || H == 128) && las_knob("LAS_LSTM_BWD_HW", 1) != 0;
  if constexpr (H == 256 || H == 512) {
    // exchanging chains without the row split.  On 8-row slices the partial sums travel as bf16 PAIRS (PACK, round 5: 512 units
    // 2.46 -> 2.06 us per step, metric-L 17.37 -> 16.67 ms; worst listener gradient against the oracle's bf16 / f64 model
    // 1.85e-3 -> 1.92e-3 / 1.166e-2 -> 1.173e-2 of max-abs; the fp32-granule form was removed in round 6).
    if (rows == 4) {
      if constexpr (H == 256) { if (hw) LAS_BWD(4, false, true); }
      LAS_BWD(4, false, false);
    }
    if (rows == 8) {
      if constexpr (H == 256) { if (hw) LAS_BWD(8, true, true); }
      LAS_BWD(8, true, false);
    }
    LAS_BWD(16, false, false);
  } else if constexpr (H == 128) {
    if (rows == 4) { if (hw) LAS_BWD(4, false, true); LAS_BWD(4, false, false); }
    if (rows == 8) { if (hw) LAS_BWD(8, false, true); LAS_BWD(8, false, false); }
    LAS_BWD(16, false, false);
  } else if constexpr (H == 64) {
    if (rows == 4) LAS_BWD(4, false, false);
    if (rows == 8) LAS_BWD(8, false, false);
    LAS_BWD(16, false, false);
  } else {                              // 1024 units: 32-unit members, rows of a quad split over wave pairs
    if (rows == 8) LAS_BWD(8, false, false);
    LAS_BWD(16, false, false);
  }
#undef LAS_BWD
}

bool supported_units(int H) { return H == 64 || H == 128 || H == 256 || H == 512 || H == 1024; }

}  // namespace

extern "C" int las_lstm_slice_rows(int B, int H, int ndir) {
  if (!supported_units(H) || B <= 0) return 0;
  return slice_rows(B, H, ndir, false);
}

extern "C" int las_lstm_fwd_workgroups(int B, int H, int ndir) {
  if (!supported_units(H) || B <= 0) return 0;
  const CoopGeom g = geom(B, H, ndir, false, slice_rows(B, H, ndir, false));
  return g.blocks + (prefetch_mode() ? g.companions : 0);
}

extern "C" size_t las_lstm_workspace_bytes(int B, int H, int ndir) {
  if (!supported_units(H) || B <= 0) return 0;
  size_t n = 0;
  for (int rows = 4; rows <= 16; rows *= 2)             // every slice height the launches may take
    for (int bwd = 0; bwd < 2; ++bwd) {
      const size_t e = geom(B, H, ndir, bwd != 0, rows).exch_bytes;
      if (e > n) n = e;
    }
  return 64 + n;
}

extern "C" int las_lstm_pack_recurrent(const float* kernel_h, int H, las_bf16* packed, void* stream) {
  LAS_REQUIRE(supported_units(H), "las_lstm_pack_recurrent: num_units must be 64, 128, 256, 512 or 1024 (got %d)", H);
  hipLaunchKernelGGL(pack_recurrent_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, kernel_h, H, packed);
  LAS_LAUNCH_CHECK("pack launch");
  return LAS_OK;
}

namespace {
int recurrent_fwd(float* xproj, const las_bf16* wpacked, const int32_t* length, las_bf16* y, float* cbuf, float* c_last,
                  float* h_last, void* workspace, int B, int T, int H, int ndir, void* stream, const FusedInput& fi) {
  LAS_REQUIRE(B > 0 && T > 0 && (ndir == 1 || ndir == 2), "las_lstm_recurrent_fwd: bad shape B=%d T=%d ndir=%d", B, T, ndir);
  LAS_REQUIRE(supported_units(H), "las_lstm_recurrent_fwd: num_units %d not in {64,128,256,512,1024}", H);
  LAS_REQUIRE(workspace != nullptr && ((uintptr_t)workspace % 16 == 0), "las_lstm_recurrent_fwd: workspace missing or misaligned");
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  if (!prefetch_mode()) {       // otherwise the companions clear the rows beyond each length
    rc = las_check_hip(hipMemsetAsync(y, 0, (size_t)B * T * ndir * H * sizeof(las_bf16), st), "memset y");
    if (rc) return rc;
  }
  // No memset of the exchange buffer: the tags of a launch start from the launch base in the workspace header (see
  // launch_arrive); the caller zeroes the workspace ONCE, when it allocates it.  The first word of the header is the STICKY
  // timeout status (it survives until the host reads and clears it: one workspace serves every layer, forward and backward).
  switch (H) {
    case 64: return launch_fwd<64>(xproj, wpacked, length, y, cbuf, c_last, h_last, workspace, B, T, ndir, st, fi);
    case 128: return launch_fwd<128>(xproj, wpacked, length, y, cbuf, c_last, h_last, workspace, B, T, ndir, st, fi);
    case 512: return launch_fwd<512>(xproj, wpacked, length, y, cbuf, c_last, h_last, workspace, B, T, ndir, st, fi);
    case 1024: return launch_fwd<1024>(xproj, wpacked, length, y, cbuf, c_last, h_last, workspace, B, T, ndir, st, fi);
    default: return launch_fwd<256>(xproj, wpacked, length, y, cbuf, c_last, h_last, workspace, B, T, ndir, st, fi);
  }
}
}  // namespace

extern "C" int las_lstm_recurrent_fwd(float* xproj, const las_bf16* wpacked, const int32_t* length, las_bf16* y,
                                      float* cbuf, float* c_last, float* h_last, void* workspace, int B, int T, int H,
                                      int ndir, void* stream) {
  return recurrent_fwd(xproj, wpacked, length, y, cbuf, c_last, h_last, workspace, B, T, H, ndir, stream, FusedInput{});
}

extern "C" int las_lstm_fused_input_chunks(int H, int Dp) { return supported_units(H) ? fused_input_chunks(H, Dp) : 0; }

extern "C" int las_lstm_pack_input(const float* kernel, int D, int H, int chunks, las_bf16* packed, void* stream) {
  LAS_REQUIRE(supported_units(H) && D > 0 && chunks > 0 && D <= 32 * chunks, "las_lstm_pack_input: bad shape D=%d H=%d chunks=%d", D, H, chunks);
  hipLaunchKernelGGL(pack_input_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, kernel, D, H, chunks, packed);
  LAS_LAUNCH_CHECK("pack input launch");
  return LAS_OK;
}

extern "C" int las_lstm_recurrent_fwd_ex(const las_lstm_fwd* p, void* stream) {
  LAS_REQUIRE(p != nullptr, "las_lstm_recurrent_fwd_ex: null argument");
  FusedInput fi{};
  if (p->x != nullptr) {
    LAS_REQUIRE(p->kx_packed != nullptr && p->bias != nullptr, "las_lstm_recurrent_fwd_ex: the fused input projection needs kx_packed and bias");
    LAS_REQUIRE(supported_units(p->H) && fused_input_chunks(p->H, p->Dp) > 0, "las_lstm_recurrent_fwd_ex: no fused input projection for H=%d, Dp=%d "
                "(las_lstm_fused_input_chunks)", p->H, p->Dp);
    LAS_REQUIRE(p->ldx >= p->Dp && p->ldx % 8 == 0 && ((uintptr_t)p->x % 16 == 0), "las_lstm_recurrent_fwd_ex: x rows must be 16-byte aligned (ldx %% 8 == 0)");
    LAS_REQUIRE(p->ready == nullptr, "las_lstm_recurrent_fwd_ex: fused and streamed input projections exclude each other");
    fi.x = reinterpret_cast<const unsigned short*>(p->x); fi.ldx = p->ldx; fi.xdir = p->x_dir_stride;
    fi.kxp = reinterpret_cast<const unsigned short*>(p->kx_packed); fi.bias = p->bias; fi.Dp = p->Dp;
  }
  if (p->ready != nullptr) {
    LAS_REQUIRE(p->ready_count > 0, "las_lstm_recurrent_fwd_ex: ready_count = column tiles per block of the streamed product");
    const int rows = p->rows_per_slice > 0 ? p->rows_per_slice : slice_rows(p->B, p->H, p->ndir, false), sbs = 256 / rows;
    fi.ready = p->ready; fi.nsb = (p->T + sbs - 1) / sbs;
    fi.flags = 16 + ((((p->B + rows - 1) / rows) * p->ndir + 15) & ~15);
    fi.want = (unsigned)p->ready_count;
  }
  if (p->rows_per_slice != 0) {
    const int r = p->rows_per_slice;
    LAS_REQUIRE(r == 16 || r == 8 || (r == 4 && members(p->H) < 16),
                "las_lstm_recurrent_fwd_ex: rows_per_slice %d is not a slice height of the %d-unit kernels", r, p->H);
    fi.rows = r;
  }
  return recurrent_fwd(p->xproj, p->wpacked, p->length, p->y, p->cbuf, p->c_last, p->h_last, p->workspace, p->B, p->T, p->H, p->ndir, stream, fi);
}

namespace {
int recurrent_bwd(const float* gates, const float* cbuf, const float* dy, const float* dc_last, const float* dh_last,
                  const las_bf16* kh_bf16, const int32_t* length, las_bf16* dz, void* workspace, int B, int T, int H, int ndir,
                  void* stream) {
  LAS_REQUIRE(B > 0 && T > 0 && (ndir == 1 || ndir == 2), "las_lstm_recurrent_bwd: bad shape");
  LAS_REQUIRE(supported_units(H), "las_lstm_recurrent_bwd: num_units %d not in {64,128,256,512,1024}", H);
  LAS_REQUIRE(workspace != nullptr && ((uintptr_t)workspace % 16 == 0), "las_lstm_recurrent_bwd: workspace missing or misaligned");
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  if (!prefetch_mode()) {
    rc = las_check_hip(hipMemsetAsync(dz, 0, (size_t)B * T * ndir * 4 * H * sizeof(las_bf16), st), "memset dz");
    if (rc) return rc;
  }
  // (no memset of the exchange buffer: launch epochs, see las_lstm_recurrent_fwd)
  switch (H) {
    case 64: return launch_bwd<64>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, workspace, B, T, ndir, st);
    case 128: return launch_bwd<128>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, workspace, B, T, ndir, st);
    case 512: return launch_bwd<512>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, workspace, B, T, ndir, st);
    case 1024: return launch_bwd<1024>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, workspace, B, T, ndir, st);
    default: return launch_bwd<256>(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, workspace, B, T, ndir, st);
  }
}
}  // namespace

extern "C" int las_lstm_recurrent_bwd(const float* gates, const float* cbuf, const float* dy, const float* dc_last,
                                      const float* dh_last, const las_bf16* kh_bf16, const int32_t* length,
                                      las_bf16* dz, void* workspace, int B, int T, int H, int ndir, void* stream) {
  return recurrent_bwd(gates, cbuf, dy, dc_last, dh_last, kh_bf16, length, dz, workspace, B, T, H, ndir, stream);
}

#ifdef LAS_STAMPS
extern "C" int las_debug_read_lstm_stamps(unsigned long long* out_host, int n) {
  return las_check_hip(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(las_lstm_stamps), sizeof(unsigned long long) * (size_t)n), "read stamps");
}
#endif
