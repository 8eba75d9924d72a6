// bf16 x bf16 -> fp32 MFMA GEMMs for the LAS path (gfx950).
//
//   NT : C[M,N] (=|+=) A[M,K] * B[N,K]^T + bias      (forward projections, dX)
//   TN : C[M,N] += sum_k A[k,M] * B[k,N]             (weight gradients; K slices through a workspace or fp32 atomics)
//
// Both stage BMxBK / BNxBK operand tiles into LDS as [row][k] (k contiguous, row stride BK+8
// elements = 144 B: the 16 rows one ds_read_b128 lane group touches land on 16 disjoint 4-bank
// groups) and feed v_mfma_f32_16x16x32_bf16.  256 threads = 4 waves in a 2x2 grid, each wave
// owns a (BM/2)x(BN/2) block of 16x16 accumulator tiles.  Global->register loads of tile k+1 are
// issued before the MFMAs of tile k and written to the other LDS buffer after them (one
// barrier per K-tile).
#include "las_common.h"

namespace {

struct GemmArgs {
  const unsigned short* A;
  const unsigned short* B;
  void* C;
  const float* bias;
  int64_t lda, ldb, ldc, sa, sb, sc;
  int M, N, K;
  int out_bf16, accumulate, atomic, split_k;
  int a_shift, period;
  int c_perm_h;   // > 0: output column n (gate-interleaved index u*4+g) is stored at TF column g*H+u, H = c_perm_h
  // fused LSTM weight gradient (gemm_tn_tr_kernel only): rows [0,M1) of the A^T side come from A (no shift), rows
  // [M1, M1+M2) from A2 (shifted by a_shift inside `period`), row M1+M2 is all ones when bias_row != nullptr (its
  // output row, the column sums of B, goes to bias_row).  M = M1 + M2 (+1).
  const unsigned short* A2 = nullptr;
  int64_t lda2 = 0;
  int M1 = -1, M2 = 0;
  float* bias_row = nullptr;
  // split-K without atomics (gemm_tn_tr_kernel): slice z stores its fp32 tile at partial + z * partial_stride
  // ([M][N], plain column order); tn_reduce_kernel folds the slices into C / bias_row afterwards.
  float* partial = nullptr;
  int64_t partial_stride = 0;
  // las_gemm_nt_masked: element (row, col) of the product is multiplied by the input-dropout mask of the cell that read it
  // (1 / keep where las_uniform(seed, stream, row * N + col) < keep, else 0) before it is stored or accumulated
  float drop_keep = 1.0f;
  unsigned drop_seed = 0, drop_stream = 0;
  // las_gemm_nt_stream (gemm_nt_ring_kernel<..., STREAM>): the rows are (utterance, step) pairs handed out in the order a
  // recurrence beside this launch consumes them; see the kernel
  const int32_t* length = nullptr;
  unsigned* ready = nullptr;
  int sB = 0, sT = 0, s_ndir = 1, s_nsb = 0, sR = 4, s_nslices = 0;
  int64_t s_astride = 0;      // direction d reads A + d * s_astride (each direction's cell behind its own input-dropout mask)
};

// Layout of the `ready` buffer shared by las_gemm_nt_stream and the recurrence it feeds (32-bit words, zero before both launches):
//   [0, 8)                      per-XCD tile queues (next tile index of the XCD's list)
//   [8]                         the launch's handshake decision (gemm_nt_ring_kernel, STREAM): 1 = XCD-local lists, 2 = spread form
//   [16, 16 + ngroups)          written by the RECURRENCE at its start: group g = dir * nslices + slice runs on XCD x -> x + 1;
//                               its members are spread over several XCDs -> LAS_STREAM_SPREAD
//   [flags, flags + ngroups*nsb) column tiles finished of (group, step block): flags = 16 + ngroups rounded up to 16
constexpr unsigned LAS_STREAM_SPREAD = 0x100u;
__host__ __device__ inline int las_stream_flags_offset(int ngroups) { return 16 + ((ngroups + 15) & ~15); }
__device__ __forceinline__ float drop_scale(const GemmArgs& g, int row, int col) {
  return las_uniform(g.drop_seed, g.drop_stream, (unsigned long long)row * g.N + col) < g.drop_keep ? 1.0f / g.drop_keep : 0.f;
}

// LDS-DMA: 16 bytes per lane from a per-lane global address to (wave-uniform LDS base in M0) + lane * 16; invisible to the
// compiler's vmcnt bookkeeping (the ring kernels count their loads themselves)
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// KB = depth of a staged K tile: 64, or 32 for the large NT products (half the LDS, so three workgroups share a CU
// instead of two: these products wait on memory most of the time, and the epilogue of one overlaps the others' loops)
template <int BM, int BN, bool TN, int KB = 64>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int BK = KB, LDK = KB + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);            // [2][BM][LDK]
  unsigned short* Bs = As + 2 * BM * LDK;                                  // [2][BN][LDK]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (non-coherent L2s), so the N tiles that share
  // one block of A rows should be 8 linear ids apart: id = x + gx*y  ->  row block (id / (8 gx))*8 + id % 8, column
  // tile (id / 8) % gx.  Each XCD then fetches an A block once instead of every XCD fetching every A block.
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    const int id = bx + gx * by;
    const int full = (gy / 8) * 8 * gx;            // ids covered by complete groups of 8 row blocks
    if (id < full) {
      by = (id / (8 * gx)) * 8 + (id & 7);
      bx = (id >> 3) % gx;
    }
  }
  const int m0 = by * BM;
  const int n0 = bx * BN;
  const int batch = blockIdx.z / g.split_k;
  const int slice = blockIdx.z % g.split_k;

  const unsigned short* A = g.A + (int64_t)batch * g.sa;
  const unsigned short* B = g.B + (int64_t)batch * g.sb;

  const int nk_total = (g.K + BK - 1) / BK;
  const int nk_per = (nk_total + g.split_k - 1) / g.split_k;
  const int kt_begin = slice * nk_per;
  const int kt_end = min(nk_total, kt_begin + nk_per);

  constexpr int FM = BM / 32, FN = BN / 32;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int CA = BM * (BK / 8) / 256;   // 16-byte chunks per thread for A
  constexpr int CB = BN * (BK / 8) / 256;
  uint4 ra[CA], rb[CB];

  auto load_tiles = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int c = tid + i * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (!TN) {
        const int row = c / (BK / 8), kc = c % (BK / 8);
        const int m = m0 + row, k = k0 + kc * 8;
        if (m < g.M && k < g.K) v = *reinterpret_cast<const uint4*>(A + (int64_t)m * g.lda + k);
      } else {
        const int kr = c / (BM / 8), mc = c % (BM / 8);
        const int m = m0 + mc * 8;
        int k = k0 + kr;
        bool ok = (k < g.K) && (m < g.M);
        if (g.period > 0) {
          const int t = k % g.period + g.a_shift;
          ok = ok && (t >= 0) && (t < g.period);
          k += g.a_shift;
        }
        if (ok) v = *reinterpret_cast<const uint4*>(A + (int64_t)k * g.lda + m);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int c = tid + i * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (!TN) {
        const int row = c / (BK / 8), kc = c % (BK / 8);
        const int n = n0 + row, k = k0 + kc * 8;
        if (n < g.N && k < g.K) v = *reinterpret_cast<const uint4*>(B + (int64_t)n * g.ldb + k);
      } else {
        const int kr = c / (BN / 8), nc = c % (BN / 8);
        const int n = n0 + nc * 8, k = k0 + kr;
        if (k < g.K && n < g.N) v = *reinterpret_cast<const uint4*>(B + (int64_t)k * g.ldb + n);
      }
      rb[i] = v;
    }
  };

  auto store_tiles = [&](int buf) {
    unsigned short* as = As + buf * BM * LDK;
    unsigned short* bs = Bs + buf * BN * LDK;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int c = tid + i * 256;
      if (!TN) {
        const int row = c / (BK / 8), kc = c % (BK / 8);
        *reinterpret_cast<uint4*>(as + row * LDK + kc * 8) = ra[i];
      } else {
        const int kr = c / (BM / 8), mc = c % (BM / 8);
        const unsigned short* e = reinterpret_cast<const unsigned short*>(&ra[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) as[(mc * 8 + j) * LDK + kr] = e[j];
      }
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int c = tid + i * 256;
      if (!TN) {
        const int row = c / (BK / 8), kc = c % (BK / 8);
        *reinterpret_cast<uint4*>(bs + row * LDK + kc * 8) = rb[i];
      } else {
        const int kr = c / (BN / 8), nc = c % (BN / 8);
        const unsigned short* e = reinterpret_cast<const unsigned short*>(&rb[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) bs[(nc * 8 + j) * LDK + kr] = e[j];
      }
    }
  };

  if (kt_begin < kt_end) {
    load_tiles(kt_begin);
    store_tiles(0);
    __syncthreads();
    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const bool more = (kt + 1 < kt_end);
      if (more) load_tiles(kt + 1);
      const unsigned short* as = As + buf * BM * LDK + (wr * (BM / 2)) * LDK;
      const unsigned short* bs = Bs + buf * BN * LDK + (wc * (BN / 2)) * LDK;
#pragma unroll
      for (int kk = 0; kk < BK; kk += 32) {
        bf16x8 af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
          af[i] = *reinterpret_cast<const bf16x8*>(as + (i * 16 + (lane & 15)) * LDK + kk + 8 * (lane >> 4));
#pragma unroll
        for (int j = 0; j < FN; ++j)
          bfr[j] = *reinterpret_cast<const bf16x8*>(bs + (j * 16 + (lane & 15)) * LDK + kk + 8 * (lane >> 4));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
      if (more) store_tiles(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + reg
  float* Cf = reinterpret_cast<float*>(g.C) + (int64_t)batch * g.sc;
  unsigned short* Cb = reinterpret_cast<unsigned short*>(g.C) + (int64_t)batch * g.sc;
  if (!TN && !g.atomic && g.c_perm_h == 0) {
    // Row-contiguous stores: each wave parks its (BM/2) x (BN/2) tile in LDS (the operand buffers are free now) and
    // writes it back 16 bytes per lane, 16 lanes per row.  The products of this path are output-bound (fp32 xproj:
    // 419 MB per layer), and four-byte stores scattered over four rows per instruction waste most of the write path.
    constexpr int WM = BM / 2, WN = BN / 2, LDC = WN + 4;
    constexpr int EH = (KB < 64 && FM >= 2) ? 2 : 1;      // row halves of the wave's tile staged at a time (LDS budget)
    constexpr int HM = WM / EH;
    __syncthreads();
    float* cs = reinterpret_cast<float*>(smem) + wave * HM * LDC;
#pragma unroll
    for (int eh = 0; eh < EH; ++eh) {
#pragma unroll
    for (int i = 0; i < FM / EH; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[(i * 16 + (lane >> 4) * 4 + r) * LDC + j * 16 + (lane & 15)] = acc[eh * (FM / EH) + i][j][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): a wave only reads back what it wrote itself
    constexpr int LPR = WN / 4;                // lanes per row
    constexpr int RPI = 64 / LPR;              // rows per instruction
    const int cl = (lane % LPR) * 4, rl = lane / LPR;
    const int col = n0 + wc * WN + cl;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias != nullptr && slice == 0)
#pragma unroll
      for (int e = 0; e < 4; ++e) if (col + e < g.N) bv[e] = g.bias[col + e];
#pragma unroll 4
    for (int r0 = 0; r0 < HM; r0 += RPI) {
      const int row = m0 + wr * WM + eh * HM + r0 + rl;
      if (row >= g.M || col >= g.N) continue;
      const float4 v4 = *reinterpret_cast<const float4*>(cs + (r0 + rl) * LDC + cl);
      float v[4] = {v4.x + bv[0], v4.y + bv[1], v4.z + bv[2], v4.w + bv[3]};
      if (g.drop_keep < 1.0f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= drop_scale(g, row, col + e);
      }
      const int64_t off = (int64_t)row * g.ldc + col;
      const bool full = (col + 3 < g.N);
      if (g.out_bf16) {
        if (full && ((off & 3) == 0)) {
          uint2 pk;
          pk.x = (unsigned)las_f2bf(v[0]) | ((unsigned)las_f2bf(v[1]) << 16);
          pk.y = (unsigned)las_f2bf(v[2]) | ((unsigned)las_f2bf(v[3]) << 16);
          *reinterpret_cast<uint2*>(Cb + off) = pk;
        } else {
          for (int e = 0; e < 4; ++e) if (col + e < g.N) Cb[off + e] = las_f2bf(v[e]);
        }
      } else if (full && ((off & 3) == 0)) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (g.accumulate) {
          const float4 c = *reinterpret_cast<const float4*>(Cf + off);
          o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
        }
        *reinterpret_cast<float4*>(Cf + off) = o;
      } else {
        for (int e = 0; e < 4; ++e)
          if (col + e < g.N) { if (g.accumulate) Cf[off + e] += v[e]; else Cf[off + e] = v[e]; }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);       // the staging rows are rewritten by the next half
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int coln = n0 + wc * (BN / 2) + j * 16 + (lane & 15);
      if (coln >= g.N) continue;
      const int col = g.c_perm_h > 0 ? (coln & 3) * g.c_perm_h + (coln >> 2) : coln;
      const float bv = (g.bias != nullptr && slice == 0) ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
        if (row >= g.M) continue;
        float v = acc[i][j][r] + bv;
        if (g.drop_keep < 1.0f) v *= drop_scale(g, row, col);
        const int64_t off = (int64_t)row * g.ldc + col;
        if (g.out_bf16) {
          Cb[off] = las_f2bf(v);
        } else if (g.atomic) {
          atomicAdd(Cf + off, v);
        } else if (g.accumulate) {
          Cf[off] += v;
        } else {
          Cf[off] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TN product with hardware-transposed LDS reads (gfx950 ds_read_b64_tr_b16, cdna_hip_programming.md T10).
// The operand tiles are staged exactly as they lie in memory, [k][m] and [k][n] with 256-byte rows (16-byte
// vector stores, no transposition on the way in), in the XOR image of T10 (b):
//     off(row, chunk) = 256*row + 16*(chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)))
// and every MFMA fragment (8 consecutive k of one m or n) is two transposed reads of a 4-row x 16-column block:
// lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3; lane i receives column i of the four rows.
// 128 x 128 output tile, BK = 64, 4 waves in a 2 x 2 grid, register-staged double buffering.
// ------------------------------------------------------------------------------------------------
constexpr int TBK = 32;     // K depth of one staged tile of the TN kernel
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ int tr_off(int row, int chunk) { return 256 * row + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__global__ __launch_bounds__(256) void gemm_tn_tr_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                       // [2][TBK][256 B]
  unsigned char* Bs = smem + 2 * TBK * 256;        // [2][TBK][256 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, and all output tiles of one K slice read the same
  // A and B rows.  Linear id -> (XCD = id % 8, j = id / 8); the j-th workgroup of an XCD takes tile j % P of slice
  // (j / P) * 8 + XCD, so a slice's operands are pulled into ONE L2 once and shared by its P tiles (instead of every XCD
  // fetching every slice).  The last gz % 8 slices keep the plain order.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const int gx = gridDim.x, gy = gridDim.y, P = gx * gy;
    const int id = bx + gx * (by + gy * bz);
    if (id < (int)(gridDim.z / 8) * 8 * P) {
      const int j = id >> 3, t = j % P;
      bz = (j / P) * 8 + (id & 7);
      by = t / gx;
      bx = t - by * gx;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int batch = bz / g.split_k, slice = bz % g.split_k;
  const unsigned short* A = g.A + (int64_t)batch * g.sa;
  const unsigned short* B = g.B + (int64_t)batch * g.sb;
  const int nk_total = (g.K + TBK - 1) / TBK;
  const int nk_per = (nk_total + g.split_k - 1) / g.split_k;
  const int kt_begin = slice * nk_per;
  const int kt_end = min(nk_total, kt_begin + nk_per);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int CH = TBK * 16 / 256;          // 16-byte chunks per thread and operand
  uint4 ra[CH], rb[CH];
  // A thread always stages the same 8-column chunk (256 % 16 == 0), rows kr0 + 16 i of every tile: which source its A
  // columns come from, the row pointers and the position inside the period are loop-carried instead of being
  // recomputed (the integer work per tile was costing more issue slots than the MFMAs).
  const int ch = tid & 15, kr0 = tid >> 4;
  int kindA = 0;                  // 0: zero, 1: rows as they lie, 2: rows shifted inside the period, 3: the ones row
  const unsigned short* pa = A;
  int64_t lda_e = 0;
  int shift = 0;
  {
    const int m = m0 + ch * 8;
    if (g.M1 >= 0) {              // fused LSTM weight gradient: [x | y shifted | ones]
      if (m < g.M1) { kindA = 1; pa = A + m; lda_e = g.lda; }
      else if (m < g.M1 + g.M2) { kindA = 2; pa = g.A2 + (m - g.M1); lda_e = g.lda2; shift = g.a_shift; }
      else if (m == g.M1 + g.M2 && g.bias_row) kindA = 3;
    } else if (m < g.M) {
      kindA = g.period > 0 ? 2 : 1;
      pa = A + m;
      lda_e = g.lda;
      shift = g.period > 0 ? g.a_shift : 0;
    }
  }
  const bool n_ok = (n0 + ch * 8) < g.N;
  int kcur = kt_begin * TBK + kr0;                       // row of chunk 0 in the tile staged next
  int t0 = g.period > 0 ? kcur % g.period : 0;
  const unsigned short* paK = pa + ((int64_t)kcur + shift) * lda_e;
  const unsigned short* pbK = B + (int64_t)kcur * g.ldb + (n0 + ch * 8);
  auto load_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
      if (kcur + 16 * i < g.K) {
        if (kindA == 1) {
          va = *reinterpret_cast<const uint4*>(paK + (int64_t)(16 * i) * lda_e);
        } else if (kindA == 2) {
          int t = t0 + 16 * i;
          while (t >= g.period) t -= g.period;
          t += shift;
          if (t >= 0 && t < g.period) va = *reinterpret_cast<const uint4*>(paK + (int64_t)(16 * i) * lda_e);
        } else if (kindA == 3) {
          va.x = 0x3F80u;                                 // bf16 1.0 in the first of the eight rows
        }
        if (n_ok) vb = *reinterpret_cast<const uint4*>(pbK + (int64_t)(16 * i) * g.ldb);
      }
      ra[i] = va;
      rb[i] = vb;
    }
    kcur += TBK;
    paK += (int64_t)TBK * lda_e;
    pbK += (int64_t)TBK * g.ldb;
    if (g.period > 0) {
      t0 += TBK;
      while (t0 >= g.period) t0 -= g.period;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = tid + i * 256;
      const int off = tr_off(c >> 4, c & 15);
      *reinterpret_cast<uint4*>(As + buf * TBK * 256 + off) = ra[i];
      *reinterpret_cast<uint4*>(Bs + buf * TBK * 256 + off) = rb[i];
    }
  };
  // transposed-read addressing of this lane: group kg = lane>>4 takes k rows 8kg..8kg+7 of each 32-deep step
  const int kg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  auto frag = [&](const unsigned char* tile, int kk, int c0) -> bf16x8 {
    const int r = kk + 8 * kg + q;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + tr_off(r, c0 + (p >> 1)) + 8 * (p & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + tr_off(r + 4, c0 + (p >> 1)) + 8 * (p & 1)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  if (kt_begin < kt_end) {
    load_tiles();
    store_tiles(0);
    __syncthreads();
    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const bool more = (kt + 1 < kt_end);
      if (more) load_tiles();
      const unsigned char* as = As + buf * TBK * 256;
      const unsigned char* bs = Bs + buf * TBK * 256;
#pragma unroll
      for (int kk = 0; kk < TBK; kk += 32) {
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = frag(as, kk, wr * 8 + 2 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag(bs, kk, wc * 8 + 2 * j);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
      if (more) store_tiles(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  if (g.partial) {
    // memory-side fp32 atomics cost far more than the product itself at these shapes (~16k per workgroup): every
    // slice stores its tile plainly (empty slices store zeros) and tn_reduce_kernel sums them
    float* P = g.partial + (int64_t)bz * g.partial_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int coln = n0 + wc * 64 + j * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wr * 64 + i * 16 + (lane >> 4) * 4 + r;
          if (row < g.M && coln < g.N) P[(int64_t)row * g.N + coln] = acc[i][j][r];
        }
      }
    }
    return;
  }
  float* Cf = reinterpret_cast<float*>(g.C) + (int64_t)batch * g.sc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int coln = n0 + wc * 64 + j * 16 + (lane & 15);
      if (coln >= g.N) continue;
      const int col = g.c_perm_h > 0 ? (coln & 3) * g.c_perm_h + (coln >> 2) : coln;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (row >= g.M) continue;
        if (g.bias_row && row == g.M - 1) { atomicAdd(g.bias_row + col, acc[i][j][r]); continue; }
        const int64_t off = (int64_t)row * g.ldc + col;
        if (g.atomic) atomicAdd(Cf + off, acc[i][j][r]);
        else if (g.accumulate) Cf[off] += acc[i][j][r];
        else if (g.out_bf16) (reinterpret_cast<unsigned short*>(g.C) + (int64_t)batch * g.sc)[off] = las_f2bf(acc[i][j][r]);
        else Cf[off] = acc[i][j][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The fused LSTM weight-gradient product on the LDS-DMA ring (same structure as gemm_nt_ring_kernel): 128 x 256 output
// tile, 8 waves as 2 x 4 with a 64 x 64 block = 2 x 2 tiles of v_mfma_f32_32x32x16_bf16 each, 32-deep stages of
// [k][128 m] (A) and two [k][128 n] sub-images (B) as they lie in memory -- 256-byte rows in the XOR image tr_off() -- filled
// by global_load_lds_dwordx4 with the chunk permutation on the SOURCE address, read back with ds_read_b64_tr_b16.  Rows
// that must read as zero (outside the period of the shifted y rows, beyond K, the padding columns) and the ones row fetch
// from a 32-byte constant block instead.  Slices store their tiles into the caller's workspace (tn_reduce_kernel sums them).
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(32))) unsigned short las_const_rows[16] = {0x3F80, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
constexpr int TR_BK = 32;
// NBI = 128-column sub-images of B per stage: 2 = a 128 x 256 output tile, waves as 2 x 4 with 64 x 64 each (round 2); 4 = a
// 128 x 512 tile, waves as 1 x 8 with 128 x 64 each (round 5): per MFMA a third fewer LDS-DMA pieces (5 per wave and stage for 16
// MFMAs against 3 for 8) and a quarter fewer fragment reads -- the instruction stream of a wave, not the LDS or the matrix pipe,
// is what the 64 x 64 form was paced by (an LDS-DMA piece costs its wave 60-180 issue cycles, MI355X_MICROARCH.md).
template <int NBI> struct TnRing {
  static constexpr int STAGES = NBI == 2 ? 4 : 3;
  static constexpr int STAGE_BYTES = TR_BK * 256 * (1 + NBI);        // A 8 KiB + NBI x 8 KiB
  static constexpr int BM = 128, BN = 128 * NBI;
  static constexpr int TM = NBI == 2 ? 2 : 4, TN = 2;                 // 32 x 32 tiles per wave
};

template <int NBI>
__global__ __launch_bounds__(512) void gemm_tn_ring_kernel(GemmArgs g) {
  typedef TnRing<NBI> R;
  constexpr int BM = R::BM, BN = R::BN, TR_STAGES = R::STAGES, TR_STAGE_BYTES = R::STAGE_BYTES, TM = R::TM, TN = R::TN;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* lds = (lds_u8*)smem;
  const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = NBI == 2 ? wave >> 2 : 0, wn = NBI == 2 ? wave & 3 : wave;       // this wave's 64-row (128-row) block and 64-column block
  // XCD-aware order as in gemm_tn_tr_kernel: the tiles of one K slice go to one XCD
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const int gx = gridDim.x, gy = gridDim.y, P = gx * gy;
    const int id = bx + gx * (by + gy * bz);
    if (id < (int)(gridDim.z / 8) * 8 * P) {
      const int j = id >> 3, t = j % P;
      bz = (j / P) * 8 + (id & 7);
      by = t / gx;
      bx = t - by * gx;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int slice = bz;
  const int nk_total = (g.K + TR_BK - 1) / TR_BK;
  const int nk_per = (nk_total + g.split_k - 1) / g.split_k;
  const int kt_begin = slice * nk_per;
  const int nk = max(0, min(nk_total, kt_begin + nk_per) - kt_begin);

  // this lane's 1 + NBI loads per stage: rows 4 wave + (lane >> 4) of the A image and of the B sub-images; LDS slot
  // (lane & 15) of a row holds source chunk slot ^ swz(row)
  const int srow = 4 * wave + (lane >> 4);
  const int chunk = (lane & 15) ^ (((srow & 3) << 2) | ((srow >> 2) & 3));
  const unsigned short* zeros = las_const_rows + 8;
  int kindA = 0;
  const unsigned short* pa = zeros;
  int64_t lda_e = 0;
  int shift = 0;
  {
    const int m = m0 + chunk * 8;
    if (m < g.M1) { kindA = 1; pa = g.A + m; lda_e = g.lda; }
    else if (m < g.M1 + g.M2) { kindA = 2; pa = g.A2 + (m - g.M1); lda_e = g.lda2; shift = g.a_shift; }
    else if (m == g.M1 + g.M2 && g.bias_row) kindA = 3;
  }
  int kcur = kt_begin * TR_BK + srow;
  int t0 = kcur % g.period;
  const unsigned short* paK = pa + ((int64_t)kcur + shift) * lda_e;
  const unsigned short* pbK[NBI];
  bool b_ok[NBI];
#pragma unroll
  for (int j = 0; j < NBI; ++j) {
    const int n = n0 + j * 128 + chunk * 8;
    b_ok[j] = n < g.N;
    pbK[j] = g.B + (int64_t)kcur * g.ldb + (b_ok[j] ? n : 0);
  }
  auto issue = [&](int stage) {
    const unsigned base = lds_base + stage * TR_STAGE_BYTES + wave * 1024;
    const bool k_ok = kcur < g.K;
    const unsigned short* sa = zeros;
    if (k_ok) {
      if (kindA == 1) sa = paK;
      else if (kindA == 2) { const int t = t0 + shift; if (t >= 0 && t < g.period) sa = paK; }
      else if (kindA == 3) sa = las_const_rows;
    }
    glds16(sa, base);
#pragma unroll
    for (int j = 0; j < NBI; ++j) {
      glds16((k_ok && b_ok[j]) ? pbK[j] : zeros, base + (1 + j) * TR_BK * 256);
      pbK[j] += (int64_t)TR_BK * g.ldb;
    }
    kcur += TR_BK;
    paK += (int64_t)TR_BK * lda_e;
    t0 += TR_BK;
    while (t0 >= g.period) t0 -= g.period;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed fragment reads: lane group (lane >> 4): k half = lane >> 5, 16-column block = (lane >> 4) & 1 of the 32 rows /
  // columns of an MFMA tile; inside a group lane 4 q + p reads row q, columns 4 p .. 4 p + 3 and receives column (lane & 15)
  const int kh = lane >> 5, blk = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
  auto frag = [&](const lds_u8* img, int kk, int c32) -> bf16x8 {        // c32: 32-column tile index inside the 128-column image
    const int r = kk + 8 * kh + q;
    const int c0 = c32 * 4 + blk * 2 + (p >> 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + tr_off(r, c0) + 8 * (p & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + tr_off(r + 4, c0) + 8 * (p & 1)));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  if (nk > 0) {
#pragma unroll
    for (int s0 = 0; s0 < TR_STAGES - 1; ++s0)
      if (s0 < nk) issue(s0);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + TR_STAGES - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((1 + NBI) * (TR_STAGES - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // (TNR_NO_GLDS / TNR_NO_DSREAD / TNR_NO_MFMA: ablation builds for scripts/gpu_tn_ablate.py -- which of the three the loop waits for)
      const lds_u8* as = lds + stage * TR_STAGE_BYTES;
      const lds_u8* bs = as + TR_BK * 256 + (wn >> 1) * TR_BK * 256;       // the 128-column sub-image of this wave's columns
#pragma unroll
      for (int kk = 0; kk < TR_BK; kk += 16) {
        bf16x8 af[TM], bfr[TN];
#ifndef TNR_NO_DSREAD
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = frag(as, kk, wm * 2 + i);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = frag(bs, kk, (wn & 1) * 2 + j);
#else
#pragma unroll
        for (int i = 0; i < TM; ++i) { af[i] = __builtin_bit_cast(bf16x8, make_uint4(kt + i, lane, kk, 3)); asm volatile("" : "+v"(af[i])); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { bfr[j] = __builtin_bit_cast(bf16x8, make_uint4(kt + j, lane, kk, 5)); asm volatile("" : "+v"(bfr[j])); }
#endif
#ifndef TNR_NO_MFMA
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
#else
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(af[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(bfr[j]));
#endif
#if !defined(TNR_NO_GLDS)
        // the LDS-DMA of stage kt + STAGES - 1 is issued BEHIND the first half's products, not in front of the stage's fragment
        // reads (round 6): the pieces' issue time -- 60-180 cycles each -- then runs in the shadow of eight MFMAs.  Alone 538 -> 556
        // TFLOP/s on the metric-L shapes, nothing in the step (profiles/r06_tn_ring_ablation.txt: the same file has what the loop
        // waits for -- without the MFMAs it takes 64 % of its time, without the LDS-DMA 69 %, without the fragment reads 95 %).
        if (kk == 0) {
          __builtin_amdgcn_sched_barrier(0);
          if (kt + TR_STAGES - 1 < nk) issue(stage == 0 ? TR_STAGES - 1 : stage - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
      }
      stage = (stage + 1 == TR_STAGES) ? 0 : stage + 1;
    }
  }
  // every slice stores its tile (empty slices: zeros); C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* P = g.partial + (int64_t)bz * g.partial_stride;
  const int l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int coln = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < g.M && coln < g.N) P[(int64_t)row * g.N + coln] = acc[i][j][r];
      }
    }
}

// C[row, perm(col)] += sum over slices of partial[slice][row][col] (last row -> bias_row when given); N % 4 == 0.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ P, int64_t stride, int nsl, float* C, int64_t ldc,
                                                        float* bias_row, int M, int N, int perm_h) {
  const int n4 = N >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * n4) return;
  const int row = (int)(idx / n4), c4 = (int)(idx - (int64_t)row * n4);
  const float* src = P + (int64_t)row * N + 4 * c4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int sl = 0; sl < nsl; ++sl) {
    const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)sl * stride);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float e[4] = {s.x, s.y, s.z, s.w};
  float* dst = (bias_row && row == M - 1) ? bias_row : C + (int64_t)row * ldc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int coln = 4 * c4 + i;
    dst[perm_h > 0 ? (coln & 3) * perm_h + (coln >> 2) : coln] += e[i];
  }
}

// ------------------------------------------------------------------------------------------------
// NT product for the bulk shapes (x K_x of layers >= 1, dX, keys; K a multiple of 64): 256 x 128 output tile,
// 512 threads = 8 waves in a 4 x 2 grid, each wave a 64 x 64 block as 2 x 2 tiles of v_mfma_f32_32x32x16_bf16 (half the
// LDS fragment bytes per flop of the 16x16x32 form: 16 KB of fragments per wave and 64-deep K tile against 1024 MFMA
// cycles per SIMD, so the LDS read path runs at about half its peak instead of at it).
// Operand tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no registers) into
// a ring of three 48-KiB stages; a wave waits only for ITS six loads of the tile it is about to read (counted vmcnt: the
// six of the next tile stay in flight across the raw s_barrier) and issues the loads of tile kt+2 right after the
// barrier that proves every wave has left tile kt-1 (whose stage they overwrite): one barrier per K tile.
// The LDS image of a tile is lane-linear per load ([row][8 chunks of 16 B], 128-B rows, eight rows per KiB); bank
// conflicts are avoided by permuting the SOURCE chunk a lane fetches: LDS slot s of row r holds chunk s ^ ((r >> 1) & 7),
// so the 32 rows one ds_read_b128 lane group touches fall on 16 distinct (row parity, slot) = 4-bank groups.
// ------------------------------------------------------------------------------------------------
// BM x BN output tile, BK-deep stages, STAGES of them; 8 waves as WGM x (8 / WGM)
// STREAM (round 4, las_gemm_nt_stream): the product feeds a recurrence that runs BESIDE it.  A 256-row tile is then not 256
// consecutive rows of A but the R utterances of ONE slice of the recurrence (one chain group) x 256 / R steps of that
// group's direction: row i = utterance slice * R + i / (256/R) at step tau = sb * (256/R) + i % (256/R) of ITS sequence,
// i.e. time t = tau (d = 0) or length - 1 - tau (d = 1, the reversed recurrence); steps beyond an utterance's length are not
// stored (nobody reads them).  The kernel is PERSISTENT and XCD-aware: a workgroup reads its XCC id, collects the chain
// groups that run on the same XCD (the recurrence publishes where its groups sit) and takes their tiles from that XCD's
// queue, step block by step block -- so a tile is written into the L2 its consumer reads from: plain stores, an L2-level
// counter (column tiles finished of (group, step block)) and plain loads are coherent there, with no device-wide release
// (a buffer_wbl2 per tile walks the whole L2 -- measured: it cost the recurrence beside it 25 % -- and write-through stores
// without it are NOT ordered against the counter: tiles arrived torn).  Groups whose members are spread over XCDs (never seen
// under an otherwise idle dispatcher, but placement is not a contract) are dealt to the XCDs round-robin and take the
// write-back fence.
// NW (round 4): waves per workgroup.  With 8 waves on a 256 x 256 tile (128 x 64 per wave) a 32-deep stage costs every wave 12
// fragment reads of 1 KiB for 16 MFMAs: 96 KiB of LDS reads + 32 KiB of LDS-DMA writes per stage and CU = 1 024 cycles of the
// 128 B/clk LDS port against 1 024 cycles of MFMA per SIMD -- the LDS port is as busy as the matrix cores, and the kernel sat
// at half the clock-derated roof.  NW = 4 gives every wave a 128 x 128 block (256 accumulator registers, one wave per SIMD):
// 16 reads for 32 MFMAs, 64 + 32 KiB per stage = 768 LDS cycles against the same 1 024 MFMA cycles.
template <int BM, int BN, int BK, int STAGES, int WGM, bool STREAM = false, int NW = 8, bool PP = false>
__global__ __launch_bounds__(NW * 64) void gemm_nt_ring_kernel(GemmArgs g) {
  constexpr int WGN = NW / WGM;
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;          // 32 x 32 tiles per wave
  constexpr int ROWB = BK * 2;                                   // bytes per tile row
  constexpr int CPR = BK / 8;                                    // 16-byte chunks per row
  constexpr int RPC = 1024 / ROWB;                               // rows per 1-KiB load
  constexpr int NA = BM / RPC / NW, NB = BN / RPC / NW;          // loads per wave and stage
  constexpr int NL = NA + NB;
  constexpr int STAGE_BYTES = (BM + BN) * ROWB;
  constexpr int KS = BK / 16;                                    // MFMA k-steps per stage
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* lds = (lds_u8*)smem;
  const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)lds);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  // ---- streamed form: this XCD's list of chain groups, then tiles from its queue until the list is done ----
  __shared__ int s_list[STREAM ? 256 : 1];
  __shared__ int s_n, s_q;
  int my_xcd = 0;
  if constexpr (STREAM) {
    static_assert(BM == 256, "streamed tiles are one slice x 256 / R steps");
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    my_xcd = (int)(xcc & 7);
    if (tid == 0) {
      // This XCD's list of chain groups: the groups that PUBLISHED this XCD (workgroup i of a launch does not always land on XCD
      // i % 8: the dispatcher carries on from where the previous launch stopped).  Every workgroup of an XCD must derive the SAME
      // list (they share one queue index: ADVICE r4), so the launch takes ONE decision, in word 8 of `ready`:
      //   1 = every group has published (the words never change afterwards): the XCD-local lists;
      //   2 = some workgroup waited STREAM_HANDSHAKE_POLLS (a few milliseconds) in vain: every workgroup deals the groups to the
      //       XCDs round-robin and produces them in the SPREAD form (device-wide release per tile), whether or not the recurrence
      //       has started.  Round 6: a recurrence that starts late -- queued behind other work, or short of CUs
      //       BECAUSE the producers that wait for it hold them -- used to find the producers gone (they left after 2^15 polls),
      //       burn its own bounded wait and have the optimiser step withheld; now the product is simply there when it arrives
      //       (tests/test_gpu_lstm.py::test_streamed_input_product_whose_chain_starts_late).
      // The first workgroup to know decides (compare-and-swap); the others follow it.
      constexpr unsigned STREAM_HANDSHAKE_POLLS = 1u << 12;
      unsigned* mode_w = g.ready + 8;
      const int ngroups = g.s_nslices * g.s_ndir;
      unsigned spins = 0, mode = 0;
      for (int gi = 0; gi < ngroups && mode == 0; ++gi)
        while (__hip_atomic_load(g.ready + 16 + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          mode = __hip_atomic_load(mode_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (mode != 0) break;
          if (++spins > STREAM_HANDSHAKE_POLLS) { atomicCAS(mode_w, 0u, 2u); break; }
          __builtin_amdgcn_s_sleep(8);
        }
      if (mode == 0) {
        if (spins <= STREAM_HANDSHAKE_POLLS) atomicCAS(mode_w, 0u, 1u);          // saw them all
        mode = __hip_atomic_load(mode_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      int n = 0;
      for (int gi = 0; gi < ngroups && gi < 256; ++gi) {
        if (mode == 1) {
          const unsigned v = __hip_atomic_load(g.ready + 16 + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v == (unsigned)my_xcd + 1u) s_list[n++] = gi;
          else if (v == LAS_STREAM_SPREAD && (gi & 7) == my_xcd) s_list[n++] = gi | 0x10000;
        } else if ((gi & 7) == my_xcd) {
          s_list[n++] = gi | 0x10000;
        }
      }
      s_n = n;
    }
    __syncthreads();
  }
  for (;;) {
  int bx = blockIdx.x, by = blockIdx.y;
  int s_sb = 0, s_d = 0, s_slice = 0, s_group = 0;
  bool s_spread = false;
  if constexpr (STREAM) {
    const int nc_tiles = g.N / g.s_ndir / BN;
    __syncthreads();                              // the previous tile's epilogue has left the LDS; s_q may be rewritten
    if (tid == 0) s_q = (int)__hip_atomic_fetch_add(g.ready + my_xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int q = s_q, per_sb = s_n * nc_tiles;
    if (per_sb == 0 || q >= per_sb * g.s_nsb) break;
    // step block by step block; inside a block the column tiles of one group are neighbours (they share the A rows)
    s_sb = q / per_sb;
    const int r = q % per_sb, e = s_list[r / nc_tiles];
    s_group = e & 0xffff;
    s_spread = (e >> 16) != 0;
    s_d = s_group / g.s_nslices;
    s_slice = s_group % g.s_nslices;
    bx = s_d * nc_tiles + r % nc_tiles;
    by = 0;
  } else {
    const int gx = gridDim.x, gy = gridDim.y;
    const int id = bx + gx * by;
    const int full = (gy / 8) * 8 * gx;
    if (id < full) {
      by = (id / (8 * gx)) * 8 + (id & 7);
      bx = (id >> 3) % gx;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int batch = STREAM ? 0 : blockIdx.z;
  // streamed tiles: global row of tile row i, or -1 when that (utterance, step) does not exist
  auto stream_row = [&](int i) -> int {
    const int sbs = 256 / g.sR;                   // steps per block (R = 4: 64)
    const int b = s_slice * g.sR + i / sbs, tau = s_sb * sbs + i % sbs;
    if (b >= g.sB) return -1;
    const int len = min(g.length[b], g.sT);
    if (tau >= len) return -1;
    return b * g.sT + (s_d == 0 ? tau : len - 1 - tau);
  };
  const unsigned short* A = g.A + (int64_t)batch * g.sa + (STREAM ? (int64_t)s_d * g.s_astride : 0);
  const unsigned short* B = g.B + (int64_t)batch * g.sb;
  const int nk = g.K / BK;

  // row swizzle: LDS slot s of row r holds source chunk s ^ f(r); f spreads the rows one ds_read_b128 lane group touches
  // ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of a 32-row fragment) over distinct 4-bank groups of the 256-byte bank row
  auto fswz = [](int r) { return CPR == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
  // this lane's source rows: chunk q = wave + 8 i covers rows RPC q .. RPC q + RPC - 1; rows past the edge re-read the last row
  const unsigned short* src[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const bool isA = i < NA;
    const int q = wave + NW * (isA ? i : i - NA);
    const int row = RPC * q + lane / CPR;
    const int ch = (lane % CPR) ^ fswz(row);
    if (isA) {
      if constexpr (STREAM) src[i] = A + (int64_t)max(stream_row(row), 0) * g.lda + 8 * ch;       // (absent rows read row 0: never stored)
      else src[i] = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + 8 * ch;
    } else src[i] = B + (int64_t)min(n0 + row, g.N - 1) * g.ldb + 8 * ch;
  }
  auto issue = [&](int stage) {
    const unsigned base = lds_base + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = wave + NW * (i < NA ? i : i - NA);
      glds16(src[i], base + (i < NA ? 0 : BM * ROWB) + q * 1024);
      src[i] += BK;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: lane reads (row = l & 31, chunk 2 ks + (l >> 5)) at slot chunk ^ f(row)
  const int l31 = lane & 31, hk = lane >> 5, swz = fswz(l31);
  const int a_row = (wm * (BM / WGM) + l31) * ROWB, b_row = BM * ROWB + (wn * (BN / WGN) + l31) * ROWB;
  int koff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) koff[ks] = ((2 * ks + hk) ^ swz) << 4;

  if constexpr (PP) {
    // PING-PONG form (round 5; MI355X_MICROARCH.md "Two waves per SIMD", cdna_hip_programming.md "The 256^2 8-phase template").  The
    // two waves of a SIMD are wave w and wave w + 4.  In the form below both run the same program between the same barriers: both
    // want the LDS at the same time, then both want the matrix pipe at the same time, and each waits while the other's phase runs.
    // Here a stage is TWO segments -- LOAD (all of the stage's fragments into registers, the LDS-DMA of a later stage issued, the
    // waits) and MATH (the stage's 16 MFMAs, nothing else) -- with a barrier after each, and waves 4..7 run ONE BARRIER BEHIND waves
    // 0..3: while one wave of a SIMD is in MATH its partner is in LOAD.  Hazards, with segments numbered s (group 0: LOAD(k) at 2k,
    // MATH(k) at 2k + 1; group 1 one later): a wave waits for ITS pieces of stage k + 1 and for its own fragment reads at the end
    // of LOAD(k), i.e. in front of a barrier that precedes every read of stage k + 1 by either group (RAW) and every LDS-DMA into
    // the slot LOAD(k) has just read (issued in LOAD(k + 1) at the earliest: WAR).
    static_assert(!STREAM && NW == 8, "ping-pong form: 8 waves, plain tiles");
    const int grp = wave >> 2;
    if (nk > 0) {
#pragma unroll
      for (int p = 0; p < STAGES - 1; ++p)
        if (p < nk) issue(p);
      if (STAGES - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL * (STAGES - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // stage 0 has landed
      if (grp == 1) __builtin_amdgcn_s_barrier();          // the stagger
      asm volatile("" ::: "memory");
      int stage = 0;
      for (int kt = 0; kt < nk; ++kt) {
        // ---- LOAD(kt) ----
        const lds_u8* st = lds + stage * STAGE_BYTES;
        bf16x8 af[KS][TM], bfr[KS][TN];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
          for (int j = 0; j < TN; ++j) bfr[ks][j] = *(const __attribute__((address_space(3))) bf16x8*)(st + b_row + j * 32 * ROWB + koff[ks]);
#pragma unroll
          for (int i = 0; i < TM; ++i) af[ks][i] = *(const __attribute__((address_space(3))) bf16x8*)(st + a_row + i * 32 * ROWB + koff[ks]);
        }
        if (kt + STAGES - 1 < nk) issue(stage == 0 ? STAGES - 1 : stage - 1);        // = (kt + STAGES - 1) % STAGES: last read in LOAD(kt - 1)
        if (kt + 1 + STAGES - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL * (STAGES - 2)) : "memory");      // own pieces of stage kt + 1
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           // own fragments are in registers
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MATH(kt) ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stage = (stage + 1 == STAGES) ? 0 : stage + 1;
      }
      if (grp == 0) __builtin_amdgcn_s_barrier();          // the barrier the other half is one ahead by
    }
  } else
  if (nk > 0) {
#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p)
      if (p < nk) issue(p);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // own loads of tile kt have landed: at most the STAGES - 2 younger tiles stay in flight (the tail drains fully)
      if (kt + STAGES - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL * (STAGES - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // everybody's have landed; everybody has left tile kt - 1
      asm volatile("" ::: "memory");
#ifndef RING_NO_GLDS
      if (kt + STAGES - 1 < nk) issue(stage == 0 ? STAGES - 1 : stage - 1);        // = (kt + STAGES - 1) % STAGES
#endif
      const lds_u8* st = lds + stage * STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 af[TM], bfr[TN];
#ifndef RING_NO_DSREAD
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const __attribute__((address_space(3))) bf16x8*)(st + a_row + i * 32 * ROWB + koff[ks]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *(const __attribute__((address_space(3))) bf16x8*)(st + b_row + j * 32 * ROWB + koff[ks]);
#else
#pragma unroll
        for (int i = 0; i < TM; ++i) { af[i] = __builtin_bit_cast(bf16x8, make_uint4(kt + i, lane, ks, 3)); asm volatile("" : "+v"(af[i])); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { bfr[j] = __builtin_bit_cast(bf16x8, make_uint4(kt + j, lane, ks, 5)); asm volatile("" : "+v"(bfr[j])); }
#endif
#ifndef RING_NO_MFMA
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
#else
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(af[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(bfr[j]));
#endif
      }
      stage = (stage + 1 == STAGES) ? 0 : stage + 1;
    }
  }

  // epilogue: each wave parks 32 rows x (32 TN) columns of its tile at a time in LDS (row stride 32 TN + 4 floats) and
  // writes the rows back 16 bytes per lane: row-contiguous stores (these products are bound by their fp32 output)
  __builtin_amdgcn_s_barrier();              // every wave has finished reading the last stage
  asm volatile("" ::: "memory");
  constexpr int WN_ = 32 * TN, LDC = WN_ + 4;
  static_assert(NW * 32 * LDC * 4 <= STAGES * STAGE_BYTES, "epilogue staging must fit in the ring");
  float* cs = reinterpret_cast<float*>(smem) + wave * 32 * LDC;
  float* Cf = reinterpret_cast<float*>(g.C) + (int64_t)batch * g.sc;
  unsigned short* Cb = reinterpret_cast<unsigned short*>(g.C) + (int64_t)batch * g.sc;
  constexpr int LPR = WN_ / 4, RPI = 64 / LPR;        // lanes per row, rows per store instruction
  const int cl = (lane % LPR) * 4, rl = lane / LPR;
  const int col = n0 + wn * WN_ + cl;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias != nullptr)
#pragma unroll
    for (int e = 0; e < 4; ++e) if (col + e < g.N) bv[e] = g.bias[col + e];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        cs[((r & 3) + 8 * (r >> 2) + 4 * hk) * LDC + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): a wave reads back only what it wrote itself
#pragma unroll 4
    for (int r0 = 0; r0 < 32; r0 += RPI) {
      int row = m0 + wm * (BM / WGM) + i * 32 + r0 + rl;
      if constexpr (STREAM) {
        row = stream_row(row);
        if (row < 0) continue;
      }
      if (row >= g.M || col >= g.N) continue;
      const float4 v4 = *reinterpret_cast<const float4*>(cs + (r0 + rl) * LDC + cl);
      float v[4] = {v4.x + bv[0], v4.y + bv[1], v4.z + bv[2], v4.w + bv[3]};
      if (g.drop_keep < 1.0f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= drop_scale(g, row, col + e);
      }
      const int64_t off = (int64_t)row * g.ldc + col;
      const bool full = (col + 3 < g.N);
      if (g.out_bf16) {
        if (full && ((off & 3) == 0)) {
          uint2 pk;
          pk.x = (unsigned)las_f2bf(v[0]) | ((unsigned)las_f2bf(v[1]) << 16);
          pk.y = (unsigned)las_f2bf(v[2]) | ((unsigned)las_f2bf(v[3]) << 16);
          *reinterpret_cast<uint2*>(Cb + off) = pk;
        } else {
          for (int e = 0; e < 4; ++e) if (col + e < g.N) Cb[off + e] = las_f2bf(v[e]);
        }
      } else if (full && ((off & 3) == 0)) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (g.accumulate) {
          const float4 c = *reinterpret_cast<const float4*>(Cf + off);
          o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
        }
        *reinterpret_cast<float4*>(Cf + off) = o;
      } else {
        for (int e = 0; e < 4; ++e)
          if (col + e < g.N) { if (g.accumulate) Cf[off + e] += v[e]; else Cf[off + e] = v[e]; }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);        // the staging rows are rewritten by the next 32 rows
  }
  if constexpr (STREAM) {
    // every wave's stores have been acknowledged by this XCD's L2 (vmcnt(0)) before the barrier; the consumer reads through the
    // same L2, so the counter (an L2 atomic) may follow at once.  Spread groups: device-wide release first (L2 write-back).
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) {
      if (s_spread) __threadfence();
      __hip_atomic_fetch_add(g.ready + las_stream_flags_offset(g.s_nslices * g.s_ndir) + s_group * g.s_nsb + s_sb, 1u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    break;
  }
  }   // for (;;): the streamed form takes the next tile of its XCD's queue
}


template <int BM, int BN, int BK, int STAGES, int WGM, int NW = 8, bool PP = false>
int launch_ring(const GemmArgs& g, int batch, hipStream_t st) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, batch);
  const size_t lds = (size_t)STAGES * (BM + BN) * BK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_ring_kernel<BM, BN, BK, STAGES, WGM, false, NW, PP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_ring_kernel<BM, BN, BK, STAGES, WGM, false, NW, PP>), grid, dim3(NW * 64), lds, st, g);
  LAS_LAUNCH_CHECK("ring gemm launch");
  return LAS_OK;
}

template <int BM, int BN, bool TN, int KB = 64>
int launch(const GemmArgs& g, int batch, hipStream_t st) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, batch * g.split_k);
  size_t lds = (size_t)2 * (BM + BN) * (KB + 8) * sizeof(unsigned short);
  const size_t stage = (size_t)4 * (BM / 2 / (KB < 64 && BM >= 64 ? 2 : 1)) * (BN / 2 + 4) * sizeof(float);   // epilogue staging
  if (!TN && stage > lds) lds = stage;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, TN, KB>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<BM, BN, TN, KB>), grid, dim3(256), lds, st, g);
  LAS_LAUNCH_CHECK("gemm launch");
  return LAS_OK;
}

// ------------------------------------------------------------------------------------------------
// Skinny NT product for the per-step decoder GEMMs (M <= 64 rows = the batch): one workgroup owns a
// 64 x 16 output block over the WHOLE K; its 4 waves take interleaved 32-deep K chunks, load their MFMA
// fragments straight from global memory (16 B per lane, no LDS staging, every load independent) and the
// four partial accumulators meet in LDS.  N/16 workgroups, no split-K atomics, no barrier in the K loop.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_nt_skinny_kernel(GemmArgs g) {
  __shared__ float part[4][64][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nk = (g.K + 31) / 32;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n = n0 + l15;
  const bool n_ok = n < g.N;
  const unsigned short* brow = g.B + (int64_t)(n_ok ? n : 0) * g.ldb + 8 * lq;
  const unsigned short* arow[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = i * 16 + l15;
    a_ok[i] = m < g.M;
    arow[i] = g.A + (int64_t)(a_ok[i] ? m : 0) * g.lda + 8 * lq;
  }
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (int kc = wave; kc < nk; kc += 4) {
    const int k = kc * 32;
    const bool k_ok = (k + 8 * lq) < g.K;
    uint4 bv = (n_ok && k_ok) ? *reinterpret_cast<const uint4*>(brow + k) : zero;
    uint4 av[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = (a_ok[i] && k_ok) ? *reinterpret_cast<const uint4*>(arow[i] + k) : zero;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[i]), __builtin_bit_cast(bf16x8, bv),
                                                       acc[i], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][i * 16 + lq * 4 + r][l15] = acc[i][r];
  __syncthreads();
  float* C = reinterpret_cast<float*>(g.C);
  for (int e = tid; e < 64 * 16; e += 256) {
    const int row = e / 16, col = e % 16;
    if (row < g.M && n0 + col < g.N) {
      float v = part[0][row][col] + part[1][row][col] + part[2][row][col] + part[3][row][col];
      if (g.bias) v += g.bias[n0 + col];
      const int64_t off = (int64_t)row * g.ldc + n0 + col;
      if (g.accumulate) C[off] += v; else C[off] = v;
    }
  }
}

__global__ void zero_rows_kernel(float* C, int64_t ldc, int M, int N, int64_t sc) {
  float* c = C + (int64_t)blockIdx.z * sc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)M * N;
       i += (int64_t)gridDim.x * blockDim.x) {
    c[(i / N) * ldc + (i % N)] = 0.f;
  }
}

}  // namespace

static int gemm_nt(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, void* C, int64_t ldc,
                   const float* bias, int M, int N, int K, int out_bf16, int accumulate, int batch,
                   int64_t sa, int64_t sb, int64_t sc, int split_k, float drop_keep, unsigned drop_seed, unsigned drop_stream,
                   void* stream) {
  LAS_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "las_gemm_nt: empty problem M=%d N=%d K=%d batch=%d", M, N, K, batch);
  LAS_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "las_gemm_nt: K, lda, ldb must be multiples of 8 (K=%d lda=%ld ldb=%ld)", K, (long)lda, (long)ldb);
  LAS_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && sa % 8 == 0 && sb % 8 == 0, "las_gemm_nt: operands must be 16-byte aligned");
  if (split_k < 1) split_k = 1;
  LAS_REQUIRE(!(out_bf16 && (accumulate || split_k > 1)), "las_gemm_nt: bf16 output cannot accumulate or split K");
  hipStream_t st = (hipStream_t)stream;
  GemmArgs g{A, B, C, bias, lda, ldb, ldc, sa, sb, sc, M, N, K, out_bf16, accumulate, split_k > 1 ? 1 : 0, split_k, 0, 0, 0};
  g.drop_keep = drop_keep;
  g.drop_seed = drop_seed;
  g.drop_stream = drop_stream;
  LAS_REQUIRE(drop_keep >= 1.0f || (split_k == 1 && batch == 1 && M > 64 && N > 64 && !out_bf16),
              "las_gemm_nt_masked: one fp32 product of bulk shape (M, N > 64), K not split");
  if (split_k > 1 && !accumulate) {
    hipLaunchKernelGGL(zero_rows_kernel, dim3(64, 1, batch), dim3(256), 0, st, (float*)C, ldc, M, N, sc);
    LAS_LAUNCH_CHECK("gemm zero");
  }
  if (M <= 64 && !out_bf16 && batch == 1 && split_k == 1 && K >= 256) {
    hipLaunchKernelGGL(gemm_nt_skinny_kernel, dim3((N + 15) / 16), dim3(256), 0, st, g);
    LAS_LAUNCH_CHECK("skinny gemm launch");
    return LAS_OK;
  }
  if (M <= 64 || N <= 64) return launch<64, 64, false>(g, batch, st);
  {
    // The bulk products run on the LDS-DMA ring kernel.  Measured on the metric-M shapes (scripts/gpu_nt_ab.py, us):
    //   K = 2048 (dX): 256 x 256 tiles, 32-deep x 4 stages (one workgroup per CU)      140 / 154  (register-staged: 188 / 203)
    //   K <= 1024 (x K_x with its fp32 output, keys): 256 x 128 tiles, 32-deep x 3 stages = 72 KiB, two workgroups
    //   per CU, so one's epilogue runs beside the other's K loop                          215 / 155 / 27  (251 / 199 / 30)
    // LAS_GEMM_RING=0: register-staged kernels everywhere; 1: 256x128x64 (3 stages); 2: 256x256 wherever N allows; 4: the
    // two-per-CU form everywhere (diagnostics, A/B timing)
    const int ring = las_knob("LAS_GEMM_RING", 5);
    if (ring && split_k == 1 && K % 64 == 0 && K >= 128 && M >= 1024 && N >= 128 && (ldc % 4 == 0 || out_bf16)) {
      const bool wide = N >= 256 && N % 256 != 128;
      if (ring == 1) return launch_ring<256, 128, 64, 3, 4>(g, batch, st);
      if (ring == 4 || (ring >= 5 && !(wide && K > 1024))) return launch_ring<256, 128, 32, 3, 4>(g, batch, st);
      // LAS_GEMM_PP (round 5): the ping-pong schedule of the 256 x 256 ring (the two waves of a SIMD alternate LOAD and MATH
      // segments: +3-6 % on the K > 1024 shapes); 0: off.  (On the two-per-CU 256 x 128 form it changed nothing -- two workgroups
      // de-phase each other already -- and that instantiation was removed in round 6.)
      const int pp = las_knob("LAS_GEMM_PP", 1);
      if (wide) return pp ? launch_ring<256, 256, 32, 4, 2, 8, true>(g, batch, st) : launch_ring<256, 256, 32, 4, 2>(g, batch, st);
      return launch_ring<256, 128, 64, 3, 4>(g, batch, st);
    }
  }
  if (M >= 4096) return launch<128, 128, false, 32>(g, batch, st);      // the bulk products: three workgroups per CU
  return launch<128, 128, false>(g, batch, st);
}

static int gemm_tn(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, void* C, int64_t ldc,
                   int M, int N, int K, int a_shift, int period, int c_perm_h, int batch, int64_t sa,
                   int64_t sb, int64_t sc, int split_k, bool store, bool out_bf16, void* stream) {
  LAS_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "las_gemm_tn: empty problem");
  LAS_REQUIRE(!store || split_k <= 1, "las_gemm_tn_store: K cannot be split");
  LAS_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= ((M + 7) / 8) * 8 && ldb >= ((N + 7) / 8) * 8,
              "las_gemm_tn: lda/ldb must be multiples of 8 covering M/N rounded up to 8");
  LAS_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && sa % 8 == 0 && sb % 8 == 0, "las_gemm_tn: operands must be 16-byte aligned");
  if (split_k < 1) split_k = 1;
  hipStream_t st = (hipStream_t)stream;
  LAS_REQUIRE(c_perm_h == 0 || N == 4 * c_perm_h, "las_gemm_tn: c_perm_h needs N == 4*H");
  GemmArgs g{A, B, C, nullptr, lda, ldb, ldc, sa, sb, sc, M, N, K, out_bf16 ? 1 : 0, store ? 0 : 1, store ? 0 : 1, split_k, a_shift, period, c_perm_h};
  if (M <= 64 || N <= 64) return launch<64, 64, true>(g, batch, st);
  {
    dim3 grid((N + 127) / 128, (M + 127) / 128, batch * split_k);
    const size_t lds = (size_t)4 * TBK * 256;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_tr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL(gemm_tn_tr_kernel, grid, dim3(256), lds, st, g);
    LAS_LAUNCH_CHECK("gemm tn launch");
    return LAS_OK;
  }
}

// ------------------------------------------------------------------------------------------------
// NT product whose B operand is a WEIGHT matrix handed in as its LAS_IMAGE_PACK_MFMA_B image (round 6).  The ring
// kernels above stage both operands through the LDS and sit where its port is as busy as the matrix cores (12 fragment reads for
// 16 MFMAs of 32x32x16 per wave and stage).  Here only the ACTIVATION operand goes through the LDS ring (LDS-DMA, 16 KB per
// 32-deep stage); a wave fetches its B fragments -- four contiguous KB per stage, the same for the waves above and below it -- from
// the image straight into registers, two stages ahead.  256 x 256 tiles, 8 waves as 2 x 4, a wave owns 128 x 64 as 8 x 4 tiles
// of v_mfma_f32_16x16x32_bf16: 8 LDS fragment reads + 4 global fragment loads for 32 MFMAs per stage.
// Every load of the K loop is inline assembly (LDS-DMA and the fragment loads share the wave's in-order vmcnt queue and the
// compiler knows of neither): every stage issues [B(kt + 3) x 4, A(kt + 3) x 2] and waits vmcnt(12) = everything up to A(kt + 1).
// K a multiple of 128, N of 256.
// ------------------------------------------------------------------------------------------------
namespace {
// TI: 16-row tiles per wave; the workgroup's tile is 32 TI rows x 256 columns.  The host picks the height that fills the last round
// of workgroups best (800 tiles of 256 rows on 256 CUs are four rounds, the last one an eighth full: 0.91 PFLOP/s for a kernel that
// runs 1.17 while every CU has a tile; 1280 tiles of 160 rows are five full rounds of five eighths of the work each).
template <int STAGES, int TI>
__global__ __launch_bounds__(512) void gemm_nt_bimg_kernel(GemmArgs g) {
  static_assert(STAGES == 4, "the LDS slot of a stage is its fragment slot");
  static_assert(TI >= 6 && TI <= 8, "a wave's rows: 96 .. 128");
  constexpr int BM = 32 * TI, BN = 256, BK = 32, ROWB = BK * 2, STAGE_BYTES = BM * ROWB;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* lds = (lds_u8*)smem;
  const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    const int id = bx + gx * by;
    const int full = (gy / 8) * 8 * gx;
    if (id < full) {
      by = (id / (8 * gx)) * 8 + (id & 7);
      bx = (id >> 3) % gx;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int nk = g.K / BK, KC = nk;
  // LDS slot s of row r holds source chunk s ^ f(r), f(r) = (-(r >> 2)) & 3: the sixteen 16-byte pieces one ds_read_b128 lane group
  // touches for a 16x16x32 A fragment (rows {0-3, 12-15} of one k piece with rows {4-11} of the next) then fall on distinct banks
  auto fswz = [](int r) { return (-(r >> 2)) & 3; };
  // A: this lane's two source rows per stage (1-KB LDS-DMA loads: 16 rows x 4 chunks; 2 TI pieces; wave w loads pieces w and w + 8,
  // past the last piece the last piece once more -- every wave has the same two loads per stage in its queue)
  const unsigned short* asrc[2];
  int apiece[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    apiece[i] = min(wave + 8 * i, 2 * TI - 1);
    const int row = 16 * apiece[i] + (lane >> 2);
    asrc[i] = g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda + 8 * ((lane & 3) ^ fswz(row));
  }
  // B: fragment (column tile nt, k chunk kc) of the image = 512 elements at (nt * KC + kc) * 512; this lane's 8 at + lane * 8
  const unsigned short* bsrc = g.B + ((int64_t)((n0 + wn * 64) / 16) * KC) * 512 + lane * 8;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));       // (a native vector: inline assembly ties it to registers)
  u32x4 bq[4][4];
  auto issue_b = [&](int slot, int kc) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(bq[slot][j]) : "v"(bsrc + ((int64_t)j * KC + kc) * 512) : "memory");
  };
  auto issue_a = [&](int stage, bool advance) {
    const unsigned base = lds_base + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      glds16(advance ? asrc[i] : asrc[i] - BK, base + apiece[i] * 1024);      // (past the end: the last stage again)
      if (advance) asrc[i] += BK;
    }
  };
  f32x4 acc[TI][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, lq = lane >> 4;
  const int a_off = (wm * 16 * TI + l15) * ROWB + ((lq ^ fswz(l15)) << 4);       // (+ i * 16 rows: f is the same for rows 16 apart)

  // The K loop has NO branch but its back edge: every stage issues the loads of the stage three on -- past the end the LAST stage
  // again, into a slot nobody reads any more -- so the queue always holds the same 12 loads behind the stage being waited for
  // and no register of an in-flight fragment ever meets a control-flow merge (a first version guarded the issues: the compiler
  // rotated the fragment slots through v_mov copies at the merges, of registers whose loads had not landed -- wrong sums).
  // nk a multiple of 4 and >= 4 (the slots are compile-time registers: the loop is written out four stages at a time).
#pragma unroll
  for (int p = 0; p < 3; ++p) { issue_b(p, p); issue_a(p, true); }
  {
    // PING-PONG form (as gemm_nt_ring_kernel's): a stage is a LOAD segment -- the stage's A fragments out of the LDS, the loads of
    // the stage three on issued, the wait for the NEXT stage's pieces (its weight fragments are then in their registers, its A
    // pieces in the LDS) -- and a MATH segment of 4 TI MFMAs, a barrier behind each; waves 4..7 run one barrier behind waves 0..3,
    // so that of the two waves of a SIMD one is in MATH while the other is in LOAD.
    const int grp = wave >> 2;
    asm volatile("s_waitcnt vmcnt(12)" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]) : : "memory");      // stage 0
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();            // the stagger
    asm volatile("" ::: "memory");
    auto pp_body = [&](int kt, auto slot_tag) {
      constexpr int SLOT = decltype(slot_tag)::value, NEXT = (SLOT + 1) % 4;
      // ---- LOAD(kt) ----
      const lds_u8* st = lds + SLOT * STAGE_BYTES;
      bf16x8 af[TI];
#pragma unroll
      for (int i = 0; i < TI; ++i) af[i] = *(const __attribute__((address_space(3))) bf16x8*)(st + a_off + i * 16 * ROWB);
      const bool more = kt + 3 < nk;
      issue_b((SLOT + 3) % 4, more ? kt + 3 : nk - 1);
      issue_a((SLOT + 3) % 4, more);                       // (its slot was last read in LOAD(kt - 1), a barrier ago for either group)
      asm volatile("s_waitcnt vmcnt(12)" : "+v"(bq[NEXT][0]), "+v"(bq[NEXT][1]), "+v"(bq[NEXT][2]), "+v"(bq[NEXT][3]) : : "memory");   // own pieces of stage kt + 1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own fragments are in registers
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- MATH(kt) ----
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 bf = __builtin_bit_cast(bf16x8, bq[SLOT][j]);
#pragma unroll
        for (int i = 0; i < TI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt < nk; kt += 4) {
      pp_body(kt, std::integral_constant<int, 0>{});
      pp_body(kt + 1, std::integral_constant<int, 1>{});
      pp_body(kt + 2, std::integral_constant<int, 2>{});
      pp_body(kt + 3, std::integral_constant<int, 3>{});
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();            // the barrier the other half is one ahead by
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the repeated loads of the last stage)

  // epilogue: a wave parks 16 rows x 64 columns of its block at a time in LDS and writes them back 16 bytes per lane
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  constexpr int LDC = 64 + 4;
  float* cs = reinterpret_cast<float*>(smem) + wave * 16 * LDC;
  float* Cf = reinterpret_cast<float*>(g.C);
  const int cl = (lane & 15) * 4, rl = lane >> 4;
  const int col = n0 + wn * 64 + cl;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias != nullptr)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = g.bias[col + e];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[(lq * 4 + r) * LDC + j * 16 + l15] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): a wave reads back only what it wrote itself
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = m0 + wm * 16 * TI + i * 16 + p * 4 + rl;
      if (row >= g.M) continue;
      const float4 v4 = *reinterpret_cast<const float4*>(cs + (p * 4 + rl) * LDC + cl);
      float4 o = make_float4(v4.x + bv[0], v4.y + bv[1], v4.z + bv[2], v4.w + bv[3]);
      float* dst = Cf + (int64_t)row * g.ldc + col;
      if (g.accumulate) {
        const float4 c = *reinterpret_cast<const float4*>(dst);
        o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
      }
      // non-temporal: 0.2-0.8 GB of fp32 that nobody on this XCD reads again (the next reader is a recurrence, rows at a time, through
      // its own L2); x K_x of metric-L's layer 1 500 -> 477 us, metric-L 16.21 -> 16.12 ms
      typedef float f4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(f4{o.x, o.y, o.z, o.w}, reinterpret_cast<f4*>(dst));
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
}

// the LAS_IMAGE_PACK_MFMA_B image of a bf16 matrix [N, K] (row stride ldb): the micro-benchmark's way to the image (the model's
// weights are packed from their fp32 masters by las_refresh_images)
__global__ __launch_bounds__(256) void pack_b_bf16_kernel(const unsigned short* B, int64_t ldb, int N, int K, unsigned short* img) {
  const int KC = K / 32;
  const int64_t total = (int64_t)N * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int e = (int)(i & 7), lane = (int)(i >> 3) & 63;
    const int64_t frag = i >> 9;
    const int n = (int)(frag / KC) * 16 + (lane & 15), k = (int)(frag % KC) * 32 + (lane >> 4) * 8 + e;
    img[i] = B[(int64_t)n * ldb + k];
  }
}
}  // namespace

extern "C" int las_pack_mfma_b_bf16(const las_bf16* B, int64_t ldb, int N, int K, las_bf16* image, void* stream) {
  LAS_REQUIRE(B && image && N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "las_pack_mfma_b_bf16: N a multiple of 16, K of 32");
  hipLaunchKernelGGL(pack_b_bf16_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, B, ldb, N, K, image);
  LAS_LAUNCH_CHECK("pack image launch");
  return LAS_OK;
}

extern "C" int las_gemm_nt_bimg(const las_bf16* A, int64_t lda, const las_bf16* b_image, float* C, int64_t ldc, const float* bias, int M,
                                int N, int K, int accumulate, void* stream) {
  LAS_REQUIRE(A && b_image && C && M > 0, "las_gemm_nt_bimg: null argument or empty problem");
  LAS_REQUIRE(N > 0 && N % 256 == 0 && K >= 128 && K % 128 == 0 && lda % 8 == 0 && ldc % 4 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)b_image % 16 == 0),
              "las_gemm_nt_bimg: N a multiple of 256, K of 128 (N = %d, K = %d), lda of 8, ldc of 4, operands 16-byte aligned", N, K);
  GemmArgs g{A, b_image, C, bias, lda, 0, ldc, 0, 0, 0, M, N, K, 0, accumulate, 0, 1, 0, 0, 0};
  // tile height 32 TI, TI = 6 .. 8: the one whose rounds of workgroups (one per CU) cost least -- rounds x height x what a lower
  // tile loses per flop (the weight fragments are amortised over fewer rows; measured: 160-row tiles give back all that five
  // full rounds instead of four save, 224-row tiles run dZ K_x^T of metric-M's layer 2 at 1.0 PFLOP/s against 0.89)
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  int ti = las_knob("LAS_BIMG_TI", 0);
  if (ti < 6 || ti > 8) {
    long best = -1;
    for (int t = 8; t >= 6; --t) {
      const long tiles = (long)((M + 32 * t - 1) / (32 * t)) * (N / 256);
      const long cost = ((tiles + cus - 1) / cus) * t * (t == 8 ? 100 : t == 7 ? 105 : 112);
      if (best < 0 || cost < best) { best = cost; ti = t; }
    }
  }
  const dim3 grid(N / 256, (M + 32 * ti - 1) / (32 * ti), 1);
  auto go = [&](auto ti_tag) {
    constexpr int TI = decltype(ti_tag)::value;
    const size_t lds = (size_t)4 * 32 * TI * 64;      // (the ring; the epilogue's staging, 34 KB, fits inside)
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bimg_kernel<4, TI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_nt_bimg_kernel<4, TI>), grid, dim3(512), lds, (hipStream_t)stream, g);
  };
  switch (ti) {
    case 6: go(std::integral_constant<int, 6>{}); break;
    case 7: go(std::integral_constant<int, 7>{}); break;
    default: go(std::integral_constant<int, 8>{}); break;
  }
  LAS_LAUNCH_CHECK("image gemm launch");
  return LAS_OK;
}

extern "C" int las_gemm_nt(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, void* C, int64_t ldc,
                           const float* bias, int M, int N, int K, int out_bf16, int accumulate, int batch,
                           int64_t sa, int64_t sb, int64_t sc, int split_k, void* stream) {
  return gemm_nt(A, lda, B, ldb, C, ldc, bias, M, N, K, out_bf16, accumulate, batch, sa, sb, sc, split_k, 1.0f, 0, 0, stream);
}

extern "C" int las_gemm_nt_stream_supported(int N, int K, int ndir) {
  return (ndir == 1 || ndir == 2) && N > 0 && (N / ndir) % 128 == 0 && N % ndir == 0 && K >= 128 && K % 64 == 0;
}

extern "C" size_t las_gemm_nt_stream_flags(int B, int T, int ndir, int rows_per_slice) {
  if (rows_per_slice != 4 && rows_per_slice != 8 && rows_per_slice != 16) return 0;
  const int nslices = (B + rows_per_slice - 1) / rows_per_slice, ngroups = nslices * ndir, sbs = 256 / rows_per_slice;
  return (size_t)las_stream_flags_offset(ngroups) + (size_t)ngroups * ((T + sbs - 1) / sbs);
}

extern "C" int las_gemm_nt_stream_dirs(const las_bf16* A, int64_t lda, int64_t a_dir_stride, const las_bf16* Bm, int64_t ldb, float* C,
                                       int64_t ldc, const float* bias, const int32_t* length, int B, int T, int N, int K, int ndir,
                                       int rows_per_slice, uint32_t* ready, void* stream) {
  LAS_REQUIRE(a_dir_stride % 8 == 0, "las_gemm_nt_stream_dirs: the directions' operands 16-byte aligned (a_dir_stride a multiple of 8)");
  LAS_REQUIRE(A && Bm && C && length && ready && B > 0 && T > 0, "las_gemm_nt_stream: null argument or empty batch");
  LAS_REQUIRE(las_gemm_nt_stream_supported(N, K, ndir), "las_gemm_nt_stream: N = %d (per direction a multiple of 128), K = %d (multiple of 64, >= 128), ndir = %d", N, K, ndir);
  LAS_REQUIRE(rows_per_slice == 4 || rows_per_slice == 8 || rows_per_slice == 16, "las_gemm_nt_stream: rows_per_slice = the recurrence's slice height (4, 8, 16)");
  LAS_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)Bm % 16 == 0),
              "las_gemm_nt_stream: operands must be 16-byte aligned, lda / ldb multiples of 8, ldc of 4");
  GemmArgs g{A, Bm, C, bias, lda, ldb, ldc, 0, 0, 0, B * T, N, K, 0, 0, 0, 1, 0, 0, 0};
  g.length = length;
  g.ready = ready;
  g.sB = B; g.sT = T; g.s_ndir = ndir; g.sR = rows_per_slice;
  g.s_astride = a_dir_stride;
  g.s_nslices = (B + rows_per_slice - 1) / rows_per_slice;
  g.s_nsb = (T + 256 / rows_per_slice - 1) / (256 / rows_per_slice);
  LAS_REQUIRE(g.s_nslices * ndir <= 256, "las_gemm_nt_stream: at most 256 chain groups");
  constexpr int BM = 256, BN = 128, BK = 32, STAGES = 3, WGM = 4;
  const size_t lds = (size_t)STAGES * (BM + BN) * BK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_ring_kernel<BM, BN, BK, STAGES, WGM, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  // persistent: two workgroups per CU would fit; whatever is resident on an XCD works through that XCD's queue, the rest of the
  // grid finds the queues empty and leaves
  // LAS_STREAM_GRID: the producer's workgroups.  Measured in one call (metric-M): 96: 6.09 ms, 128: 6.02, 192: 6.02, 256: 6.04, 512: 6.05
  int sgrid = las_knob("LAS_STREAM_GRID", 192);
  if (sgrid < 8 || sgrid > 1024) sgrid = 192;
  hipLaunchKernelGGL((gemm_nt_ring_kernel<BM, BN, BK, STAGES, WGM, true>), dim3(sgrid), dim3(512), lds, (hipStream_t)stream, g);
  LAS_LAUNCH_CHECK("streamed gemm launch");
  return LAS_OK;
}

extern "C" int las_gemm_nt_masked(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, float* C, int64_t ldc,
                                  int M, int N, int K, int accumulate, float keep, uint32_t seed, uint32_t stream_id,
                                  void* stream) {
  LAS_REQUIRE(keep > 0.f && keep <= 1.f && ldc == N, "las_gemm_nt_masked: keep in (0, 1], C contiguous (ldc == N: the mask is indexed row * N + col)");
  return gemm_nt(A, lda, B, ldb, C, ldc, nullptr, M, N, K, 0, accumulate, 1, 0, 0, 0, 1, keep, seed, stream_id, stream);
}

extern "C" int las_gemm_tn(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, float* C, int64_t ldc,
                           int M, int N, int K, int a_shift, int period, int c_perm_h, int batch, int64_t sa,
                           int64_t sb, int64_t sc, int split_k, void* stream) {
  return gemm_tn(A, lda, B, ldb, C, ldc, M, N, K, a_shift, period, c_perm_h, batch, sa, sb, sc, split_k, false, false, stream);
}

extern "C" size_t las_gemm_tn_ws_bytes(int M, int N, int split_k) {
  return (split_k > 1 && M > 0 && N > 0 && N % 4 == 0) ? sizeof(float) * (size_t)split_k * (size_t)M * (size_t)N : 0;
}

// C += A^T B with a result that does not depend on scheduling (bit-identical from run to run): the K slices store their
// tiles in the caller's workspace and tn_reduce_kernel adds them to C in slice order -- no fp32 atomics.  Shapes the
// slice kernel is not built for (N not a multiple of 4) or a workspace that is too small run unsplit: one workgroup per output
// tile, one contributor per element.  (Narrow products -- M or N <= 64: the token rows of the cell kernel, the projection
// layer -- take the 128 x 128 slice kernel too when their caller splits K: unsplit on the 64-row kernel they ran 16
// workgroups over K = B*U and took 140 us.)
extern "C" int las_gemm_tn_ws(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, float* C, int64_t ldc,
                              int M, int N, int K, int a_shift, int period, int c_perm_h, int split_k, float* workspace,
                              size_t workspace_bytes, void* stream) {
  const size_t need = las_gemm_tn_ws_bytes(M, N, split_k);
  if (need == 0 || workspace == nullptr || workspace_bytes < need)
    return gemm_tn(A, lda, B, ldb, C, ldc, M, N, K, a_shift, period, c_perm_h, 1, 0, 0, 0, 1, false, false, stream);
  LAS_REQUIRE(M > 0 && N > 0 && K > 0, "las_gemm_tn_ws: empty problem");
  LAS_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= ((M + 7) / 8) * 8 && ldb >= ((N + 7) / 8) * 8,
              "las_gemm_tn_ws: lda/ldb must be multiples of 8 covering M/N rounded up to 8");
  LAS_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0), "las_gemm_tn_ws: operands must be 16-byte aligned");
  LAS_REQUIRE(c_perm_h == 0 || N == 4 * c_perm_h, "las_gemm_tn_ws: c_perm_h needs N == 4*H");
  GemmArgs g{A, B, C, nullptr, lda, ldb, ldc, 0, 0, 0, M, N, K, 0, 1, 1, split_k, a_shift, period, c_perm_h};
  g.partial = workspace;
  g.partial_stride = (int64_t)M * N;
  dim3 grid((N + 127) / 128, (M + 127) / 128, split_k);
  const size_t lds = (size_t)4 * TBK * 256;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_tr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_tn_tr_kernel, grid, dim3(256), lds, (hipStream_t)stream, g);
  LAS_LAUNCH_CHECK("gemm tn (workspace) launch");
  const int64_t n = (int64_t)M * (N / 4);
  hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace,
                     g.partial_stride, split_k, C, ldc, (float*)nullptr, M, N, c_perm_h);
  LAS_LAUNCH_CHECK("gemm tn (workspace) reduce launch");
  return LAS_OK;
}

extern "C" int las_gemm_tn_store(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, void* C, int64_t ldc,
                                 int M, int N, int K, int a_shift, int period, int c_perm_h, int batch, int64_t sa,
                                 int64_t sb, int64_t sc, int out_bf16, void* stream) {
  LAS_REQUIRE(!(out_bf16 && c_perm_h), "las_gemm_tn_store: no column permutation with bf16 output");
  return gemm_tn(A, lda, B, ldb, C, ldc, M, N, K, a_shift, period, c_perm_h, batch, sa, sb, sc, 1, true, out_bf16 != 0, stream);
}

extern "C" size_t las_gemm_tn_lstm_workspace_bytes(int D, int H, int split_k) {
  return split_k > 1 ? sizeof(float) * (size_t)split_k * (size_t)(D + H + 1) * (size_t)(4 * H) : 0;
}

extern "C" int las_gemm_tn_lstm(const las_bf16* x, int64_t ldx, int D, const las_bf16* y, int64_t ldy, int H, int a_shift,
                                int period, const las_bf16* dz, int64_t ldz, float* kernel_grad, float* bias_grad, int K,
                                int split_k, float* workspace, void* stream) {
  LAS_REQUIRE(D >= 0 && H > 0 && K > 0 && period > 0 && D % 8 == 0 && H % 8 == 0, "las_gemm_tn_lstm: bad shape D=%d H=%d K=%d", D, H, K);
  LAS_REQUIRE((D == 0 || (x && ldx % 8 == 0 && ((uintptr_t)x % 16 == 0))) && y && dz && kernel_grad && bias_grad && ldy % 8 == 0 &&
                  ldz % 8 == 0 && ((uintptr_t)y % 16 == 0) && ((uintptr_t)dz % 16 == 0),
              "las_gemm_tn_lstm: operands must be 16-byte aligned with strides that are multiples of 8");
  // the tile shape travels with the call (round 6, ADVICE r5: it used to be a process-global knob the Python layer rewrote before
  // every launch): LAS_TN_SPLIT_WIDE in split_k asks for 128 x 512 output tiles
  const bool want_wide = (split_k & LAS_TN_SPLIT_WIDE) != 0;
  split_k &= LAS_TN_SPLIT_WIDE - 1;
  if (split_k < 1) split_k = 1;
  GemmArgs g{x, dz, kernel_grad, nullptr, ldx, ldz, 4 * (int64_t)H, 0, 0, 0, D + H + 1, 4 * H, K, 0, 1, 1, split_k, a_shift, period, H};
  g.A2 = y;
  g.lda2 = ldy;
  g.M1 = D;
  g.M2 = H;
  g.bias_row = bias_grad;
  if (workspace && split_k > 1) {
    g.partial = workspace;
    g.partial_stride = (int64_t)g.M * g.N;
  }
  const int ring = las_knob("LAS_TN_RING", 1);          // 0: the register-staged 128 x 128 kernel (diagnostics, A/B timing)
  if (ring && g.partial) {
    // 128 x 512 output tiles (round 5; 128 x 64 per wave) where the caller asks for them and N is a multiple of 512;
    // LAS_TN_WIDE=0 (A/B override): 128 x 256 everywhere
    const bool wide = want_wide && las_knob("LAS_TN_WIDE", 1) != 0 && g.N % 512 == 0;
    static bool attr_ring = false;
    if (!attr_ring) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_ring_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                TnRing<2>::STAGES * TnRing<2>::STAGE_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_ring_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                TnRing<4>::STAGES * TnRing<4>::STAGE_BYTES);
      attr_ring = true;
    }
    // (round 5: the ping-pong schedule of gemm_nt_ring_kernel was tried on both forms and measured SLOWER -- 64 x 64 per wave: a MATH
    // segment of eight MFMAs is shorter than the LOAD segment beside it, 179 -> 226 us alone; 128 x 64 per wave: 155 -> 168 us alone,
    // 5.57 -> 6.10 ms in the metric-L step; not kept)
    if (wide) {
      dim3 grid((g.N + 511) / 512, (g.M + 127) / 128, split_k);
      hipLaunchKernelGGL(gemm_tn_ring_kernel<4>, grid, dim3(512), (size_t)TnRing<4>::STAGES * TnRing<4>::STAGE_BYTES, (hipStream_t)stream, g);
    } else {
      dim3 grid((g.N + 255) / 256, (g.M + 127) / 128, split_k);
      hipLaunchKernelGGL(gemm_tn_ring_kernel<2>, grid, dim3(512), (size_t)TnRing<2>::STAGES * TnRing<2>::STAGE_BYTES, (hipStream_t)stream, g);
    }
  } else {
  dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, split_k);
  const size_t lds = (size_t)4 * TBK * 256;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_tr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_tn_tr_kernel, grid, dim3(256), lds, (hipStream_t)stream, g);
  }
  LAS_LAUNCH_CHECK("lstm weight-gradient gemm launch");
  if (g.partial) {
    const int64_t n = (int64_t)g.M * (g.N / 4);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g.partial,
                       g.partial_stride, split_k, kernel_grad, g.ldc, bias_grad, g.M, g.N, H);
    LAS_LAUNCH_CHECK("lstm weight-gradient reduce launch");
  }
  return LAS_OK;
}
