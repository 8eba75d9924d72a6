/*
 * las_hip.h — C-ABI of liblas_hip.so, the MI355X (gfx950) implementation of the
 * Listen-Attend-Spell training hot path of sciforce/phones-las.
 *
 * The reference has no FFI: the path sits behind Python functions that build a TF-1 graph
 * (SURVEY.md §8b).  Each entry point below names the reference interface it replaces
 * (file:line under /root/reference).  The Python host in phones-las_amd/ mirrors those
 * Python interfaces (las.ops.*, las.model.*, model_helper.las_model_fn) on top of this ABI
 * through ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors); nothing is
 *     allocated or freed inside the library; scratch comes from caller-provided workspaces
 *     whose sizes the *_workspace_bytes() / *_ws_bytes() queries next to each entry point report;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue;
 *   - row-major, batch-major [B,T,C] tensors; weights in TF layout: LSTM kernel [D+H, 4H]
 *     with gate column order i, j, f, o and bias [4H] (tf.nn.rnn_cell.LSTMCell, forget_bias 1);
 *   - las_bf16 = raw bfloat16 bits; fp32 master weights / optimiser state / losses;
 *   - return 0 on success, a negative las_status otherwise; las_last_error() gives the text
 *     (thread-local).  There is NO CPU fallback: without a HIP device every compute entry
 *     point fails with LAS_ERR_HIP.
 */
#ifndef LAS_HIP_H
#define LAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t las_bf16;

enum las_status {
  LAS_OK = 0,
  LAS_ERR_ARG = -1,      /* bad shape / alignment / unsupported configuration */
  LAS_ERR_HIP = -2,      /* a HIP runtime call failed                          */
  LAS_ERR_WORKSPACE = -3 /* workspace too small                                */
};

enum las_dec_mode {      /* which halves of a decoder step a las_decoder_step_* call runs */
  LAS_DEC_FUSED = 0,          /* LSTM cell, then attention queried by its output (AttentionWrapper around one cell) */
  LAS_DEC_CELL_ONLY = 1,      /* one LSTM cell of a MultiRNNCell / AttentionMultiCell stack (las/model.py:20-69,194-196) */
  LAS_DEC_ATTENTION_ONLY = 2  /* attention queried by `query` (the top cell's output) */
};

enum las_attention {     /* las/model.py:153-166 --attention_type: how a score is formed */
  LAS_ATT_LUONG = 0,              /* h . keys                                  (LuongAttention) */
  LAS_ATT_BAHDANAU = 1,           /* v . tanh(keys + Wq h)                     (BahdanauAttention) */
  LAS_ATT_CUSTOM = 2,             /* relu(Wq h) . keys, keys = relu(memory Wm) (CustomAttention, las/model.py:72-101) */
  LAS_ATT_LUONG_MONOTONIC = 3,    /* h . keys + score_bias                     (LuongMonotonicAttention) */
  LAS_ATT_BAHDANAU_MONOTONIC = 4  /* v . tanh(keys + Wq h) + score_bias        (BahdanauMonotonicAttention) */
};

enum las_att_norm {      /* how scores become alignments */
  LAS_NORM_SOFTMAX = 0,
  LAS_NORM_MONOTONIC_PARALLEL = 1, /* tf.contrib.seq2seq.monotonic_attention(sigmoid(score [+ noise]), prev, 'parallel') */
  LAS_NORM_MONOTONIC_HARD = 2      /* ... (score > 0, prev, 'hard'): inference only, no backward */
};

int las_version(void);
const char* las_last_error(void);
/* The library's diagnostics / A-B switches are integer environment variables (LAS_*: slice heights, member counts, kernel
 * families; see README) that it reads ONCE, at their first use.  las_set_knob overrides one afterwards (the tests' hook). */
int las_set_knob(const char* name, int value);

/* ------------------------------------------------------------------------------------------
 * Dense products on MFMA (bf16 operands, fp32 accumulate).  These replace the Eigen/cuBLAS
 * matmuls TF issues for tf.layers.Dense and the LSTM kernels (model_helper.py:349,
 * las/model.py:168-169,251-257; Appendix A.1 of SURVEY.md).
 * ---------------------------------------------------------------------------------------- */

/* C[M,N] (=|+=) A[M,K] * B[N,K]^T + bias[N].   A, B bf16 with K contiguous (lda, ldb in
 * elements, multiples of 8; K multiple of 8).  C is fp32 (out_bf16 = 0) or bf16.
 * accumulate != 0 adds into C (fp32 only).  batch > 1 runs `batch` independent products with
 * element strides sa/sb/sc.  split_k > 1 splits K over workgroups with fp32 atomics (C must
 * then be fp32 and is pre-zeroed by the call unless accumulate). */
int las_gemm_nt(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, void* C, int64_t ldc,
                const float* bias, int M, int N, int K, int out_bf16, int accumulate, int batch,
                int64_t sa, int64_t sb, int64_t sc, int split_k, void* stream);
/* C [M, N] fp32 (=|+=) A [M, K] * W [N, K]^T + bias with the WEIGHT operand handed in as its LAS_IMAGE_PACK_MFMA_B image (dst_rows =
 * N, dst_cols = K: [N / 16][K / 32][64 lanes][8]): only the activation operand goes through the LDS, a wave fetches its weight
 * fragments from the image straight into registers.  N a multiple of 256, K of 128.  las_pack_mfma_b_bf16: the image of a
 * bf16 matrix [N, K] (the model's weights are packed from their fp32 masters by las_refresh_images). */
int las_pack_mfma_b_bf16(const las_bf16* B, int64_t ldb, int N, int K, las_bf16* image, void* stream);
int las_gemm_nt_bimg(const las_bf16* A, int64_t lda, const las_bf16* b_image, float* C, int64_t ldc, const float* bias, int M, int N,
                     int K, int accumulate, void* stream);
/* The input product of a recurrent layer that runs BESIDE the product (round 4; las_gemm_nt_stream_dirs): C [B*T, N] fp32 = A [B*T, K] * Bm [N, K]^T + bias,
 * row b*T + t = utterance b at time t, the N columns in ndir halves (direction d reads columns [d*N/ndir, (d+1)*N/ndir)).  A
 * 256 x 128 output tile is the rows_per_slice (= las_lstm_slice_rows) utterances of ONE chain group x 256 / rows_per_slice
 * STEPS of its direction's time order (step tau = time tau forward, time length - 1 - tau in the reversed direction; steps
 * beyond an utterance's length are not computed).  The kernel is persistent and XCD-aware: the recurrence
 * (las_lstm_recurrent_fwd_ex(ready = ...), launched FIRST, on another stream) publishes in `ready` on which XCD each of its
 * groups runs; a product workgroup serves the groups of ITS XCD, step block by step block, so that a tile is written into the
 * L2 its consumer reads from, and counts finished column tiles per (group, step block) there.  `ready`:
 * las_gemm_nt_stream_flags(...) 32-bit words, zero before both launches.  N / ndir a multiple of 128, K of 64. */
int las_gemm_nt_stream_supported(int N, int K, int ndir);
size_t las_gemm_nt_stream_flags(int B, int T, int ndir, int rows_per_slice);
/* One A operand per direction: the columns of direction d are formed from A + d * a_dir_stride (elements; 0: one operand).  Under
 * DropoutWrapper(input_keep_prob) the fw and bw cells of a layer read the same input through independent masks
 * (las/ops.py:14-18): the two masked copies lie a_dir_stride apart; a MultiRNNCell stack per direction (las/model.py:111-142)
 * reads its own column range of the layer below: a_dir_stride = that range's width. */
int las_gemm_nt_stream_dirs(const las_bf16* A, int64_t lda, int64_t a_dir_stride, const las_bf16* Bm, int64_t ldb, float* C,
                            int64_t ldc, const float* bias, const int32_t* length, int B, int T, int N, int K, int ndir,
                            int rows_per_slice, uint32_t* ready, void* stream);

/* C (=|+=) (A B^T) * mask / keep, mask[row, col] = [las_uniform(seed, stream_id, row * N + col) < keep]: the gradient through a
 * cell's input dropout (DropoutWrapper(input_keep_prob), las/ops.py:14-18; the mask las_dropout_bf16 drew for that cell in
 * the forward pass) folded into the epilogue of dX = dZ K_x^T.  One direction stores, the other accumulates: no partial dX
 * buffers, no las_dropout_bwd pass over them.  C fp32, contiguous (ldc == N); bulk shapes (M, N > 64). */
int las_gemm_nt_masked(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, float* C, int64_t ldc,
                       int M, int N, int K, int accumulate, float keep, uint32_t seed, uint32_t stream_id, void* stream);

/* C[M,N] += sum_k A[k,M] * B[k,N]  (weight gradients X^T dZ; TF autodiff of the matmuls above,
 * model_helper.py:415).  A [K,M] and B [K,N] bf16 with M resp. N contiguous.  C fp32, always
 * accumulated with atomics (zero it first).  a_shift/period: row k of A is taken from
 * k + a_shift when 0 <= (k % period) + a_shift < period and is zero otherwise (the h_{t-1}
 * operand of dK_h without materialising a shifted copy; period = T).  period = 0 disables.
 * c_perm_h = H > 0: B's columns are gate-interleaved (n = u*4+g, the layout the recurrent kernels
 * use for gates/dz); C is written in TF order (column g*H+u).  N must be 4H. */
int las_gemm_tn(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, float* C, int64_t ldc,
                int M, int N, int K, int a_shift, int period, int c_perm_h, int batch, int64_t sa,
                int64_t sb, int64_t sc, int split_k, void* stream);
/* C += A^T B with a result that does not depend on scheduling (two runs are bit-identical): the K slices store their
 * 128 x 128 tiles in `workspace` (las_gemm_tn_ws_bytes(M, N, split_k) bytes, owned by the caller, one per stream that
 * issues such products) and a second kernel adds them to C in slice order -- no fp32 atomics.  The speller's weight
 * gradients (TF autodiff of the decoder's matmuls, model_helper.py:415) use this form so that training is reproducible.
 * Shapes the slice kernel is not built for (N % 4 != 0) and calls without a sufficient workspace run unsplit (one contributor
 * per output element).  Arguments as las_gemm_tn, batch 1. */
size_t las_gemm_tn_ws_bytes(int M, int N, int split_k);
int las_gemm_tn_ws(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, float* C, int64_t ldc,
                   int M, int N, int K, int a_shift, int period, int c_perm_h, int split_k, float* workspace,
                   size_t workspace_bytes, void* stream);
/* C = A^T B: the same product storing its result (no zeroing of C beforehand, no atomics, K not split): the batched
 * d(keys) = ds^T h and d(memory) = alignments^T d(context) of the speller's backward, whose K is only the U decoder steps
 * (TF autodiff of the attention mechanism's matmuls, model_helper.py:415).  C fp32, or bf16 with out_bf16 != 0 (d(keys)
 * is only ever an operand of further products). */
int las_gemm_tn_store(const las_bf16* A, int64_t lda, const las_bf16* B, int64_t ldb, void* C, int64_t ldc,
                      int M, int N, int K, int a_shift, int period, int c_perm_h, int batch, int64_t sa,
                      int64_t sb, int64_t sc, int out_bf16, void* stream);

/* All weight gradients of one LSTM direction in ONE product (dz is read once instead of three times):
 *   kernel_grad[0:D, :]   += x^T dz              (input half of the [D+H, 4H] TF kernel)
 *   kernel_grad[D:D+H, :] += shift(y)^T dz       (recurrent half: row k of y pairs with dz row k - a_shift inside a
 *                                                 `period` of rows, i.e. h_{t-1} of the direction's own time order)
 *   bias_grad[:]          += column sums of dz
 * x [K, D] bf16 (row stride ldx), y [K, H] bf16 (ldy), dz [K, 4H] bf16 with GATE-INTERLEAVED columns (ldz); outputs
 * fp32 in TF column order (g*H+u).  K is cut into split_k slices; with `workspace`
 * (las_gemm_tn_lstm_workspace_bytes, caller-owned, reusable once the stream has passed this call) every slice stores
 * its partial product and a second kernel sums them into the outputs; workspace NULL accumulates with fp32 atomics
 * instead (several times slower at training shapes).  split_k | LAS_TN_SPLIT_WIDE asks for 128 x 512 output tiles (4H a multiple
 * of 512; half the workgroups per K slice, so callers double the slices) instead of 128 x 256. */
#define LAS_TN_SPLIT_WIDE 0x10000
size_t las_gemm_tn_lstm_workspace_bytes(int D, int H, int split_k);
int las_gemm_tn_lstm(const las_bf16* x, int64_t ldx, int D, const las_bf16* y, int64_t ldy, int H, int a_shift,
                     int period, const las_bf16* dz, int64_t ldz, float* kernel_grad, float* bias_grad, int K,
                     int split_k, float* workspace, void* stream);
/* dst_bf16[r, c] = src_f32[r, c] (transpose = 0) or dst[c, r] = src[r, c] (transpose = 1), with the
 * destination window [dst_rows, dst_cols] (row stride ldd) zero-padded.  `batch` windows at element
 * strides src_bstride / dst_bstride.  src_col_perm_h = H > 0 reads the source's columns through the
 * gate-interleaving permutation (logical column u*4+g <- TF column g*H+u; cols must be 4H).  Used to
 * derive the bf16 operand copies of the fp32 master weights and to cast/pad the (B,T,F) feature batch
 * (utils/dataset_utils.py:254-281 padded_batch). */
int las_cast_bf16(const float* src, int64_t lds, int rows, int cols, las_bf16* dst, int64_t ldd,
                  int dst_rows, int dst_cols, int transpose, int batch, int64_t src_bstride,
                  int64_t dst_bstride, int src_col_perm_h, void* stream);

/* The image rebuilds of a whole model in one launch (after every optimiser step the bf16 operand copies of the fp32
 * master weights are rebuilt; one launch per image is launch-bound).  `jobs_dev` is a DEVICE array of njobs entries:
 *   LAS_IMAGE_CAST            the las_cast_bf16 window (batch 1) with the same field meanings;
 *   LAS_IMAGE_PACK_RECURRENT  las_lstm_pack_recurrent: src = K_h (ld 4H), rows = H, dst = packed image;
 *   LAS_IMAGE_BIAS_INTERLEAVE fp32 dst[u*4+g] = src[g*H+u], rows = H (the LSTM bias in the recurrent kernels' order);
 *   LAS_IMAGE_COPY_F32        fp32 dst[0:cols] = src[0:cols];
 *   LAS_IMAGE_PACK_INPUT      las_lstm_pack_input: src = kernel (ld 4H), rows = D, cols = H, dst_rows = chunks, dst = packed image;
 *   LAS_IMAGE_PACK_MFMA_B     bf16 image of src [rows, cols] (row stride lds) in matrix-core B-fragment order, zero padded to
 *                             dst_rows (multiple of 16) x dst_cols (multiple of 32): [dst_rows / 16][dst_cols / 32][64 lanes][8]
 *                             with lane l = row tile * 16 + (l & 15), columns chunk * 32 + (l >> 4) * 8 + 0..7 -- every
 *                             fragment one contiguous KB (rows 2^k bytes apart all land on one L2 channel otherwise);
 *                             transpose = 1: the image of src^T (src is [cols, rows], row stride lds); perm_h = H: the source's
 *                             column axis is read through the gate interleaving (index u * 4 + g <- column g * H + u); ldd > 0: the
 *                             window is the K range [reserved, reserved + dst_cols) of an image whose whole K is ldd. */
enum las_image_kind { LAS_IMAGE_CAST = 0, LAS_IMAGE_PACK_RECURRENT = 1, LAS_IMAGE_BIAS_INTERLEAVE = 2, LAS_IMAGE_COPY_F32 = 3,
                      LAS_IMAGE_PACK_MFMA_B = 4, LAS_IMAGE_PACK_INPUT = 5 };
typedef struct las_image_job {
  const float* src;
  void* dst;
  int64_t lds, ldd;
  int32_t rows, cols, dst_rows, dst_cols;
  int32_t transpose, perm_h, kind, reserved;
} las_image_job;
int las_refresh_images(const las_image_job* jobs_dev, int njobs, void* stream);

/* out[r, c] = bf16(a[r, c] + b[r, c]) over a [rows, cols] window (row strides in elements), b nullable: the general decoder's
 * d(attention_t) = d(output projection)_t + d(feed into step t+1) as the attention layer's bf16 GEMM operand, in one launch
 * instead of a copy, an add and a cast. */
int las_add_cast_bf16(const float* a, int64_t lda, const float* b, int64_t ldb, las_bf16* out, int64_t ldo, int rows, int cols,
                      void* stream);

/* A one-wave kernel that keeps `stream` busy for about `microseconds` (0..1000): put in front of side-stream work that
 * becomes runnable at the same moment as a persistent recurrent launch on the main stream and should not take CUs before
 * that launch's workgroups are resident (a chain that finds some of its CUs taken starts late as a whole). */
int las_stream_delay(int microseconds, void* stream);
/* Diagnostics / tests: `workgroups` one-wave workgroups that each add 1 to counts[XCD they run on] (8 device words) and then stay
 * resident for spin_us microseconds holding lds_bytes of LDS: a census of the dispatcher's workgroup -> XCD placement, and a way to
 * take CUs away from a launch that needs them (the persistent kernels' bounded waits under CU pressure: tests/test_gpu_step_forms.py). */
int las_xcd_histogram(uint32_t* counts, int workgroups, int spin_us, int lds_bytes, void* stream);

/* Can a kernel on `setter_stream` run while an EARLIER-enqueued kernel on `waiter_stream` is still running?  A process that
 * has created many streams gets them mapped onto a handful of hardware queues; two streams that share one run their kernels
 * in submission order.  The streamed input products (las_gemm_nt_stream_dirs beside las_lstm_recurrent_fwd_ex) need the answer to be
 * yes for their pair of streams.  Enqueues a one-thread kernel on waiter_stream that waits up to wait_us for a word that a
 * one-thread kernel enqueued AFTERWARDS on setter_stream writes; once both streams are synchronised, words[1] == 1: concurrent,
 * 2: the setter never ran beside the waiter.  words: two int32 in device memory. */
int las_stream_concurrency_probe(void* waiter_stream, void* setter_stream, int32_t* words, int wait_us);

/* Several small fills / copies in ONE launch (the decoder's per-step scratch: zeroed accumulators, the initial state
 * rows).  The job table travels as a kernel argument: no device-side table, nothing to upload.  Every job is a 2-D window
 * [rows, cols] of ELEMENTS with row strides ldd / lds (elements):
 *   LAS_FILL_ZERO32   dst (4-byte elements) = 0;             LAS_FILL_ZERO16   dst (2-byte elements) = 0;
 *   LAS_FILL_COPY_F32 dst fp32 = src fp32;                   LAS_FILL_CAST_BF16 dst bf16 = round(src fp32).
 * src == NULL in a copy / cast job writes zeros.  njobs <= LAS_FILL_MAX_JOBS. */
enum las_fill_kind { LAS_FILL_ZERO32 = 0, LAS_FILL_ZERO16 = 1, LAS_FILL_COPY_F32 = 2, LAS_FILL_CAST_BF16 = 3 };
enum { LAS_FILL_MAX_JOBS = 12 };
typedef struct las_fill_job {
  void* dst;
  const void* src;
  int64_t rows, cols, ldd, lds;
  int32_t kind, reserved;
} las_fill_job;
int las_fill_many(const las_fill_job* jobs_host, int njobs, void* stream);

/* out[n] += sum_m X[m, n] for a bf16 [M,N] matrix (bias gradients).  out_perm_h = H > 0: X's columns are
 * gate-interleaved (u*4+g) and the sum of column n lands at TF index g*H+u. */
int las_colsum_bf16(const las_bf16* X, int64_t ldx, int M, int N, float* out, int out_perm_h, void* stream);
/* The same sums with a result that does not depend on scheduling (no fp32 atomics: the row chunks' partial sums meet in
 * `workspace` and the last block of a column group adds them in chunk order).  workspace: las_colsum_ws_bytes(M, N) bytes,
 * 16-byte aligned, owned by the caller, its first 4096 bytes (the per-group counters) ZERO before the first use; launches
 * leave them zero.  One workspace per stream that issues such sums. */
size_t las_colsum_ws_bytes(int M, int N);
int las_colsum_bf16_ws(const las_bf16* X, int64_t ldx, int M, int N, float* out, int out_perm_h, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Listener: one (Bi)LSTM layer = las/ops.py:23-46 `bilstm` (tf.nn.bidirectional_dynamic_rnn /
 * dynamic_rnn with sequence_length over LSTMCell of las/ops.py:10-20).
 * ---------------------------------------------------------------------------------------- */

/* Pack the recurrent half K_h = kernel[D:D+H, :] (fp32, ld 4H) of one direction into the
 * MFMA-fragment-major bf16 image the forward recurrent kernel streams (size H*4H). */
int las_lstm_pack_recurrent(const float* kernel_h, int H, las_bf16* packed, void* stream);

/* Forward recurrence.  xproj [B,T,ndir*4H] fp32 holds x_t*K_x + b on entry (las_gemm_nt) and
 * the activated gates (sigma(i), tanh(j), sigma(f+1), sigma(o)) on exit (saved for backward).
 * Within a direction the 4H columns are GATE-INTERLEAVED: column u*4+g holds gate g (i,j,f,o) of
 * unit u, so one lane reads/writes its four gates with a single 16-byte access; build the operands
 * with las_cast_bf16(..., src_col_perm_h = H).  dz of the backward uses the same column order.
 * wpacked: ndir images from las_lstm_pack_recurrent.  y [B,T,ndir*H] bf16: outputs, fw in
 * [0,H) and bw in [H,2H) (the tf.concat of las/ops.py:81), zero for t >= length.  cbuf
 * [B,T,ndir*H] fp32: cell states (saved).  c_last/h_last [ndir,B,H] fp32: final states
 * (state carried through t >= length, Appendix A.3).  Direction 1 runs the time-reversed
 * recurrence of bidirectional_dynamic_rnn (reverse_sequence on the valid prefix). */
int las_lstm_recurrent_fwd(float* xproj, const las_bf16* wpacked, const int32_t* length, las_bf16* y,
                           float* cbuf, float* c_last, float* h_last, void* workspace, int B, int T, int H,
                           int ndir, void* stream);

/* The forward recurrence WITH the input projection (round 4; the bottom listener layer, las/ops.py:35-46 over the features):
 * x [B,T,ldx] bf16 (direction d reads x + d * x_dir_stride elements: 0 when both directions read the same input, the copy
 * stride when each has its own input-dropout mask), Dp = its valid width (multiple of 8, the columns from the layer's D on
 * zero), kx_packed = ndir images of las_lstm_pack_input, bias [ndir*4H] fp32 gate-interleaved.  z_t = x_t K_x + b + h_{t-1} K_h
 * is formed inside the step (the x products are issued while the step waits for its peers' h), so no las_gemm_nt and no
 * [B,T,ndir*4H] fp32 round trip precede the launch; `gates` is output only.  Supported when las_lstm_fused_input_chunks(H, Dp)
 * > 0 (H in {128, 256, 512}, Dp <= 96); everything else as las_lstm_recurrent_fwd. */
int las_lstm_fused_input_chunks(int H, int Dp);
/* K_x = kernel[0:D, :] (fp32, ld 4H) of one direction -> the fragment-major bf16 image over `chunks` (=
 * las_lstm_fused_input_chunks) 32-deep K chunks, (H/16) * chunks * 4 * 512 elements, zero from row D on. */
int las_lstm_pack_input(const float* kernel, int D, int H, int chunks, las_bf16* packed, void* stream);
/* The forward recurrence with its options (las_lstm_recurrent_fwd = all of them off):
 *   x != NULL      fused input projection, as described above;
 *   ready != NULL  STREAMED input projection: xproj is being produced by a las_gemm_nt_stream_dirs launch that runs beside this
 *                  one on another stream (launch this kernel first; hold the product back a few microseconds with
 *                  las_stream_delay so that the chain's workgroups are resident first).  ready = the buffer shared with the
 *                  product (las_gemm_nt_stream_flags words, zeroed by the caller before both launches: this launch writes its
 *                  groups' XCDs into it and waits on the product's counters); ready_count = column tiles per block = N / ndir / 128. */
typedef struct las_lstm_fwd {
  float* xproj;                  /* [B,T,ndir*4H] fp32: x K_x + b on entry (unless x != NULL), the saved gates on exit */
  const las_bf16* wpacked;
  const int32_t* length;
  las_bf16* y;
  float* cbuf;
  float* c_last;
  float* h_last;
  void* workspace;
  int32_t B, T, H, ndir;
  const las_bf16* x;             /* fused input projection (optional) */
  int64_t ldx, x_dir_stride;
  int32_t Dp, reserved0;
  const las_bf16* kx_packed;
  const float* bias;
  uint32_t* ready;               /* streamed input projection (optional): shared with las_gemm_nt_stream_dirs, written by both */
  int32_t ready_count;
  int32_t rows_per_slice;        /* 0: the library's choice (las_lstm_slice_rows); 4 / 8 / 16: this launch's utterances per slice -- the caller of a
                                  * streamed product passes the value it laid `ready` out for (round 5: 8-row slices under a product that the
                                  * 4-row form, with twice the chain workgroups, leaves too few CUs) */
} las_lstm_fwd;
int las_lstm_recurrent_fwd_ex(const las_lstm_fwd* p, void* stream);

/* Bytes of device scratch the two recurrent kernels need for (B, H, ndir): a 64-byte header plus the
 * inter-workgroup exchange buffer of the cooperating groups (0 = unsupported num_units).  The caller
 * hands it over ZEROED ONCE, when it allocates it, and otherwise leaves it alone; the launches keep it consistent
 * themselves (round 4: no memset node per launch -- every tag in the exchange buffer is offset by a launch base kept in
 * header word 4, which the last workgroup of a launch moves past all tags of that launch, so what earlier launches left
 * behind can never satisfy a wait; one launch at a time per workspace).  The header's first uint32 is STICKY -- a launch
 * ORs a bit into it when a bounded inter-workgroup wait timed out (results of that launch are invalid), and only the host
 * clears it after reading it (one workspace serves every layer, forward and backward: see las_status_collect). */
size_t las_lstm_workspace_bytes(int B, int H, int ndir);
/* Utterances per slice the recurrent kernels will use for this shape (16 = full MFMA tiles; 8 = half-filled tiles on
 * twice as many chains, chosen for 256 units while every chain and its prefetch companion still find a CU each: the
 * per-step latency is mostly element-wise work per lane).  LAS_LSTM_ROWS=16|8 in the environment overrides (tests). */
int las_lstm_slice_rows(int B, int H, int ndir);
/* Workgroups (chain members + prefetch companions, one CU each) of a forward launch for this shape: a streamed input product
 * (las_gemm_nt_stream_dirs) beside it needs CUs of its own, so callers only stream when this leaves some (0 = unsupported). */
int las_lstm_fwd_workgroups(int B, int H, int ndir);

/* Backward recurrence (reverse-mode AD of the loop above).
 * gates/cbuf: saved by the forward.  dy [B,T,ndir*H] fp32: gradient w.r.t. y.  dc_last/dh_last
 * [ndir,B,H] fp32 or NULL.  kh_bf16: ndir bf16 copies of K_h [H,4H] with GATE-INTERLEAVED columns (u*4+g,
 * las_cast_bf16(..., src_col_perm_h = H)): the gate columns of a block of units are contiguous.
 * dz [B,T,ndir*4H] bf16 (out): gradient w.r.t. the gate pre-activations, zero for t >= length;
 * the caller derives dX, dK_x, dK_h, db from it with las_gemm_nt / las_gemm_tn / las_colsum. */
int las_lstm_recurrent_bwd(const float* gates, const float* cbuf, const float* dy, const float* dc_last,
                           const float* dh_last, const las_bf16* kh_bf16, const int32_t* length,
                           las_bf16* dz, void* workspace, int B, int T, int H, int ndir, void* stream);

/* len_out[b] = len[b]/2 + len[b]%2  (las/ops.py:65 pyramidal_stack). */
int las_pyramid_lengths(const int32_t* len_in, int32_t* len_out, int B, void* stream);
/* The lengths after 1, 2, ..., levels stackings in one launch: len_out [levels, B] (the listener asks for all of them before
 * its first layer, so that no tiny launch sits between a layer's recurrence and the next layer's input product). */
int las_pyramid_lengths_multi(const int32_t* len_in, int32_t* len_out, int B, int levels, void* stream);

/* ------------------------------------------------------------------------------------------
 * Speller step kernels: AttentionWrapper(LSTMCell, Luong|Bahdanau) of las/model.py:145-202
 * driven by BasicDecoder/dynamic_decode (las/model.py:295-296,346-347); SURVEY.md A.5-A.7.
 * ---------------------------------------------------------------------------------------- */

/* One decoder step for every batch row, after z = [attention_{t-1}, h_{t-1}] * K[E:, :] has been
 * produced by las_gemm_nt: adds the token rows K[id] (the one-hot feed of las/model.py:246 is a row
 * gather) and the bias, applies the LSTM cell, scores h_t against the keys (Luong: dot; Bahdanau:
 * v.tanh(keys + W_q h)), takes the softmax over t' < mem_len and forms context = align * values.
 * Every per-step tensor is addressed as base + b * row_stride, so the caller can keep all U steps in
 * [B,U,...] buffers.  `parts` workgroups per utterance split the context columns (speed only). */
typedef struct las_dec_step {
  int32_t B, Hd, M, Tm, attention, mode;  /* Tm = padded memory length T'; enum las_attention; enum las_dec_mode */
  const float* z;                /* [B,4Hd] contiguous: recurrent + attention-feed part of the pre-activations */
  const las_bf16* tok_rows;      /* [E,4Hd] bf16 rows of the cell kernel for the token feed, or NULL */
  const int32_t* tok_ids;        /* previous token of utterance b at tok_ids[b * tok_stride] */
  int64_t tok_stride;
  const float* bias;             /* [4Hd] */
  const float* c_prev;           /* c_{t-1}, row stride ldcp */
  int64_t ldcp;
  float* gates_out;              /* activated gates i,j,f,o (saved), row stride ldg */
  int64_t ldg;
  float* c_out;                  /* c_t (saved), row stride ldco */
  int64_t ldco;
  las_bf16* h_out;               /* h_t as bf16, row stride ldh */
  int64_t ldh;
  las_bf16* h_out2;              /* optional second copy (next step's GEMM operand), row stride ldh2; LAS_DEC_ATTENTION_ONLY:
                                  * optional copy of the QUERY (the attention layer's operand [query | context]) */
  int64_t ldh2;
  const las_bf16* keys;          /* [B,Tm,Hd] */
  const las_bf16* values;        /* [B,Tm,M]  (memory, zero beyond mem_len) */
  const int32_t* mem_len;        /* [B] */
  const las_bf16* wq;            /* Bahdanau query_layer kernel [Hd(in),Hd(out)] bf16, or NULL */
  const float* att_v;            /* Bahdanau attention_v [Hd], or NULL */
  float* align_out;              /* alignments fp32 (saved), row stride lda */
  las_bf16* align_bf16;          /* optional bf16 copy (operand of the d(memory) GEMM), row stride lda */
  int64_t lda;
  float* pq_out;                 /* Bahdanau processed query (saved), row stride ldpq, or NULL */
  int64_t ldpq;
  las_bf16* ctx_out;             /* context as bf16, row stride ldc */
  int64_t ldc;
  las_bf16* ctx_out2;            /* optional second copy (next step's GEMM operand), row stride ldc2 */
  int64_t ldc2;
  /* DropoutWrapper(input_keep_prob = drop_keep) on the cell input [token one-hot | attention] (las/ops.py:14-18):
   * element e of utterance b at step t uses draw (drop_seed, drop_stream, (t*B + b)*feed_width + e).  The token row
   * is scaled here; ctx_out2 is written already masked for step+1.  drop_keep >= 1 disables. */
  float drop_keep;
  uint32_t drop_seed, drop_stream;
  int32_t step, feed_width;
  int32_t feed_plain;            /* ctx_out2 under drop_keep < 1: 0 = masked with the draws above; 1 = written unmasked; 2 = masked with
                                  * draw (drop_seed, feed_stream0 + step + 1, b * M + column): one generator stream per step */
  const las_bf16* query;         /* LAS_DEC_ATTENTION_ONLY: the query [B,Hd] bf16, row stride ldq */
  int64_t ldq;
  /* monotonic attention (enum las_att_norm != 0); all NULL/0 otherwise */
  int32_t norm;
  const float* score_bias;       /* device scalar `attention_score_bias` added to every score */
  const float* prev_align;       /* alignments of step t-1, row stride ldpa; NULL = the initial dirac at frame 0 */
  int64_t ldpa;
  float* p_out;                  /* p_choose (saved for backward), row stride ldp, or NULL */
  int64_t ldp;
  float noise_scale;             /* sigmoid_noise: score += noise_scale * N(0,1), draw (step*B + b)*Tm + t; 0 = none */
  uint32_t noise_seed, noise_stream;
  uint32_t feed_stream0;         /* feed_plain 2 */
} las_dec_step;
int las_decoder_step_fwd(const las_dec_step* s, int parts, void* stream);

/* All U steps of the fused decoder (las_dec_mode FUSED, softmax attentions) in ONE launch: a persistent kernel whose
 * workgroups alternate between the per-step product z_t = x_t K (x_t = row t of the operand buffer: [attention_{t-1},
 * h_{t-1}], written by step t-1 through h_out2 / ctx_out2) and the step itself; inside a step they hand results to
 * each other as tagged 8-byte granules through the workspace and meet at one flag barrier; the keys / values a
 * workgroup needs stay in its LDS for the whole sequence when they fit (else they are read from L2 every step).  `s` describes step 0; the inc_* fields are the element increments of
 * its per-step pointers (h_out2 / ctx_out2 are not written by the last step).  Replaces the U x (las_gemm_nt +
 * las_decoder_step_fwd) loop of las/model.py:276-296 when scheduled sampling is off.  `workspace`
 * (las_decoder_persist_workspace_bytes) is handed over zeroed once; the call zeroes everything behind its 64-byte status
 * header; the header's first uint32 is sticky: a bit is ORed in when a bounded wait timed out, the host clears it. */
typedef struct las_dec_persist {
  las_dec_step s;
  int32_t U, K_in;               /* steps; columns of the operand rows / of kT (multiple of 32) */
  int64_t inc_tok, inc_cprev, inc_gates, inc_cout, inc_h, inc_h2, inc_align, inc_pq, inc_ctx, inc_ctx2;
  const las_bf16* x;             /* operand rows: utterance b, step t at x + b*ldx + t*inc_x */
  int64_t ldx, inc_x;
  const las_bf16* kT;            /* [4Hd, K_in] bf16, row n = output column n, row stride ldk */
  int64_t ldk;
  const las_bf16* wq_packed;     /* optional (Bahdanau / Custom, Hd 128 or 256): the LAS_IMAGE_PACK_MFMA_B image of the query layer's
                                  * kernel TRANSPOSED (rows = outputs): the processed query h_t Wq on the matrix cores */
  float* sc_all;                 /* unused (the raw scores travel through the workspace); may be NULL */
  int64_t ld_sc;                 /* >= Tm, multiple of 32 (whole cache lines per row) */
  void* workspace;
  /* scheduled sampling (utils/training_helper.py:48-87), sampling_prob > 0: after step t the kernel replaces the token
   * fed at step t+1 (s.tok_ids[b*tok_stride + t+1], which must hold the teacher's token on entry) by a draw from
   * Categorical(logits_t), logits_t = context_t W_proj + b, for the utterances selected with probability sampling_prob.
   * The draws are counter-based (same generator streams as las_sample_tokens(step = t)), so a group of utterances
   * without a selection at step t skips the phase; logits are formed for the selected utterances only and are NOT an
   * output (the caller computes all logits with one las_gemm_nt after the loop; `logits` / `ld_logits` / `teacher` are
   * unused and may be NULL / 0). */
  float sampling_prob;
  uint32_t seed;
  const int32_t* teacher;
  int64_t teacher_stride;
  const las_bf16* wprojT;        /* [V, M] bf16, row stride ldw */
  int64_t ldw;
  const float* bproj;            /* [Vp] */
  float* logits;                 /* fp32, utterance stride ld_logits, Vp per step */
  int64_t ld_logits;
  float* plog;                   /* [U, B, 4, Vp] fp32 scratch */
  int32_t V, Vp;
  /* Attention layer (attention_layer_size, or 2 * binf_count under --binf_projection: las/model.py:179-200), walT != NULL:
   * attention_t = [h_t | context_t] W_al (tf.layers.Dense, no bias) is the decoder's output and the feed of step t+1: the
   * launch writes it (bf16) to att_out[b*ld_att + t*A ..] and into columns [x_att_off, x_att_off + A) of the operand row of
   * step t+1; s.ctx_out2 must then be NULL (the context is not fed).  A: a multiple of 16, at most 512; (Hd + M) / 32 <= 40.
   * Monotonic normalisers (s.norm = LAS_NORM_MONOTONIC_PARALLEL with the monotonic attentions): s.p_out receives p_choose of
   * every step (step increment inc_p, the alignments' increment for the rows of s.align_out it chains through). */
  const las_bf16* walT;          /* [A, Hd + M] bf16, row n = output column n, row stride ld_wal */
  int64_t ld_wal;
  int32_t A, x_att_off;
  las_bf16* att_out;
  int64_t ld_att;
  int64_t inc_p;
  /* Second decoder cell (decoder_layers = 2, the reference's default depth; round 4), k1T != NULL; softmax attentions, no
   * attention layer; las_decoder_persist2_supported.  Input dropout (s.drop_keep < 1, the reference's default 0.8): the token row's
   * scale as in `s`; element (b, c), c < win_l, of cell l's input row at step t is scaled by draw (s.drop_seed, in_stream_l + t,
   * b * win_l + c) -- the streams of U x las_dropout_bf16 on the step-by-step path; win0 = M.  Cell 0's attention feed is
   * written masked into its operand rows `x`; cell 1's row is masked where the launch reads its pieces (h1, s.ctx_out, s.h_out
   * stay undropped).  Scheduled sampling as above, on
   * the output row of the wiring (the context, or h1_t).
   *   wiring 0  MultiRNNCell inside the AttentionWrapper (las/model.py:194-200): cell 1 reads [h0_t | h1_{t-1}] (K1_in = 2 Hd), the
   *             attention is queried with h1_t and its context is the output and the feed of cell 0;
   *   wiring 1  AttentionMultiCell (--bottom_only, las/model.py:36-69): the attention is queried with h0_t, cell 1 reads
   *             [attention_t | attention_{t-1} | h1_{t-1}] (K1_in = 2 M + Hd), its h1_t is the output.
   * k1T [4Hd, K1_in] bf16 (row n = output column n of cell 1's kernel, rows in that order; row stride ldk1), bias1 [4Hd];
   * c1 [B, U+1, Hd] fp32 and h1 [B, U+1, Hd] bf16: row 0 = the initial state (in), row t+1 = the state after step t (out);
   * gates1 [B, U, 4Hd] fp32 (out).  The fields of `s` describe cell 0 and the attention as without a second cell. */
  const las_bf16* k1T;
  int64_t ldk1;
  int32_t K1_in, wiring;
  const float* bias1;
  float* c1;
  float* gates1;
  las_bf16* h1;
  int32_t win0, win1;            /* masked leading columns of cell 0's operand row (the attention feed) and of cell 1's */
  uint32_t in_stream0, in_stream1;
  /* Dense token feed (embedding_size > 0 under input dropout, las/model.py:230-237; s.tok_rows NULL): the operand rows carry the
   * (masked) embedded token in their first T0 columns, filled by the caller for the teacher's tokens; win0 = T0 + M,
   * s.feed_width = win0.  Under scheduled sampling the launch rewrites those columns of row t+1 for a sampled token from
   * emb [V, >= T0] bf16 (row stride ld_emb, zero beyond the embedding width). */
  const las_bf16* emb;
  int64_t ld_emb;
  int32_t T0, reserved_t0;
} las_dec_persist;
int las_decoder_persist_supported(int Hd, int M, int K_in, int attention, int norm);   /* 1 if the shapes fit */
int las_decoder_persist2_supported(int Hd, int M, int K_in, int K1_in, int attention, int wiring);   /* ... with a second cell */
/* ... with an attention layer of A outputs (A = 0: none) and / or a monotonic normaliser (decoder_units 128 / 256) */
int las_decoder_persist_al_supported(int Hd, int M, int K_in, int A, int attention, int norm);
size_t las_decoder_persist_workspace_bytes(int B, int Tm, int Hd, int M);   /* status, group flags, exchange granules */
/* Utterances a persistent launch works on at once: the 32 workgroups of every group of 8 utterances need a CU each and
 * must be resident together: 64 on a 256-CU MI355X.  A launch accepts up to four times as many (its blocks are laid out
 * in chunks of 8 groups that the dispatcher completes one after the other). */
int las_decoder_persist_max_batch(void);
int las_decoder_persist_fwd(const las_dec_persist* p, void* stream);

/* Backward of one decoder step.  d(context) = dctx_a + dctx_b; the kernel
 * back-propagates through context, softmax and score into h_t, adds dh_rec, runs the LSTM cell
 * backward and emits dz (bf16) for the dense products; dc is updated in place to dc_{t-1}. */
typedef struct las_dec_step_bwd {
  int32_t B, Hd, M, Tm, attention, mode;
  const float* dctx_a;           /* fp32, row stride ldda (e.g. dlogits * W_proj^T), or NULL */
  int64_t ldda;
  const float* dctx_b;           /* fp32, row stride lddb (attention-feed gradient from step t+1), or NULL */
  int64_t lddb;
  las_bf16* dctx_save;           /* optional bf16 copy of the total (operand of the d(memory) GEMM) */
  int64_t ldds;
  const float* dh_rec;           /* gradient into h_t from step t+1 (row stride ldr), or NULL */
  int64_t ldr;
  float* dc;                     /* [B,Hd] contiguous, in: dc_t, out: dc_{t-1} */
  const float* gates;            /* saved, row stride ldg */
  int64_t ldg;
  const float* c_new;            /* c_t, row stride ldcn */
  int64_t ldcn;
  const float* c_prev;           /* c_{t-1}, row stride ldcp */
  int64_t ldcp;
  const float* align;            /* saved alignments, row stride lda */
  int64_t lda;
  const float* pq;               /* Bahdanau saved processed query, row stride ldpq */
  int64_t ldpq;
  const las_bf16* keys;
  const las_bf16* values;
  const int32_t* mem_len;
  const las_bf16* wq_t;          /* Bahdanau query_layer kernel transposed [Hd(out),Hd(in)] bf16 */
  const float* att_v;
  las_bf16* dz;                  /* out: gate pre-activation gradient, row stride ldz */
  int64_t ldz;
  las_bf16* ds_out;              /* out: score gradient bf16 (Luong: operand of the d(keys) GEMM), row stride ldso */
  int64_t ldso;
  float* dkeys_acc;              /* Bahdanau: [B,Tm,Hd] fp32 accumulated d(keys) */
  float* dv_acc;                 /* Bahdanau: [Hd] fp32 accumulated d(attention_v) */
  las_bf16* dpq_out;             /* Bahdanau: d(processed query) bf16 (operand of d(query_layer)), row stride lddpq */
  int64_t lddpq;
  float drop_keep;               /* as in las_dec_step: dctx_b is masked with step+1's attention-feed mask */
  uint32_t drop_seed, drop_stream;
  int32_t step, feed_width;
  float* dq_out;                 /* LAS_DEC_ATTENTION_ONLY: d(query) fp32 out, row stride lddq */
  int64_t lddq;
  const float* dh_b;             /* further gradient sources into h_t (fp32), or NULL */
  int64_t ldhb;
  const float* dh_c;
  int64_t ldhc;
  /* monotonic attention (LAS_NORM_MONOTONIC_PARALLEL); NULL/0 otherwise */
  int32_t norm;
  const float* p;                /* saved p_choose, row stride ldp */
  int64_t ldp;
  const float* prev_align;       /* alignments of step t-1 (row stride ldpa); NULL = the dirac at frame 0 */
  int64_t ldpa;
  float* dalign_carry;           /* [B,Tm] fp32 (row stride ldcarry): in: gradient into align_t from step t+1's
                                  * normaliser (zero at the last step); out: the same for align_{t-1} */
  int64_t ldcarry;
  float* dbias_acc;              /* accumulated d(attention_score_bias) (one float) */
} las_dec_step_bwd;
int las_decoder_step_bwd(const las_dec_step_bwd* s, void* stream);

/* The backward counterpart of las_decoder_persist_fwd (softmax attentions): all U steps, last to first, in one launch,
 * replacing the U x (las_decoder_step_bwd + las_gemm_nt) loop.  `s` describes step 0 (dctx_b / dh_rec are ignored:
 * the feed gradient dz_t K^T of step t+1 -- [B, W] fp32: d attention_{t-1} in columns [0,M), d h_{t-1} in [M,W) -- travels
 * inside the launch through the workspace); per-step pointers advance by the inc_* element counts.  dfeed_all receives
 * ONE row block [B, W]: step 0's, the gradient into the initial feed / state.  s.dc is updated in place to dc_{-1}. */
typedef struct las_dec_persist_bwd {
  las_dec_step_bwd s;
  int32_t U, W;                  /* steps; W = M + Hd */
  int64_t inc_a, inc_save, inc_gates, inc_c, inc_align, inc_dz, inc_ds, inc_pq;
  const las_bf16* kc;            /* [W, 4Hd] bf16: row n = row n of the cell kernel below the token rows, stride ldk */
  int64_t ldk;
  float* dfeed_all;              /* [U, B, W] fp32 */
  void* sum_workspace;           /* optional, Bahdanau: las_decoder_sum_workspace_bytes(blocks, Hd) bytes with blocks = 32 * (ceil(B / 8) rounded up to 8: the launch's grid), first word zero
                                  * before the first use (the launch leaves it zero): d(attention_v) is then summed over the
                                  * workgroups in a fixed order instead of with fp32 atomics (bit-reproducible training) */
  float* dhp_all;                /* unused (the partial dh travel through the workspace); may be NULL */
  void* workspace;               /* las_decoder_persist_workspace_bytes(B, Tm, Hd, M) */
  /* Second decoder cell (decoder_layers = 2; round 4), k1c != NULL: the backward of las_dec_persist's two wirings in one launch;
   * decoder_units 128 / 256; las_decoder_persist2_bwd_supported.  `s` describes cell 0 and the attention (s.gates / c_new /
   * c_prev / dz / dc: cell 0's), the fields here cell 1 with the SAME row strides and step increments (gates1 [B,U,4Hd] like
   * s.gates, c1 [B,U+1,Hd]: row t = c1_{t-1}, dz1 like s.dz, dc1 [B,Hd] in/out like s.dc).  Replaces the U x (2 cell backward
   * launches + attention backward + 2 las_gemm_nt + the adds between them) loop of the step-by-step path.
   *   wiring 0: s.dctx_a = d(outputs) [.., M] (the context is the output); cell 1's product P1_t = dz1_t K1^T is
   *             [d h0_t | d h1_{t-1}] (W1 = 2 Hd);
   *   wiring 1: s.dctx_a = NULL, d_out1 = d(outputs) [.., Hd] (h1_t is the output, row stride ld_dout1, step increment
   *             inc_dout1); P1_t = [d attention_t | d attention_{t-1} | d h1_{t-1}] (W1 = 2 M + Hd).
   * k1c [W1, 4Hd] bf16: row n = input row n of cell 1's kernel.  dfeed_all [B, W] / dfeed1_all [B, W1] receive step 0's
   * products (the gradients into the initial feed / states; the masked columns are returned UNMASKED).  Input dropout
   * (s.drop_keep < 1): element (b, c), c < win_l, of cell l's input row at step t was scaled by draw (s.drop_seed,
   * in_stream_l + t, b * win_l + c) (las_dec_persist); its gradient is scaled likewise where the products are consumed. */
  const las_bf16* k1c;
  int64_t ldk1;
  int32_t W1, wiring;
  const float* gates1;
  const float* c1;
  las_bf16* dz1;
  float* dc1;
  float* dfeed1_all;
  const float* d_out1;
  int64_t ld_dout1, inc_dout1;
  int32_t win0, win1;
  uint32_t in_stream0, in_stream1;
} las_dec_persist_bwd;
int las_decoder_persist_bwd_supported(int Hd, int M, int W, int attention, int norm);
int las_decoder_persist2_bwd_supported(int Hd, int M, int W, int W1, int attention, int wiring);
int las_decoder_persist_bwd(const las_dec_persist_bwd* p, void* stream);

/* All U backward steps of a single-cell decoder with an attention layer (attention_layer_size / --binf_projection,
 * las/model.py:179-200) and / or a monotonic normaliser in ONE launch: one workgroup per utterance walks its chain from the
 * last step to the first, no exchange between workgroups (TF autodiff of dynamic_decode, model_helper.py:415).  `s` is the
 * step struct of las_decoder_step_bwd for step 0 (fused mode; its dctx_* / dh_* fields are set by the launch), per-step
 * pointers advance by the inc_* element counts; s.p / s.align serve the monotonic chain (prev_align = the row before).
 *   A > 0: attention layer of A outputs: d(attention_t) = d_out[b, t, :A] + d(feed)_{t+1}[:A] (saved as bf16 in datt_out
 *          [B, U, A] for d(W_al)), d[query | context] = d(attention_t) W_al^T with the attention layer's kernel [Hd + M, A]
 *          handed in as its LAS_IMAGE_PACK_MFMA_B image waln_packed (dst_rows = Hd + M, dst_cols = A rounded up to 32);
 *   A = 0: the context is output and feed: d(context_t) = d_out[b, t, :M] + d(feed)_{t+1}[:M].
 * d(feed)_t = dz_t kn^T, kn [W0, 4 Hd] (row n = row n of the cell kernel below the token rows), W0 = (A or M) + Hd, handed in
 * as its LAS_IMAGE_PACK_MFMA_B image kn_packed (dst_rows = W0 rounded up to 16, dst_cols = 4 Hd).
 * dfeed_out (nullable) [B, W0] fp32: step 0's, the gradient into the initial feed / state; s.dc ends as dc_{-1}. */
typedef struct las_dec_seq_bwd {
  las_dec_step_bwd s;
  int32_t U, A, W0, reserved;
  int64_t inc_gates, inc_c, inc_align, inc_dz, inc_ds, inc_save, inc_pq;
  const float* d_out;            /* fp32 [B, U, .]: utterance stride ld_dout, step stride inc_dout */
  int64_t ld_dout, inc_dout;
  las_bf16* datt_out;            /* bf16, utterance stride ld_datt, A per step (A > 0) */
  int64_t ld_datt;
  const las_bf16* waln_packed;
  int64_t reserved1;
  const las_bf16* kn_packed;
  int64_t reserved2;
  float* dfeed_out;
  /* A > 0, optional: VW = values W_c [B, T', A] fp32 (W_c: rows [Hd, Hd + M) of the attention layer's kernel; utterance stride
   * ld_vw), formed by the caller once per train step.  With it d(alignments)_t = VW d(attention_t) (T' x A multiply-adds from
   * an LDS copy) instead of values . d(context_t) (a pass over the utterance's T' x M values at every step). */
  const float* vw;
  int64_t ld_vw;
  /* optional: las_decoder_sum_workspace_bytes(B, Hd + 1) bytes, first word zero before the first use (the launch leaves it
   * zero): d(attention_v) [Hd] and d(attention_score_bias) are summed over the steps in the utterance's workgroup and over the
   * utterances in a fixed order, instead of one fp32 atomic per step and utterance */
  void* sum_workspace;
  /* optional (Bahdanau scores, with vw and sum_workspace): las_decoder_seq_xchg_bytes(B, Tm, Hd, M, W0) bytes, zero before the first
   * use.  The launch then runs FOUR workgroups per utterance when 32 * ceil(B / 8) of them find a CU each: three take three
   * quarters of the frames of the query path (tanh recompute, d(keys)) and of the tiles of the two matrix-vector products off
   * the one that walks the chain, exchanging their operands and results through this workspace at every step (A <= 128); sum_workspace must then hold 32 * ceil(B / 8) rows.  Its
   * first 64 bytes are a STICKY status word (bit 5: a bounded wait timed out, results invalid) like the other one-launch kernels'. */
  void* xchg_workspace;
  /* optional, with xchg_workspace: the LAS_IMAGE_PACK_MFMA_B image of the query layer's kernel [Hd(in), Hd(out)] (dh = d(processed
   * query) Wq^T on the matrix cores; without it the product runs from s.wq_t as in the other kernels) */
  const las_bf16* wq_packed;
} las_dec_seq_bwd;
size_t las_decoder_seq_xchg_bytes(int B, int Tm, int Hd, int M, int W0);
/* bytes of the fixed-order sum workspace of the one-launch backward decoders: a counter line + `blocks` rows of n floats */
size_t las_decoder_sum_workspace_bytes(int blocks, int n);
int las_decoder_seq_bwd_supported(int Hd, int M, int A, int W0, int Tm, int attention, int norm);
int las_decoder_seq_bwd(const las_dec_seq_bwd* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stochastic TRAIN-mode pieces (counter-based generator: forward and backward regenerate the same
 * draws; nothing is stored).  DropoutWrapper(input_keep_prob) of las/ops.py:14-18 and the scheduled
 * sampling of utils/training_helper.py:48-87.
 * ---------------------------------------------------------------------------------------- */
/* y = x * m / keep, m ~ Bernoulli(keep) per element (element index r*cols + c). */
int las_dropout_bf16(const las_bf16* x, int64_t ldx, las_bf16* y, int64_t ldy, int rows, int cols, float keep,
                     uint32_t seed, uint32_t stream_id, void* stream);
/* The two directions' copies at once (independent masks for the fw and the bw cell: streams stream0 / stream1, the element
 * indices of las_dropout_bf16); x is read once.  cols and the row strides in multiples of 8. */
int las_dropout_bf16_pair(const las_bf16* x, int64_t ldx, las_bf16* y0, las_bf16* y1, int64_t ldy, int rows, int cols,
                          float keep, uint32_t seed, uint32_t stream0, uint32_t stream1, void* stream);
/* The input masks of U decoder steps in one pass, in place: element (b, t, c), c < cols, at x[b * ldb + t * ldt + c] is scaled by
 * draw (seed, stream0 + t, b * idx_cols + c) (idx_cols >= cols: the width of the masked window these columns lead; 0 = cols)
 * -- U x las_dropout_bf16(rows = B, stream = stream0 + t).  The two-cell one-launch
 * decoder leaves its operand rows undropped; the weight-gradient products (las/model.py:194-200 cells under
 * DropoutWrapper, las/ops.py:14-18) read them dropped.  cols, ldb, ldt: multiples of 8. */
int las_dropout_bf16_steps(las_bf16* x, int64_t ldb, int64_t ldt, int B, int U, int cols, int idx_cols, float keep,
                           uint32_t seed, uint32_t stream0, void* stream);
/* out = a * mask(stream_a)/keep (+ b * mask(stream_b)/keep when b != NULL); contiguous [rows, cols] fp32. */
int las_dropout_bwd(const float* a, const float* b, float* out, int rows, int cols, float keep, uint32_t seed,
                    uint32_t stream_a, uint32_t stream_b, void* stream);
/* out[i] = mask_i / keep: the realised mask, for replaying a run through the oracle. */
int las_dropout_mask(float* out, int64_t total, float keep, uint32_t seed, uint32_t stream_id, void* stream);
/* out[b*U + t, ids[b, t]] = dropout factor of that one-hot entry (1 when keep >= 1); rest 0. */
int las_onehot_bf16(const int32_t* ids, int64_t id_stride_b, int B, int U, int V, las_bf16* out, int64_t ldo,
                    float keep, uint32_t seed, uint32_t stream_id, int feed_width, void* stream);
/* next[b] = token ~ Categorical(logits[b, :V]) with probability prob, else teacher[b] (scheduled sampling). */
int las_sample_tokens(const float* logits, int64_t ldl, int V, const int32_t* teacher, int64_t teacher_stride,
                      int32_t* next, int64_t next_stride, int B, float prob, uint32_t seed, uint32_t step,
                      void* stream);

/* out[r, c] = a[r, c] + b[r, c] * mask / keep, mask drawn at index idx_base + r * idx_ld + c of (seed, stream_id);
 * keep >= 1: plain sum; a == NULL: treated as zero.  (Gradient through the decoder cell's input dropout.) */
int las_add_masked(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int rows, int cols,
                   float keep, uint32_t seed, uint32_t stream_id, uint64_t idx_base, int64_t idx_ld, void* stream);

/* out[i] = N(0,1) draw i of (seed, stream_id): the realised monotonic-attention score noise, for replaying a run
 * through the oracle. */
int las_normal_fill(float* out, int64_t n, uint32_t seed, uint32_t stream_id, void* stream);
/* y = max(x, 0) on bf16 (CustomAttention keys); dx = dy * (y > 0) on fp32 with the bf16 forward output as mask. */
int las_relu_bf16(las_bf16* x, int64_t n, void* stream);
int las_relu_bwd(float* d, const las_bf16* y, int64_t n, void* stream);

/* p[i] += std * N(0,1): the periodic Gaussian weight noise on `*kernel` variables (model_helper.py:418-432). */
int las_add_noise(float* p, int64_t n, float std, uint32_t seed, uint32_t stream_id, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss: tf.contrib.seq2seq.sequence_loss as used by compute_loss (model_helper.py:24-30):
 * loss = sum_{b,t<len_b} CE(logits[b,t], targets[b,t]) / (sum_b len_b + 1e-12).
 * logits [B*U rows, row stride ldl] fp32; targets [B,U] contiguous; loss_out[0] += loss (zero it
 * first); dlogits (bf16, row stride ldd) receives (softmax - onehot) * w / sum(w) * grad_scale, zero
 * for masked steps.
 * ---------------------------------------------------------------------------------------- */
int las_seq_ce_loss(const float* logits, int64_t ldl, const int32_t* targets, const int32_t* target_len,
                    int B, int U, int V, float grad_scale, float* loss_out, las_bf16* dlogits, int64_t ldd,
                    void* stream);

/* Output projection + the loss above + its backward through the projection in ONE launch (round 4; the phone decoder's TRAIN
 * path, las/model.py:251-257 + model_helper.py:24-30): logits [B*U, Vp] fp32 = ctx [B*U, M] (bf16, row stride ld_ctx) wprojT^T +
 * bproj (wprojT [Vp, M]: the projection kernel transposed, zero rows from V on; bproj [Vp]); *loss_out = the sequence loss
 * (STORED, not added: nothing to zero first); dlogits [B*U, Vp] bf16 as las_seq_ce_loss leaves it; dctx [B*U, M] fp32 (row stride
 * ld_dctx) = dlogits wproj^T with wproj [M, Vp] bf16.  targets [B, >= U] int32 with row stride target_stride.  Supported when
 * las_proj_ce_supported(V, Vp, M): Vp a multiple of 16 up to 128, M a multiple of 128.  workspace: las_proj_ce_workspace_bytes
 * bytes, 16-byte aligned, ZERO before the first use (every launch leaves its counter zero); one launch at a time per workspace.
 * The workgroups' partial losses meet in slots of the workspace and are added in a fixed order: *loss_out is bit-reproducible. */
int las_proj_ce_supported(int V, int Vp, int M);
size_t las_proj_ce_workspace_bytes(int B, int U);
int las_proj_ce(const las_bf16* ctx, int64_t ld_ctx, const las_bf16* wprojT, const float* bproj, const las_bf16* wproj,
                const int32_t* targets, int64_t target_stride, const int32_t* target_len, int B, int U, int V, int Vp, int M,
                float grad_scale, float* logits, las_bf16* dlogits, float* dctx, int64_t ld_dctx, float* loss_out,
                void* workspace, void* stream);

/* One step of tf.contrib.seq2seq.BeamSearchDecoder (las/model.py:312-319, length_penalty_weight 0) for B utterances of
 * K beams: logits [B*K, V] fp32 (row stride ldl) of this step; log_probs / finished / lengths [B,K] are the beam state
 * (in: before the step, out: after, already re-ordered); word_ids / parent_ids [B,K] receive the chosen token and the
 * beam it extends (the caller gathers its decoder state with parent_ids and backtracks with gather_tree at the end).
 * Initial state: log_probs = [0, -inf, ...], finished = 0, lengths = 0.  Ties resolve to the lower candidate index. */
int las_beam_step(const float* logits, int64_t ldl, float* log_probs, int32_t* finished, int32_t* lengths,
                  int32_t* word_ids, int32_t* parent_ids, int B, int K, int V, int eos, void* stream);

/* sequence_loss_sigmoid / compute_loss_sigmoid of the sigmoid-output decoder (--binary_outputs without
 * --binf_projection; model_helper.py:81-95,98-130): logits [B*U rows, row stride ldl] fp32 over nf binary features,
 * targets bf16 0/1 rows (row stride ldt), seq_len [B] = the weights' sequence_mask lengths (TRAIN: target lengths; EVAL:
 * max(target, decoded) lengths after the caller padded both to the longer).  loss_out[0] += sum_{b,t<len_b}
 * mean_f BCE(logits, targets) / (sum_b len_b + 1e-12); dlogits (bf16, row stride ldd, may be NULL) its gradient times
 * grad_scale, zero for masked steps. */
int las_seq_sigmoid_loss(const float* logits, int64_t ldl, const las_bf16* targets, int64_t ldt, const int32_t* seq_len,
                         int B, int U, int nf, float grad_scale, float* loss_out, las_bf16* dlogits, int64_t ldd,
                         void* stream);
/* ScheduledSigmoidHelper.sample + next_inputs (utils/training_helper.py:89-119 with binf_to_ipa None, :57-74): utterance
 * b feeds next[b, 0:nf] = Bernoulli(sigmoid(logits[b, f])) draws with probability prob (one draw per utterance and step),
 * else teacher[b, 0:nf] (bf16 0/1 rows; NULL = zeros).  Draws come from the counter-based generator (seed, step). */
int las_sample_features(const float* logits, int64_t ldl, int nf, const las_bf16* teacher, int64_t ldt, las_bf16* next,
                        int64_t ldn, int B, float prob, uint32_t seed, uint32_t step, void* stream);

/* compute_log_probs_loss of the binf_projection decoder (model_helper.py:132-146, :327-331): x [rows, 2*nf] bf16 holds
 * [log p(f=1) | log p(f=0)] per decoder step (ALL rows count, padded steps too).  loss_out += weight * mean(|e^a + e^b
 * - 1| + relu(a) + relu(b)); dx (fp32, row stride ldd, may be NULL) = grad_scale * weight * d(mean)/dx. */
int las_log_probs_loss(const las_bf16* x, int64_t ldx, int rows, int nf, float weight, float grad_scale,
                       float* loss_out, float* dx, int64_t ldd, void* stream);

/* ------------------------------------------------------------------------------------------
 * CTC head (model_helper.py:347-367): tf.nn.ctc_loss_v2 with dense labels [B,U] (row stride ldlab; blank
 * index `blank` = 0 in the reference, labels include the trailing </s>), logits [B,T,ldl] fp32 with C
 * valid classes, per-utterance label_len / logit_len.  loss_out[0] += loss_scale * sum_b nll_b (pass
 * ctc_weight / B); per_example[b] = nll_b (optional); dlogits (bf16, [B,T,ldl]) = grad_scale *
 * d nll_b / d logits (pass ctc_weight / B / replicas).  Workspace: las_ctc_workspace_bytes().
 * ---------------------------------------------------------------------------------------- */
size_t las_ctc_workspace_bytes(int B, int T, int C_padded, int U);
int las_ctc_loss(const float* logits, int64_t ldl, const int32_t* labels, int64_t ldlab, const int32_t* label_len,
                 const int32_t* logit_len, int B, int T, int C, int U, int blank, float loss_scale,
                 float grad_scale, void* workspace, float* loss_out, float* per_example, las_bf16* dlogits,
                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Train op (model_helper.py:403-417): L2 over all variables, per-tensor clip_by_norm(g, 2),
 * tf.train.AdamOptimizer update.  Parameters, gradients and Adam slots are flat fp32 buffers;
 * seg_offsets[nseg+1] gives the tensor boundaries (device int64).
 * ---------------------------------------------------------------------------------------- */

/* out[0] += sum_i x[i]^2 (the L2 regulariser value of model_helper.py:411-413 is scale/2 times this). */
int las_sumsq(const float* x, int64_t n, float* out, void* stream);
/* grads[i] += l2_scale * params[i]; sumsq[s] = ||grads_s||^2  (sumsq zeroed inside).  param_sumsq (nullable):
 * *param_sumsq = sum_i params[i]^2 from the same pass (the value las_sumsq(params) would give).
 * workspace (nullable; las_grad_l2_norms_ws_bytes(nseg, total) bytes, 16-byte aligned, its first 64 bytes ZERO before the
 * first use, left zero by every launch): with it the workgroups' partial sums are added in a fixed order instead of with fp32
 * atomics, so the norms -- and through the clip factors the whole update -- are bit-identical from run to run. */
size_t las_grad_l2_norms_ws_bytes(int nseg, int64_t total);
int las_grad_l2_norms(float* grads, const float* params, const int64_t* seg_offsets, int nseg,
                      int64_t total, float l2_scale, float* sumsq, float* param_sumsq, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The same pass ADDING into sumsq / param_sumsq (nothing is zeroed): after las_train_op_begin, which clears them once per
 * step, the passes over several gradient buckets (and their param_sumsq slots) need no memsets of their own. */
int las_grad_l2_norms_acc(float* grads, const float* params, const int64_t* seg_offsets, int nseg,
                          int64_t total, float l2_scale, float* sumsq, float* param_sumsq, void* workspace,
                          size_t workspace_bytes, void* stream);
/* First launch of the train op: las_status_collect (flag nullable: n = 0) and sumsq[0..nseg) = 0, param_sumsq[0..npsq) = 0. */
int las_train_op_begin(const uint32_t* const* status_words, int n, float* flag, float* sumsq, int nseg,
                       float* param_sumsq, int npsq, void* stream);
/* *out = *audio_loss + half_l2_scale * sum(param_sumsq[0..npsq)): the loss train.py logs (audio loss + L2 term,
 * model_helper.py:411-413) from the sums the norms pass left behind.  audio_loss nullable (0). */
int las_total_loss(const float* audio_loss, const float* param_sumsq, int npsq, float half_l2_scale, float* out,
                   void* stream);
/* grads_s *= clip / max(||grads_s||, clip)  (clip_by_norm, model_helper.py:416). */
int las_grad_clip(float* grads, const int64_t* seg_offsets, int nseg, int64_t total, const float* sumsq,
                  float clip, void* stream);
/* TF-form Adam (epsilon outside the bias correction).  The 1-based update count t is `step`, or
 * *step_dev when step_dev != NULL (a device counter: keeps a captured hipGraph replayable).
 * skip_flag (nullable, device): when *skip_flag != 0 the launch changes nothing (las_status_collect: a persistent
 * kernel of this step reported a timeout, on this replica or -- the flag travels with the all-reduced gradients --
 * on another one; the reference has no such state: tf.train.AdamOptimizer, model_helper.py:404,417). */
int las_adam_update(float* params, float* m, float* v, const float* grads, int64_t total, float lr,
                    float beta1, float beta2, float eps, int step, const int32_t* step_dev, const float* skip_flag,
                    void* stream);
/* *flag = 1.0f when any of the n status words is non-zero, else 0.0f.  status_words: device array of n device
 * pointers, each to the first 32-bit word of a persistent kernel's workspace (las_lstm_recurrent_fwd/bwd,
 * las_decoder_persist_fwd/bwd: bit set = a bounded inter-workgroup wait timed out; the launches never clear it,
 * the host does after reading it). */
int las_status_collect(const uint32_t* const* status_words, int n, float* flag, void* stream);
/* las_grad_clip followed by las_adam_update in one pass over the buffers (single replica: nothing happens between
 * the two); grads holds the clipped gradient afterwards, exactly as after las_grad_clip. */
int las_clip_adam_update(float* params, float* m, float* v, float* grads, const int64_t* seg_offsets, int nseg,
                         int64_t total, const float* sumsq, float clip, float lr, float beta1, float beta2, float eps,
                         int step, const int32_t* step_dev, const float* skip_flag, void* stream);
/* *counter += delta on the stream (tf.train.get_global_step increment, model_helper.py:417) -- unless *skip_flag != 0 (the flag
 * las_adam_update / las_clip_adam_update honour): the Adam
 * step count t of a step whose update was withheld (a persistent kernel timed out on some replica) is not consumed, so
 * the bias corrections of the next applied step are those of an uninterrupted run.  skip_flag may be NULL. */
int las_counter_add_unless(int32_t* counter, int32_t delta, const float* skip_flag, void* stream);

/* ------------------------------------------------------------------------------------------
 * Acoustic front-end (fp32, table driven): preprocess_all.py:69-130 (librosa) and
 * utils/features_utils.py:5-20 (tf.contrib.signal).  Tables (window, DFT twiddles [n_fft,bins], filterbank /
 * DCT matrices [K,N], Savitzky-Golay taps and edge matrices) are built on the host.
 * ---------------------------------------------------------------------------------------- */
/* out[f,k] = |sum_n window[n] x[f*hop + n - pad]  e^{-2 pi i k n / n_fft}|^power, pad = n_fft/2 reflect if center. */
int las_fe_stft(const float* wave, int num_samples, int n_fft, int hop, int center, int power, const float* window,
                const float* costab, const float* sintab, int bins, float* out, int64_t ldo, int frames,
                void* stream);
/* C[M,N] = epilogue(A[M,K] W[K,N]); epilogue 0: none, 1: log(x + eps), 2: 10 log10(max(x, eps)). */
int las_fe_matmul(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int M, int N,
                  int K, int epilogue, float eps, void* stream);
/* x = max(x, max(x) - top_db) over the whole [rows, cols] window (librosa power_to_db top_db); scratch: 1 float. */
int las_fe_top_db(float* x, int64_t ldx, int rows, int cols, float top_db, float* scratch, void* stream);
/* out[f*ldo] = sqrt(mean(frame_f^2)) with centred reflect-padded frames (librosa.feature.rms). */
int las_fe_rms(const float* wave, int num_samples, int frame_length, int hop, float* out, int64_t ldo, int frames,
               void* stream);
/* out[t, c*out_stride + out_offset] = Savitzky-Golay derivative of x[:, c] along t (mode 'interp'); taps == NULL
 * copies x instead (used to interleave [c, dc, ddc] as preprocess_all.py:127-129 does). */
int las_fe_delta(const float* x, int64_t ldx, int T, int F, const float* taps, const float* edge_lo,
                 const float* edge_hi, int width, float* out, int64_t ldo, int out_stride, int out_offset,
                 void* stream);

/* The same pipeline over a BATCH of utterances in three launches (preprocess_all.py:69-130 loops over files; the kernels
 * loop over the frames of all of them): `waves` holds the signals back to back, wave_off[n_utt + 1] (device int64) their
 * first samples, frame_off[n_utt + 1] (device int32) the first output frame of each; outputs are [total_frames, .] with the
 * utterances' frames back to back.  Reflect padding, the top_db floor (under the utterance's OWN maximum) and the
 * Savitzky-Golay edge windows are per utterance; a frame's arithmetic is that of the per-utterance entry points above, in the
 * same order (bit-identical results).
 *   las_fe_batch_melspec: frame -> window -> DFT -> |.|^power -> filterbank mel [bins, n_mels] -> epilogue (0: id, 1:
 *     log(x + eps), 2: 10 log10(max(x, eps))) * scale -> out [total_frames, n_mels]; utt_max [n_utt] (nullable) receives
 *     every utterance's maximum.
 *   las_fe_batch_finish: max(row, utt_max[u] - top_db) (utt_max NULL: as it is) -> DCT dct [n_mels, n_out] (NULL: the first
 *     n_out columns) -> out[f, 0..n_out); energy != 0: out[f, n_out] = RMS of the frame (librosa.feature.rms, center=True).
 *   las_fe_batch_delta: out[f, 3c..3c+2] = x[f, c] and its first / second Savitzky-Golay derivatives along the utterance's
 *     time axis (taps / edge matrices of both orders; every utterance needs >= width frames). */
int las_fe_batch_melspec(const float* waves, const int64_t* wave_off, const int32_t* frame_off, int n_utt, int total_frames,
                         int n_fft, int hop, int center, int power, const float* window, const float* costab,
                         const float* sintab, int bins, const float* mel, int n_mels, int epilogue, float eps, float scale,
                         float* out, float* utt_max, void* stream);
int las_fe_batch_finish(const float* mel_db, int n_mels, const int32_t* frame_off, int n_utt, int total_frames,
                        const float* utt_max, float top_db, const float* dct, int n_out, const float* waves,
                        const int64_t* wave_off, int frame_length, int hop, int energy, float* out, int64_t ldo, void* stream);
int las_fe_batch_delta(const float* x, int64_t ldx, const int32_t* frame_off, int n_utt, int total_frames, int F,
                       const float* taps1, const float* lo1, const float* hi1, const float* taps2, const float* lo2,
                       const float* hi2, int width, float* out, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------
 * Input path (utils/dataset_utils.py:138-283 of the reference: tf.data.TFRecordDataset ->
 * tf.parse_single_sequence_example -> (x - mean) / std -> padded_batch).  Host functions; `data` is the image of a
 * TFRecord file (e.g. an mmap), nothing is copied that the caller does not ask for.
 * ---------------------------------------------------------------------------------------- */
/* Walk the TFRecord framing (uint64 length | masked crc32c | payload | masked crc32c; preprocess_all.py:164-167 writes it
 * with tf.io.TFRecordWriter).  For the first max_records records: payload offset / length and, when the arrays are
 * given, the number of frames ('inputs' float lists), of labels and of label token bytes of the SequenceExample
 * (-1 if the payload is not one).  verify_crc != 0 checks both checksums of every record (as TFRecordDataset does).
 * Returns the number of records in the file (may exceed max_records), or a negative las_status. */
int64_t las_tfrecord_index(const uint8_t* data, size_t nbytes, int verify_crc, int64_t max_records, int64_t* offsets,
                           int64_t* lengths, int32_t* n_frames, int32_t* n_labels, int64_t* label_bytes);
/* Parse n SequenceExamples {'inputs': FixedLenSequenceFeature([num_channels], float32), 'labels':
 * FixedLenSequenceFeature([], string)} (utils/dataset_utils.py:141-153) given by their payload offsets / lengths:
 * frames -> packed rows [sum T, num_channels] with frame_row_offsets [n+1]; labels -> their token bytes back to back
 * with token_offsets [total tokens + 1] and label_counts [n].  A frame of another width is an error, as in TF. */
int las_tfrecord_parse_batch(const uint8_t* data, const int64_t* offsets, const int64_t* lengths, int n, int num_channels,
                             float* frames, int64_t frame_rows_capacity, int64_t* frame_row_offsets, uint8_t* label_bytes,
                             int64_t label_bytes_capacity, int32_t* token_offsets, int64_t token_capacity,
                             int32_t* label_counts);
/* Vocabulary lookup of a batch's label tokens (utils/vocab_utils.py create_vocab_table: index_table_from_tensor with
 * default_value = <unk>): token k = label_bytes[token_offsets[k] .. token_offsets[k+1]) -> ids[k].  The table is open
 * addressing over the tokens' FNV-1a 64-bit hashes (keys[table_size], 0 = empty slot, a hash of 0 is stored as 1;
 * vals[table_size] = ids; table_size a power of two; built once by the caller, phones_las_amd.utils.fast_input). */
int las_vocab_lookup(const uint8_t* label_bytes, const int32_t* token_offsets, int64_t n_tokens, const uint64_t* keys,
                     const int32_t* vals, int64_t table_size, int32_t default_id, int32_t* ids);
/* Device: out[b, t, f] = bf16((frames[row_b + t, f] - mean[f]) / std[f]) for t < T_b, f < num_channels, else 0 -- the
 * normalisation of utils/dataset_utils.py:217-220 (in double, as numpy does with the float64 norm.dmp arrays; mean/std
 * NULL: no normalisation), the bf16 cast and the zero padding to [B, T_padded, F_padded] in one pass over the packed
 * frames of a batch; lengths_out[b] (nullable) = min(T_b, T_padded). */
int las_normalize_pad_bf16(const float* frames, const int64_t* frame_row_offsets, const double* mean, const double* stdv,
                           int num_channels, las_bf16* out, int B, int T_padded, int F_padded, int32_t* lengths_out,
                           void* stream);

/* Host-side CRC-32C (Castagnoli) of a buffer: the checksum of the TFRecord framing the reference's data
 * files use (preprocess_all.py:164-167 tf.io.TFRecordWriter; utils/dataset_utils.py:157 TFRecordDataset). */
uint32_t las_crc32c(const void* data, size_t n);

/* ------------------------------------------------------------------------------------------
 * Data parallelism: the cross-replica gradient SUM of tf.tpu.CrossShardOptimizer (model_helper.py:405-406; train.py:129-140,
 * 157-160) as RCCL all-reduces over xGMI, for hosts that bind this library without torch (the Python host's default transport
 * is torch.distributed, backend "nccl" = the same RCCL).  One process per GPU.  RCCL is resolved at run time (a copy the process
 * already holds is reused; LAS_RCCL_LIB names another): las_dp_available() = 1 when it was found.
 *   rank 0: las_dp_unique_id(id)  -> 128 bytes, handed to every rank by the host's own means (file, socket, environment);
 *   all:    las_dp_init(id, rank, nranks, &comm)   with the rank's HIP device current;
 *   step:   las_dp_allreduce_bucket(comm, grads + begin, count, stream)   in place, fp32, SUM, asynchronous on `stream` -- one call
 *           per exchange bucket, in the same order on every rank (the reference order: each replica scales its loss by 1/N,
 *           clips ITS per-tensor gradients, then the clipped gradients are summed; las_grad_clip before, las_adam_update after);
 *   end:    las_dp_finalize(comm).
 * ---------------------------------------------------------------------------------------- */
typedef struct las_dp_comm las_dp_comm;
int las_dp_available(void);
int las_dp_unique_id(void* id_out_128_bytes);
int las_dp_init(const void* id_128_bytes, int rank, int nranks, las_dp_comm** comm_out);
int las_dp_allreduce_bucket(las_dp_comm* comm, float* grads, int64_t count, void* stream);
int las_dp_finalize(las_dp_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* LAS_HIP_H */
