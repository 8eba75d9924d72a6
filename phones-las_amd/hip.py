"""ctypes binding of liblas_hip.so (include/las_hip.h).  Every wrapper takes torch CUDA tensors,
checks dtype/contiguity, and enqueues on torch's current stream.  Failures raise LasError with the
library's message; a missing library raises at import of this module's ``lib()`` — never a fallback."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get('LAS_HIP_LIB') or os.path.join(_HERE, 'liblas_hip.so')   # LAS_HIP_LIB: diagnostics builds
_lib = None


class LasError(RuntimeError):
    pass


_i32, _i64, _f32, _vp = C.c_int, C.c_int64, C.c_float, C.c_void_p

_SIGS = {
    'las_version': ([], C.c_int),
    'las_last_error': ([], C.c_char_p),
    'las_gemm_nt': ([_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _vp], C.c_int),
    'las_gemm_tn': ([_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _vp], C.c_int),
    'las_gemm_nt_masked': ([_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _f32, C.c_uint32, C.c_uint32, _vp], C.c_int),
    'las_gemm_tn_ws_bytes': ([_i32, _i32, _i32], C.c_size_t),
    'las_gemm_tn_ws': ([_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, C.c_size_t, _vp], C.c_int),
    'las_gemm_tn_store': ([_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _vp], C.c_int),
    'las_cast_bf16': ([_vp, _i64, _i32, _i32, _vp, _i64, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _vp], C.c_int),
    'las_refresh_images': ([_vp, C.c_int, _vp], C.c_int),
    'las_fill_many': ([_vp, C.c_int, _vp], C.c_int),
    'las_stream_delay': ([C.c_int, _vp], C.c_int),
    'las_stream_concurrency_probe': ([_vp, _vp, _vp, C.c_int], C.c_int),
    'las_add_cast_bf16': ([_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp], C.c_int),
    'las_colsum_bf16': ([_vp, _i64, _i32, _i32, _vp, _i32, _vp], C.c_int),
    'las_colsum_ws_bytes': ([_i32, _i32], C.c_size_t),
    'las_colsum_bf16_ws': ([_vp, _i64, _i32, _i32, _vp, _i32, _vp, C.c_size_t, _vp], C.c_int),
    'las_lstm_pack_recurrent': ([_vp, _i32, _vp, _vp], C.c_int),
    'las_lstm_recurrent_fwd': ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp], C.c_int),
    'las_lstm_recurrent_bwd': ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp], C.c_int),
    'las_set_knob': ([C.c_char_p, _i32], C.c_int),
    'las_xcd_histogram': ([_vp, _i32, _i32, _i32, _vp], C.c_int),
    'las_lstm_fused_input_chunks': ([_i32, _i32], C.c_int),
    'las_lstm_pack_input': ([_vp, _i32, _i32, _i32, _vp, _vp], C.c_int),
    'las_lstm_recurrent_fwd_ex': ([_vp, _vp], C.c_int),
    'las_gemm_nt_stream_supported': ([_i32, _i32, _i32], C.c_int),
    'las_gemm_nt_stream_flags': ([_i32, _i32, _i32, _i32], C.c_size_t),
    'las_pack_mfma_b_bf16': ([_vp, _i64, _i32, _i32, _vp, _vp], C.c_int),
    'las_gemm_nt_bimg': ([_vp, _i64, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp], C.c_int),
    'las_gemm_nt_stream_dirs': ([_vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp], C.c_int),
    'las_lstm_workspace_bytes': ([_i32, _i32, _i32], C.c_size_t),
    'las_lstm_slice_rows': ([_i32, _i32, _i32], C.c_int),
    'las_lstm_fwd_workgroups': ([_i32, _i32, _i32], C.c_int),
    'las_pyramid_lengths': ([_vp, _vp, _i32, _vp], C.c_int),
    'las_pyramid_lengths_multi': ([_vp, _vp, _i32, _i32, _vp], C.c_int),
    'las_decoder_step_fwd': ([_vp, _i32, _vp], C.c_int),
    'las_decoder_step_bwd': ([_vp, _vp], C.c_int),
    'las_decoder_seq_bwd_supported': ([_i32] * 7, C.c_int),
    'las_decoder_sum_workspace_bytes': ([_i32, _i32], C.c_size_t),
    'las_decoder_seq_xchg_bytes': ([_i32] * 5, C.c_size_t),
    'las_decoder_seq_bwd': ([_vp, _vp], C.c_int),
    'las_seq_ce_loss': ([_vp, _i64, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _i64, _vp], C.c_int),
    'las_proj_ce_supported': ([_i32, _i32, _i32], C.c_int),
    'las_proj_ce_workspace_bytes': ([_i32, _i32], C.c_size_t),
    'las_proj_ce': ([_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _i64, _vp, _vp, _vp], C.c_int),
    'las_seq_sigmoid_loss': ([_vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _i64, _vp], C.c_int),
    'las_sample_features': ([_vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _f32, C.c_uint32, C.c_uint32, _vp], C.c_int),
    'las_grad_l2_norms_ws_bytes': ([_i32, _i64], C.c_size_t),
    'las_grad_l2_norms': ([_vp, _vp, _vp, _i32, _i64, _f32, _vp, _vp, _vp, C.c_size_t, _vp], C.c_int),
    'las_clip_adam_update': ([_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp, _vp], C.c_int),
    'las_grad_clip': ([_vp, _vp, _i32, _i64, _vp, _f32, _vp], C.c_int),
    'las_adam_update': ([_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i32, _vp, _vp, _vp], C.c_int),
    'las_status_collect': ([_vp, _i32, _vp, _vp], C.c_int),
    'las_grad_l2_norms_acc': ([_vp, _vp, _vp, _i32, _i64, _f32, _vp, _vp, _vp, C.c_size_t, _vp], C.c_int),
    'las_train_op_begin': ([_vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp], C.c_int),
    'las_total_loss': ([_vp, _vp, _i32, _f32, _vp, _vp], C.c_int),
    'las_counter_add_unless': ([_vp, _i32, _vp, _vp], C.c_int),
    'las_sumsq': ([_vp, _i64, _vp, _vp], C.c_int),
    'las_crc32c': ([C.c_char_p, C.c_size_t], C.c_uint32),
    'las_dp_available': ([], C.c_int),
    'las_dp_unique_id': ([_vp], C.c_int),
    'las_dp_init': ([_vp, _i32, _i32, _vp], C.c_int),
    'las_dp_allreduce_bucket': ([_vp, _vp, _i64, _vp], C.c_int),
    'las_dp_finalize': ([_vp], C.c_int),
    'las_tfrecord_index': ([_vp, C.c_size_t, _i32, _i64, _vp, _vp, _vp, _vp, _vp], C.c_int64),
    'las_tfrecord_parse_batch': ([_vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp], C.c_int),
    'las_vocab_lookup': ([_vp, _vp, _i64, _vp, _vp, _i64, _i32, _vp], C.c_int),
    'las_normalize_pad_bf16': ([_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp], C.c_int),
    'las_dropout_bf16': ([_vp, _i64, _vp, _i64, _i32, _i32, _f32, C.c_uint32, C.c_uint32, _vp], C.c_int),
    'las_dropout_bf16_pair': ([_vp, _i64, _vp, _vp, _i64, _i32, _i32, _f32, C.c_uint32, C.c_uint32, C.c_uint32, _vp], C.c_int),
    'las_dropout_bf16_steps': ([_vp, _i64, _i64, _i32, _i32, _i32, _i32, _f32, C.c_uint32, C.c_uint32, _vp], C.c_int),
    'las_dropout_bwd': ([_vp, _vp, _vp, _i32, _i32, _f32, C.c_uint32, C.c_uint32, C.c_uint32, _vp], C.c_int),
    'las_dropout_mask': ([_vp, _i64, _f32, C.c_uint32, C.c_uint32, _vp], C.c_int),
    'las_onehot_bf16': ([_vp, _i64, _i32, _i32, _i32, _vp, _i64, _f32, C.c_uint32, C.c_uint32, _i32, _vp], C.c_int),
    'las_ctc_workspace_bytes': ([_i32, _i32, _i32, _i32], C.c_size_t),
    'las_ctc_loss': ([_vp, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp], C.c_int),
    'las_fe_stft': ([_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp], C.c_int),
    'las_fe_matmul': ([_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _vp], C.c_int),
    'las_fe_top_db': ([_vp, _i64, _i32, _i32, _f32, _vp, _vp], C.c_int),
    'las_fe_rms': ([_vp, _i32, _i32, _i32, _vp, _i64, _i32, _vp], C.c_int),
    'las_fe_delta': ([_vp, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _vp], C.c_int),
    'las_fe_batch_melspec': ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _f32, _f32, _vp, _vp, _vp], C.c_int),
    'las_fe_batch_finish': ([_vp, _i32, _vp, _i32, _i32, _vp, _f32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _vp], C.c_int),
    'las_fe_batch_delta': ([_vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp], C.c_int),
    'las_add_masked': ([_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _f32, C.c_uint32, C.c_uint32, C.c_uint64, _i64, _vp], C.c_int),
    'las_add_noise': ([_vp, _i64, _f32, C.c_uint32, C.c_uint32, _vp], C.c_int),
    'las_gemm_tn_lstm_workspace_bytes': ([C.c_int, C.c_int, C.c_int], C.c_size_t),
    'las_gemm_tn_lstm': ([_vp, _i64, C.c_int, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp, _i64, _vp, _vp, C.c_int, C.c_int, _vp, _vp], C.c_int),
    'las_decoder_persist_supported': ([C.c_int] * 5, C.c_int),
    'las_decoder_persist_al_supported': ([_i32, _i32, _i32, _i32, _i32, _i32], C.c_int),
    'las_decoder_persist2_supported': ([_i32, _i32, _i32, _i32, _i32, _i32], C.c_int),
    'las_decoder_persist_workspace_bytes': ([C.c_int, C.c_int, C.c_int, C.c_int], C.c_size_t),
    'las_decoder_persist_max_batch': ([], C.c_int),
    'las_decoder_persist_fwd': ([_vp, _vp], C.c_int),
    'las_decoder_persist_bwd_supported': ([C.c_int] * 5, C.c_int),
    'las_decoder_persist2_bwd_supported': ([_i32, _i32, _i32, _i32, _i32, _i32], C.c_int),
    'las_decoder_persist_bwd': ([_vp, _vp], C.c_int),
    'las_beam_step': ([_vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp], C.c_int),
    'las_log_probs_loss': ([_vp, _i64, C.c_int, C.c_int, _f32, _f32, _vp, _vp, _i64, _vp], C.c_int),
    'las_normal_fill': ([_vp, _i64, C.c_uint32, C.c_uint32, _vp], C.c_int),
    'las_relu_bf16': ([_vp, _i64, _vp], C.c_int),
    'las_relu_bwd': ([_vp, _vp, _i64, _vp], C.c_int),
    'las_sample_tokens': ([_vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _f32, C.c_uint32, C.c_uint32, _vp], C.c_int),
}

# entry points declared in include/las_hip.h whose kernels are not written yet (shrinks to empty)
_PENDING = set()
EXPORTS = tuple(n for n in _SIGS if n not in _PENDING)


class LstmFwd(C.Structure):
    """struct las_lstm_fwd (include/las_hip.h)."""
    _fields_ = [('xproj', _vp), ('wpacked', _vp), ('length', _vp), ('y', _vp), ('cbuf', _vp), ('c_last', _vp), ('h_last', _vp),
                ('workspace', _vp), ('B', _i32), ('T', _i32), ('H', _i32), ('ndir', _i32),
                ('x', _vp), ('ldx', _i64), ('x_dir_stride', _i64), ('Dp', _i32), ('reserved0', _i32), ('kx_packed', _vp), ('bias', _vp),
                ('ready', _vp), ('ready_count', _i32), ('rows_per_slice', _i32)]


class DecStep(C.Structure):
    """struct las_dec_step (include/las_hip.h)."""
    _fields_ = [('B', _i32), ('Hd', _i32), ('M', _i32), ('Tm', _i32), ('attention', _i32), ('mode', _i32),
                ('z', _vp), ('tok_rows', _vp), ('tok_ids', _vp), ('tok_stride', _i64), ('bias', _vp),
                ('c_prev', _vp), ('ldcp', _i64), ('gates_out', _vp), ('ldg', _i64), ('c_out', _vp), ('ldco', _i64),
                ('h_out', _vp), ('ldh', _i64), ('h_out2', _vp), ('ldh2', _i64), ('keys', _vp), ('values', _vp),
                ('mem_len', _vp), ('wq', _vp), ('att_v', _vp), ('align_out', _vp), ('align_bf16', _vp),
                ('lda', _i64), ('pq_out', _vp), ('ldpq', _i64), ('ctx_out', _vp), ('ldc', _i64),
                ('ctx_out2', _vp), ('ldc2', _i64), ('drop_keep', _f32), ('drop_seed', C.c_uint32),
                ('drop_stream', C.c_uint32), ('step', _i32), ('feed_width', _i32), ('feed_plain', _i32), ('query', _vp), ('ldq', _i64),
                ('norm', _i32), ('score_bias', _vp), ('prev_align', _vp), ('ldpa', _i64), ('p_out', _vp), ('ldp', _i64),
                ('noise_scale', _f32), ('noise_seed', C.c_uint32), ('noise_stream', C.c_uint32), ('feed_stream0', C.c_uint32)]


class DecPersist(C.Structure):
    """struct las_dec_persist (include/las_hip.h)."""
    _fields_ = [('s', DecStep), ('U', _i32), ('K_in', _i32)] + [(n, _i64) for n in (
        'inc_tok', 'inc_cprev', 'inc_gates', 'inc_cout', 'inc_h', 'inc_h2', 'inc_align', 'inc_pq', 'inc_ctx', 'inc_ctx2')] + [
        ('x', _vp), ('ldx', _i64), ('inc_x', _i64), ('kT', _vp), ('ldk', _i64), ('wq_packed', _vp), ('sc_all', _vp), ('ld_sc', _i64), ('workspace', _vp),
        ('sampling_prob', _f32), ('seed', C.c_uint32), ('teacher', _vp), ('teacher_stride', _i64), ('wprojT', _vp), ('ldw', _i64),
        ('bproj', _vp), ('logits', _vp), ('ld_logits', _i64), ('plog', _vp), ('V', _i32), ('Vp', _i32),
        ('walT', _vp), ('ld_wal', _i64), ('A', _i32), ('x_att_off', _i32), ('att_out', _vp), ('ld_att', _i64), ('inc_p', _i64),
        ('k1T', _vp), ('ldk1', _i64), ('K1_in', _i32), ('wiring', _i32), ('bias1', _vp), ('c1', _vp), ('gates1', _vp), ('h1', _vp),
        ('win0', _i32), ('win1', _i32), ('in_stream0', C.c_uint32), ('in_stream1', C.c_uint32),
        ('emb', _vp), ('ld_emb', _i64), ('T0', _i32), ('reserved_t0', _i32)]


class DecStepBwd(C.Structure):
    """struct las_dec_step_bwd (include/las_hip.h)."""
    _fields_ = [('B', _i32), ('Hd', _i32), ('M', _i32), ('Tm', _i32), ('attention', _i32), ('mode', _i32),
                ('dctx_a', _vp), ('ldda', _i64), ('dctx_b', _vp), ('lddb', _i64), ('dctx_save', _vp), ('ldds', _i64),
                ('dh_rec', _vp), ('ldr', _i64), ('dc', _vp), ('gates', _vp), ('ldg', _i64), ('c_new', _vp),
                ('ldcn', _i64), ('c_prev', _vp), ('ldcp', _i64), ('align', _vp), ('lda', _i64), ('pq', _vp),
                ('ldpq', _i64), ('keys', _vp), ('values', _vp), ('mem_len', _vp), ('wq_t', _vp), ('att_v', _vp),
                ('dz', _vp), ('ldz', _i64), ('ds_out', _vp), ('ldso', _i64), ('dkeys_acc', _vp), ('dv_acc', _vp),
                ('dpq_out', _vp), ('lddpq', _i64), ('drop_keep', _f32), ('drop_seed', C.c_uint32),
                ('drop_stream', C.c_uint32), ('step', _i32), ('feed_width', _i32), ('dq_out', _vp), ('lddq', _i64),
                ('dh_b', _vp), ('ldhb', _i64), ('dh_c', _vp), ('ldhc', _i64),
                ('norm', _i32), ('p', _vp), ('ldp', _i64), ('prev_align', _vp), ('ldpa', _i64), ('dalign_carry', _vp),
                ('ldcarry', _i64), ('dbias_acc', _vp)]


class DecSeqBwd(C.Structure):
    """struct las_dec_seq_bwd (include/las_hip.h)."""
    _fields_ = [('s', DecStepBwd), ('U', _i32), ('A', _i32), ('W0', _i32), ('reserved', _i32)] + [(n, _i64) for n in (
        'inc_gates', 'inc_c', 'inc_align', 'inc_dz', 'inc_ds', 'inc_save', 'inc_pq')] + [
        ('d_out', _vp), ('ld_dout', _i64), ('inc_dout', _i64), ('datt_out', _vp), ('ld_datt', _i64), ('waln_packed', _vp), ('reserved1', _i64),
        ('kn_packed', _vp), ('reserved2', _i64), ('dfeed_out', _vp), ('vw', _vp), ('ld_vw', _i64), ('sum_workspace', _vp), ('xchg_workspace', _vp), ('wq_packed', _vp)]


class DecPersistBwd(C.Structure):
    """struct las_dec_persist_bwd (include/las_hip.h)."""
    _fields_ = [('s', DecStepBwd), ('U', _i32), ('W', _i32)] + [(n, _i64) for n in (
        'inc_a', 'inc_save', 'inc_gates', 'inc_c', 'inc_align', 'inc_dz', 'inc_ds', 'inc_pq')] + [
        ('kc', _vp), ('ldk', _i64), ('dfeed_all', _vp), ('sum_workspace', _vp), ('dhp_all', _vp), ('workspace', _vp),
        ('k1c', _vp), ('ldk1', _i64), ('W1', _i32), ('wiring', _i32), ('gates1', _vp), ('c1', _vp), ('dz1', _vp), ('dc1', _vp),
        ('dfeed1_all', _vp), ('d_out1', _vp), ('ld_dout1', _i64), ('inc_dout1', _i64), ('win0', _i32), ('win1', _i32),
        ('in_stream0', C.c_uint32), ('in_stream1', C.c_uint32)]


ATT_LUONG, ATT_BAHDANAU, ATT_CUSTOM, ATT_LUONG_MONOTONIC, ATT_BAHDANAU_MONOTONIC = 0, 1, 2, 3, 4
NORM_SOFTMAX, NORM_MONOTONIC_PARALLEL, NORM_MONOTONIC_HARD = 0, 1, 2
ATT_ADDITIVE = (ATT_BAHDANAU, ATT_BAHDANAU_MONOTONIC)
ATT_USES_WQ = (ATT_BAHDANAU, ATT_BAHDANAU_MONOTONIC, ATT_CUSTOM)
ATT_MONOTONIC = (ATT_LUONG_MONOTONIC, ATT_BAHDANAU_MONOTONIC)
DEC_FUSED, DEC_CELL_ONLY, DEC_ATTENTION_ONLY = 0, 1, 2


# ---------------------------------------------------------------------------------------------
# optional per-kernel timing with HIP events on the launching stream (bench.py's roofline table)
# ---------------------------------------------------------------------------------------------
_prof = None


class KernelTimer:
    """with KernelTimer() as kt: ... enqueue steps ...; kt.table() -> {family: (launches, ms, flops)} after a device
    synchronise.  Each timed launch is bracketed by two events on the stream it is launched on (torch's current stream
    at the call), so launches on the second stream are timed on that stream."""

    def __enter__(self):
        global _prof
        self.records = []
        self.outer, _prof = _prof, self
        return self

    def __exit__(self, *exc):
        global _prof
        _prof = self.outer
        return False

    def table(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, e0, e1 in self.records:
            n, ms, fl = out.get(name, (0, 0.0, 0.0))
            out[name] = (n + 1, ms + e0.elapsed_time(e1), fl + flops)
        return out


def prof_begin(name, flops):
    if _prof is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return (name, float(flops), e0)


def prof_end(tok):
    if tok is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        _prof.records.append(tok + (e1,))


def addr(t, offset_elems=0):
    """Raw device address (int) of tensor ``t`` advanced by ``offset_elems`` elements; None -> 0."""
    if t is None:
        return 0
    if not t.is_cuda:
        raise LasError('expected a CUDA tensor; the LAS ops have no CPU path')
    return t.data_ptr() + offset_elems * t.element_size()


def lib_path():
    return _LIB_PATH


def lib():
    """Load liblas_hip.so (built by phones-las_amd/build.py).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise LasError('%s not found: run `python phones-las_amd/build.py` (hipcc, gfx950); '
                           'there is no CPU fallback' % _LIB_PATH)
        l = C.CDLL(_LIB_PATH)
        for name, (args, res) in _SIGS.items():
            if name in _PENDING:
                continue
            fn = getattr(l, name)          # AttributeError here = the .so is stale: rebuild
            fn.argtypes = args
            fn.restype = res
        _lib = l
    return _lib


def set_knob(name, value):
    """Override one of the library's LAS_* integer switches (it reads the environment only once): the tests' hook."""
    check(lib().las_set_knob(name.encode(), int(value)))


def check(rc):
    if rc != 0:
        raise LasError('liblas_hip error %d: %s' % (rc, lib().las_last_error().decode()))


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def p(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise LasError('expected a CUDA tensor; the LAS ops have no CPU path')
    return C.c_void_p(t.data_ptr())


def _req(t, dtype, name):
    if t.dtype != dtype:
        raise LasError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise LasError('%s must be contiguous' % name)


# ---------------------------------------------------------------------------------------------
# thin typed wrappers
# ---------------------------------------------------------------------------------------------
def gemm_nt(A, B, C_, M, N, K, lda=None, ldb=None, ldc=None, bias=None, out_bf16=False, accumulate=False,
            batch=1, sa=0, sb=0, sc=0, split_k=1):
    lda = lda if lda is not None else A.stride(-2)
    ldb = ldb if ldb is not None else B.stride(-2)
    ldc = ldc if ldc is not None else C_.stride(-2)
    tok = prof_begin('gemm_nt', 2.0 * M * N * K * batch)
    check(lib().las_gemm_nt(p(A), lda, p(B), ldb, p(C_), ldc, p(bias), M, N, K, int(out_bf16), int(accumulate),
                            batch, sa, sb, sc, split_k, stream()))
    prof_end(tok)


_TN_ATOMIC = os.environ.get('LAS_TN_ATOMIC', '0') == '1'      # diagnostics / A-B timing: split K with fp32 atomics (round 1-2)
_TN_WS = {}


_ws_lane = 0


class ws_lane:
    """with ws_lane(k): products issued inside take workspace k.  las.ops.Overlap.fork() enters lane 1 / 2 together with its
    side streams, so that products running at the same time on different streams never share a workspace (a key taken from
    the stream itself would change under graph capture, which runs on a stream of its own)."""

    def __init__(self, lane):
        self.lane = lane

    def __enter__(self):
        global _ws_lane
        self.prev, _ws_lane = _ws_lane, self.lane

    def __exit__(self, *exc):
        global _ws_lane
        _ws_lane = self.prev
        return False


def _tn_workspace(nbytes):
    """The K-slice workspace of las_gemm_tn_ws for the current lane (ws_lane: 0 = the main stream, 1 / 2 = the side streams
    of the backward pass; products of one lane run in stream order and share it).  Grows on demand; growing inside a graph
    capture is refused (run the step once eagerly first, as every captured flow here does)."""
    if nbytes == 0:
        return None
    key = (torch.cuda.current_device(), _ws_lane)
    ws = _TN_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise LasError('gemm_tn: new weight-gradient workspace during graph capture (run the step once eagerly first)')
        ws = _TN_WS[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device='cuda')
    return ws


def gemm_tn(A, B, C_, M, N, K, lda=None, ldb=None, ldc=None, a_shift=0, period=0, batch=1, sa=0, sb=0, sc=0,
            split_k=1, c_perm_h=0, store=False):
    """C += A^T B (K slices through a workspace, summed in a fixed order; batched products: fp32 atomics, zero C first), or
    with store=True C = A^T B (no zeroing, K not split; C fp32 or bf16)."""
    lda = lda if lda is not None else A.stride(-2)
    ldb = ldb if ldb is not None else B.stride(-2)
    ldc = ldc if ldc is not None else C_.stride(-2)
    tok = prof_begin('gemm_tn', 2.0 * M * N * K * batch)
    if store:
        check(lib().las_gemm_tn_store(p(A), lda, p(B), ldb, p(C_), ldc, M, N, K, a_shift, period, c_perm_h, batch, sa, sb,
                                      sc, int(C_.dtype == torch.bfloat16), stream()))
    elif batch == 1 and not _TN_ATOMIC:
        # split K through a workspace + fixed-order reduce: no fp32 atomics, bit-identical from run to run
        ws = _tn_workspace(lib().las_gemm_tn_ws_bytes(M, N, split_k))
        check(lib().las_gemm_tn_ws(p(A), lda, p(B), ldb, p(C_), ldc, M, N, K, a_shift, period, c_perm_h, split_k,
                                   p(ws), ws.numel() if ws is not None else 0, stream()))
    else:
        check(lib().las_gemm_tn(p(A), lda, p(B), ldb, p(C_), ldc, M, N, K, a_shift, period, c_perm_h, batch, sa, sb, sc,
                                split_k, stream()))
    prof_end(tok)


class ImageJob(C.Structure):
    """las_image_job of include/las_hip.h."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('lds', C.c_int64), ('ldd', C.c_int64),
                ('rows', C.c_int32), ('cols', C.c_int32), ('dst_rows', C.c_int32), ('dst_cols', C.c_int32),
                ('transpose', C.c_int32), ('perm_h', C.c_int32), ('kind', C.c_int32), ('reserved', C.c_int32)]


class FillJob(C.Structure):
    """las_fill_job of include/las_hip.h."""
    _fields_ = [('dst', C.c_void_p), ('src', C.c_void_p), ('rows', C.c_int64), ('cols', C.c_int64), ('ldd', C.c_int64),
                ('lds', C.c_int64), ('kind', C.c_int32), ('reserved', C.c_int32)]


FILL_MAX_JOBS = 12


def _window(t):
    """(rows, cols, row stride in elements) of a 1-D / 2-D tensor view with unit inner stride (or any contiguous tensor)."""
    if t.is_contiguous():
        return 1, t.numel(), t.numel()
    if t.dim() == 2 and t.stride(1) == 1:
        return t.shape[0], t.shape[1], t.stride(0)
    raise LasError('fill_many: expected a contiguous tensor or a 2-D view with unit inner stride, got shape %s stride %s'
                   % (tuple(t.shape), tuple(t.stride())))


def fill_many(zero=(), copy=()):
    """One launch (las_fill_many) instead of a torch kernel per tensor: `zero` -- tensors / 2-D views (2- or 4-byte
    elements) to clear; `copy` -- (dst, src) pairs, src fp32 (or None = zeros), dst fp32 or bf16, same shape."""
    jobs = []
    for t in zero:
        if t.numel() == 0:
            continue
        if not t.is_cuda or t.element_size() not in (2, 4):
            raise LasError('fill_many: CUDA tensors of 2- or 4-byte elements only')
        r, c, ld = _window(t)
        jobs.append(FillJob(t.data_ptr(), None, r, c, ld, 0, 0 if t.element_size() == 4 else 1, 0))
    for dst, src in copy:
        if dst.numel() == 0:
            continue
        if src is not None and (src.dtype != torch.float32 or tuple(src.shape) != tuple(dst.shape)):
            raise LasError('fill_many: copy sources are fp32 tensors of the destination\'s shape')
        if dst.dtype not in (torch.float32, torch.bfloat16):
            raise LasError('fill_many: copy destinations are fp32 or bf16')
        r, c, ld = _window(dst)
        lds = 0
        if src is not None:
            rs, cs, lds = _window(src)
            if (rs, cs) != (r, c):
                if src.is_contiguous() and r * c == rs * cs:      # a contiguous source read through the destination's window
                    lds = c
                elif dst.is_contiguous() and r * c == rs * cs:    # a contiguous destination written through the source's
                    r, c, ld = rs, cs, cs
                else:
                    raise LasError('fill_many: source and destination windows differ')
        jobs.append(FillJob(dst.data_ptr(), src.data_ptr() if src is not None else None, r, c, ld, lds,
                            2 if dst.dtype == torch.float32 else 3, 0))
    for i in range(0, len(jobs), FILL_MAX_JOBS):
        part = jobs[i:i + FILL_MAX_JOBS]
        arr = (FillJob * len(part))(*part)
        check(lib().las_fill_many(arr, len(part), stream()))


IMAGE_CAST, IMAGE_PACK_RECURRENT, IMAGE_BIAS_INTERLEAVE, IMAGE_COPY_F32, IMAGE_PACK_MFMA_B, IMAGE_PACK_INPUT = 0, 1, 2, 3, 4, 5
_image_batch = None
_image_tables = {}        # bytes of a job table -> its device copy (tables repeat every step: uploaded once)


class image_batch:
    """Collects the image rebuilds issued inside the block (cast_bf16 with batch 1, pack_recurrent, bias_interleave,
    copy_f32) and runs them as ONE las_refresh_images launch at exit.  The jobs must be independent of each other and of
    anything else enqueued inside the block.  A table seen before is not uploaded again, so a block whose jobs repeat
    (same tensors every step) can be captured into a HIP graph after one eager run."""

    def __enter__(self):
        global _image_batch
        self.outer, _image_batch = _image_batch, self
        self.jobs = []
        return self

    def __exit__(self, exc_type, exc, tb):
        global _image_batch
        _image_batch = self.outer
        if exc_type is None and self.jobs:
            if self.outer is not None:
                self.outer.jobs.extend(self.jobs)
            else:
                _run_image_jobs(self.jobs)
        return False


def _run_image_jobs(jobs):
    arr = (ImageJob * len(jobs))(*jobs)
    key = bytes(arr)
    table = _image_tables.get(key)
    if table is None:
        if torch.cuda.is_current_stream_capturing():
            raise LasError('image_batch: new job table during graph capture (run the block once eagerly first)')
        # (the tables are NEVER freed: a captured HIP graph replays its refresh launch with the table's address baked in.  Until round 6
        #  the cache was emptied at 256 entries; in a process that had built that many different models -- the GPU test suite, once
        #  the weight images of las_gemm_nt_bimg added tables -- a later table or tensor took a freed table's memory, and the next graph
        #  replay walked garbage jobs: HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in tests/test_gpu_step_forms.py.  A table is 48 bytes
        #  per image.)
        table = torch.frombuffer(bytearray(key), dtype=torch.uint8).cuda()
        _image_tables[key] = table
    check(lib().las_refresh_images(p(table), len(jobs), stream()))


def _image_job(kind, src, dst, lds=0, ldd=0, rows=0, cols=0, dst_rows=0, dst_cols=0, transpose=0, perm_h=0):
    if not (src.is_cuda and dst.is_cuda):
        raise LasError('expected CUDA tensors; the LAS ops have no CPU path')
    job = ImageJob(src.data_ptr(), dst.data_ptr(), lds, ldd, rows, cols, dst_rows, dst_cols, int(transpose), perm_h, kind, 0)
    if _image_batch is not None:
        _image_batch.jobs.append(job)
    else:
        _run_image_jobs([job])


def cast_bf16(src, rows, cols, dst, dst_rows, dst_cols, ldd=None, transpose=False, lds=None, batch=1, sbs=0, dbs=0,
              perm_h=0):
    """dst window [dst_rows, dst_cols] (row stride ldd) = bf16(src[rows, cols]) (transposed if asked), zero padded."""
    lds = lds if lds is not None else (src.stride(-2) if src.dim() >= 2 else cols)
    ldd = ldd if ldd is not None else (dst.stride(-2) if dst.dim() >= 2 else dst_cols)
    if _image_batch is not None and batch == 1:
        _image_job(IMAGE_CAST, src, dst, lds, ldd, rows, cols, dst_rows, dst_cols, transpose, perm_h)
        return
    check(lib().las_cast_bf16(p(src), lds, rows, cols, p(dst), ldd, dst_rows, dst_cols, int(transpose), batch,
                              sbs, dbs, perm_h, stream()))


def pack_recurrent(kernel_h, H, packed):
    """las_lstm_pack_recurrent (joins an open image_batch)."""
    if _image_batch is not None:
        _image_job(IMAGE_PACK_RECURRENT, kernel_h, packed, rows=H)
        return
    check(lib().las_lstm_pack_recurrent(p(kernel_h), H, p(packed), stream()))


def pack_input(kernel, D, H, chunks, packed):
    """las_lstm_pack_input (joins an open image_batch): the K_x image of the fused input projection."""
    if packed.numel() != (H // 16) * chunks * 4 * 512:
        raise LasError('pack_input: destination of %d elements, expected %d' % (packed.numel(), (H // 16) * chunks * 4 * 512))
    if _image_batch is not None:
        _image_job(IMAGE_PACK_INPUT, kernel, packed, rows=D, cols=H, dst_rows=chunks)
        return
    check(lib().las_lstm_pack_input(p(kernel), D, H, chunks, p(packed), stream()))


def pack_mfma_b(src, rows, cols, dst, lds=None, transpose=False, perm_h=0, image_k=0, k0=0, dst_rows=None, dst_cols=None):
    """dst = the LAS_IMAGE_PACK_MFMA_B image of fp32 src[rows, cols] (joins an open image_batch): dst holds
    ceil(rows / 16) * 16 x ceil(cols / 32) * 32 bf16 elements in B-fragment order.  perm_h = H: the source's column axis through the
    gate interleaving; image_k / k0: the window is the K range [k0, k0 + dst_cols) of an image whose whole K is image_k (dst = the
    whole image then)."""
    dr = dst_rows if dst_rows is not None else -(-rows // 16) * 16
    dc = dst_cols if dst_cols is not None else -(-cols // 32) * 32
    if dst.numel() != dr * (image_k or dc):
        raise LasError('pack_mfma_b: destination of %d elements, expected %d' % (dst.numel(), dr * (image_k or dc)))
    job = ImageJob(src.data_ptr(), dst.data_ptr(), lds if lds is not None else src.stride(-2), image_k, rows, cols, dr, dc, int(transpose),
                   perm_h, IMAGE_PACK_MFMA_B, k0)
    if _image_batch is not None:
        _image_batch.jobs.append(job)
    else:
        _run_image_jobs([job])


def gemm_nt_bimg(A, image, C_, M, N, K, lda=None, ldc=None, bias=None, accumulate=False):
    """C_ [M, N] fp32 (=|+=) A [M, K] W^T + bias, W as its LAS_IMAGE_PACK_MFMA_B image (las_gemm_nt_bimg)."""
    lda = lda if lda is not None else A.stride(-2)
    ldc = ldc if ldc is not None else C_.stride(-2)
    tok = prof_begin('gemm_nt', 2.0 * M * N * K)
    check(lib().las_gemm_nt_bimg(p(A), lda, p(image), p(C_), ldc, p(bias), M, N, K, int(accumulate), stream()))
    prof_end(tok)


def gemm_nt_bimg_wanted(M, N, K):
    """Shapes the image form takes; it measured faster than the ring kernels on every bulk shape of the steps (profiles/r06_gemm_vs_lib.txt)."""
    return BIMG and M >= 1024 and N % 256 == 0 and K % 128 == 0 and K >= 512


BIMG = os.environ.get('LAS_GEMM_BIMG', '1') != '0'       # (0: the ring kernels everywhere -- A/B timing)


def bias_interleave(bias, H, dst):
    """fp32 dst[u*4+g] = bias[g*H+u]: the LSTM bias in the gate-interleaved column order of the recurrent kernels."""
    _image_job(IMAGE_BIAS_INTERLEAVE, bias, dst, rows=H)


def copy_f32(src, n, dst):
    """fp32 dst[0:n] = src[0:n] (joins an open image_batch)."""
    _image_job(IMAGE_COPY_F32, src, dst, cols=n)


_CS_WS = {}


def colsum_bf16(X, M, N, out, ldx=None, perm_h=0):
    """out[n] += sum_m X[m, n] (bias gradients), bit-identical from run to run: the row chunks meet in a per-lane workspace
    (ws_lane) instead of fp32 atomics."""
    ldx = ldx if ldx is not None else X.stride(-2)
    if _TN_ATOMIC:
        check(lib().las_colsum_bf16(p(X), ldx, M, N, p(out), perm_h, stream()))
        return
    need = lib().las_colsum_ws_bytes(M, N)
    key = (torch.cuda.current_device(), _ws_lane)
    ws = _CS_WS.get(key)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise LasError('colsum_bf16: new workspace during graph capture (run the step once eagerly first)')
        ws = _CS_WS[key] = torch.zeros(max(need, 1 << 18), dtype=torch.uint8, device='cuda')
    check(lib().las_colsum_bf16_ws(p(X), ldx, M, N, p(out), perm_h, p(ws), ws.numel(), stream()))
