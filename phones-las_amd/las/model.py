"""MI355X host side of the reference's ``las/model.py``: ``listener`` (las/model.py:104-142) and
``speller`` (las/model.py:205-349) with the same argument meaning, built on liblas_hip.so.

TF builds a graph and differentiates it; here each function runs the HIP kernels eagerly and, in TRAIN
mode, records what the hand-written backward needs (``Listener.backward`` / ``Speller.backward``).

``Speller`` below is the fused fast path (one decoder layer, luong | bahdanau, no attention layer, one-hot token feed);
``make_speller`` routes every other configuration -- and the binary-feature decoders -- to speller_general.GeneralSpeller.
"""
import ctypes as C
import os

import torch

from .. import hip
from . import ops
from .ops import TRAIN, EVAL, PREDICT, LSTMStateTuple

__all__ = ['listener', 'speller', 'Listener', 'Speller', 'make_speller']


def _r8(n):
    return (n + 7) // 8 * 8


# =================================================================================================
# listener
# =================================================================================================
class Listener:
    """Pyramidal BiLSTM encoder (las/ops.py:68-87) with cached bf16 weight images and a tape."""

    def __init__(self, hparams, variables, num_channels):
        self.hp = hparams
        self.F = num_channels
        self.Fp = _r8(num_channels)
        H = hparams.num_units
        nd = 1 if hparams.unidirectional else 2
        self.layers = []
        D, Dp = num_channels, self.Fp
        for l in range(hparams.num_layers):
            if hparams.use_pyramidal:
                self.layers.append(ops.LayerWeights(variables, 'listener/bilstm_{}'.format(l), D, Dp, H,
                                                    hparams.unidirectional))
                D = nd * H * (1 if l == 0 else 2)
            else:       # one MultiRNNCell stack per direction (las/model.py:111-133): layer l reads H columns
                self.layers.append(ops.LayerWeights(variables, 'listener', D, Dp, H, hparams.unidirectional,
                                                    cell_path='/{dir}/multi_rnn_cell/cell_%d/lstm_cell' % l))
                D = H
            Dp = D
        self.time_multiple = 2 ** max(0, hparams.num_layers - 1) if hparams.use_pyramidal else 1
        self.tape = None
        self.before_bottom_grads = None

    def refresh(self, variables, layers=None):
        """layers: which layers' images (default: all)."""
        with hip.image_batch():                 # every image of these layers in one launch
            for l, w in enumerate(self.layers):
                if layers is None or l in layers:
                    w.refresh(variables)

    def pad_features(self, x):
        """fp32 [B,T,F] -> bf16 [B,Tp,Fp], zero padded (Tp multiple of 2^(L-1), Fp multiple of 8)."""
        B, T, F = x.shape
        assert F == self.F, (F, self.F)
        m = self.time_multiple
        Tp = (T + m - 1) // m * m
        out = torch.empty(B, Tp, self.Fp, dtype=torch.bfloat16, device=x.device)
        hip.cast_bf16(x, T, F, out, Tp, self.Fp, ldd=self.Fp, lds=F, batch=B, sbs=T * F, dbs=Tp * self.Fp)
        return out

    def forward(self, encoder_inputs, source_sequence_length, mode, seed=0, after_first_layer=None):
        """after_first_layer: (f, g) -- f() is called between layer 0's input projection and its recurrence, g() once layer 0
        is enqueued (LasModel.refresh_images: f starts the rebuild of the upper layers' weight images on the second stream,
        beside layer 0's recurrence; g makes this stream wait for them)."""
        x = encoder_inputs
        if x.dtype != torch.bfloat16:
            x = self.pad_features(x.contiguous())
        self.tape = [] if mode == TRAIN else None
        if not self.hp.use_pyramidal:
            return ops.stacked_bilstm(x, source_sequence_length, mode, self.hp, weights=self.layers, tape=self.tape,
                                      in_features=self.F, seed=seed, after_first_layer=after_first_layer)
        return ops.pyramidal_bilstm(x, source_sequence_length, mode, self.hp, weights=self.layers,
                                    tape=self.tape, in_features=self.F, seed=seed, after_first_layer=after_first_layer)

    def backward(self, d_outputs, d_state, grads, overlap=None):
        """d_outputs: fp32 gradient w.r.t. the encoder outputs [B,T',M]; d_state: (dc, dh) [nd,B,H] or None."""
        self.backward_begin(d_outputs, d_state)
        self.backward_layers(len(self._bwd['recs']), grads, overlap)

    def backward_begin(self, d_outputs, d_state):
        """Start a backward pass that backward_layers() walks top-down in pieces (the data-parallel step exchanges the
        gradients of the layers that are done while the lower layers are still running)."""
        recs = [r for r in self.tape if r['kind'] == 'bilstm']
        self._bwd = dict(recs=recs, dy=d_outputs, d_state=d_state, next=len(recs) - 1)

    def backward_layers(self, n, grads, overlap=None, defer_last=False):
        """Backward of the next n layers (from the top).  Returns the number of layers still to do.  defer_last: the
        weight-gradient products of the last of these layers are not launched yet -- the next call launches them first
        (so that a HIP graph can end here with nothing pending on the second stream)."""
        st = self._bwd
        recs = st['recs']
        pending = st.pop('deferred', None)
        if pending is not None:
            pending()
        for i in range(n):
            l = st['next']
            if l < 0:
                break
            r = recs[l]
            dy = st['dy'].contiguous().view(r['B'], r['T'], r['nd'] * r['H'])
            defer = defer_last and (i == n - 1) and l > 0
            if l == 0 and overlap is not None:
                # everything the side stream holds before the bottom layer's weight-gradient products: LasModel's train op
                # starts on the other tensors while those products (which nothing else hides) are still running
                self.before_bottom_grads = overlap.mark()
            res = ops.bilstm_backward(r, dy, st['d_state'] if l == len(recs) - 1 else None, grads, need_dx=(l > 0),
                                      overlap=overlap, defer_weight_grads=defer, exposed=(l == 0 and overlap is not None and os.environ.get('LAS_TN_EXPOSED', '1') != '0'))
            if defer:
                st['dy'], st['deferred'] = res
            else:
                st['dy'] = res
            st['next'] = l - 1
        left = st['next'] + 1
        if left == 0:
            self.tape = None
            self._bwd = None
        return left


def listener(encoder_inputs, source_sequence_length, mode, hparams, *, variables=None, module=None):
    """las/model.py:104-142.  Returns ((encoder_outputs, source_sequence_length), encoder_state)."""
    if module is None:
        module = Listener(hparams, variables, encoder_inputs.shape[-1])
    return module.forward(encoder_inputs, source_sequence_length, mode)


# =================================================================================================
# speller
# =================================================================================================
_ATT = {'luong': hip.ATT_LUONG, 'bahdanau': hip.ATT_BAHDANAU, 'custom': hip.ATT_CUSTOM,
        'luong_monotonic': hip.ATT_LUONG_MONOTONIC, 'bahdanau_monotonic': hip.ATT_BAHDANAU_MONOTONIC}
_ATT_FUSED = ('luong', 'bahdanau')            # mechanisms of the fused single-cell fast path


def make_speller(hparams, variables, memory_depth, binf2phone=None, scope='speller', phones_only=False, binf_var=None):
    """The decoder for ``hparams``: the fused single-cell path when the configuration allows it, else the general
    cell stack (speller_general.GeneralSpeller: multi-layer, --bottom_only AttentionMultiCell, attention layer,
    embedding, the two binary-feature decoders).  phones_only: the plain phone decoder of a --multitask model (the
    first las.model.speller call of model_helper.py:211-217, binary_outputs=False); scope: variable-name prefix."""
    d = hparams
    if d.attention_type not in _ATT:
        raise ValueError('attention_type %r is not one of %s' % (d.attention_type, sorted(_ATT)))
    binf = bool(getattr(d, 'binf_projection', False))
    sigmoid = bool(getattr(d, 'binary_outputs', False)) and not binf and not phones_only
    binf = binf and not phones_only
    if (d.num_layers == 1 and not d.attention_layer_size and not d.embedding_size and d.attention_type in _ATT_FUSED
            and not binf and not sigmoid and scope == 'speller'):
        return Speller(hparams, variables, memory_depth)
    from .speller_general import GeneralSpeller
    return GeneralSpeller(hparams, variables, memory_depth, _ATT[d.attention_type],
                          binf2phone=binf2phone if (binf or sigmoid) else None, sigmoid=sigmoid, scope=scope,
                          binf_var=binf_var if (binf or sigmoid) else None)


class Speller:
    """AttentionWrapper(LSTMCell) decoder + projection layer (las/model.py:145-202,251-257): the fused fast path."""

    def __init__(self, hparams, variables, memory_depth):
        d = hparams
        if d.attention_type not in _ATT_FUSED:
            raise ValueError('attention_type %r is not on the fused path (use make_speller)' % d.attention_type)
        if d.num_layers != 1:
            raise ValueError('decoder_layers must be 1 on the HIP path this round')
        if d.attention_layer_size:
            raise ValueError('attention_layer_size is not implemented on the HIP path this round')
        if d.embedding_size:
            raise ValueError('embedding_size > 0 is not implemented on the HIP path this round')
        self.hp = d
        self.att = _ATT[d.attention_type]
        self.V, self.Vp = d.target_vocab_size, _r8(d.target_vocab_size)
        self.Hd, self.M = d.num_units, memory_depth
        if self.Hd % 8 or self.M % 8:
            raise ValueError('decoder_units and encoder depth must be multiples of 8')
        V, Vp, Hd, M = self.V, self.Vp, self.Hd, self.M
        bf, dev = torch.bfloat16, 'cuda'
        self.wmemT = torch.empty(Hd, M, dtype=bf, device=dev)
        self.wmem = torch.empty(M, Hd, dtype=bf, device=dev)
        self.kcT = torch.empty(4 * Hd, M + Hd, dtype=bf, device=dev)
        self.kc = torch.empty(M + Hd, 4 * Hd, dtype=bf, device=dev)
        self.tok = torch.empty(V, 4 * Hd, dtype=bf, device=dev)
        self.wprojT = torch.empty(Vp, M, dtype=bf, device=dev)
        self.wproj = torch.empty(M, Vp, dtype=bf, device=dev)
        self.bproj = torch.zeros(Vp, dtype=torch.float32, device=dev)
        if self.att == hip.ATT_BAHDANAU:
            self.wq = torch.empty(Hd, Hd, dtype=bf, device=dev)
            self.wq_t = torch.empty(Hd, Hd, dtype=bf, device=dev)
        self._variables = variables
        self.refresh(variables)
        self.saved = None
        self.fused_loss = None          # (loss, dlogits) of the last forward_train when it formed the loss itself (las_proj_ce)

    # names of the TF variables this module owns
    K_MEM = 'speller/memory_layer/kernel'
    K_CELL = 'speller/decoder_cell_0/lstm_cell/kernel'
    B_CELL = 'speller/decoder_cell_0/lstm_cell/bias'
    K_PROJ = 'speller/projection_layer/kernel'
    B_PROJ = 'speller/projection_layer/bias'
    K_Q = 'speller/query_layer/kernel'
    V_ATT = 'speller/attention_v'

    def refresh(self, var):
        V, Vp, Hd, M = self.V, self.Vp, self.Hd, self.M
        wm, kc, wp = var[self.K_MEM], var[self.K_CELL], var[self.K_PROJ]
        assert wm.shape == (M, Hd) and kc.shape == (V + M + Hd, 4 * Hd) and wp.shape == (M, V)
        with hip.image_batch():                 # independent images: one launch
            hip.cast_bf16(wm, M, Hd, self.wmemT, Hd, M, transpose=True)
            hip.cast_bf16(wm, M, Hd, self.wmem, M, Hd)
            hip.cast_bf16(kc[V:], M + Hd, 4 * Hd, self.kcT, 4 * Hd, M + Hd, transpose=True, lds=4 * Hd)
            hip.cast_bf16(kc[V:], M + Hd, 4 * Hd, self.kc, M + Hd, 4 * Hd, lds=4 * Hd)
            hip.cast_bf16(kc, V, 4 * Hd, self.tok, V, 4 * Hd, lds=4 * Hd)
            hip.cast_bf16(wp, M, V, self.wprojT, Vp, M, transpose=True)
            hip.cast_bf16(wp, M, V, self.wproj, M, Vp)
            hip.copy_f32(var[self.B_PROJ], V, self.bproj)
            if self.att == hip.ATT_BAHDANAU:
                hip.cast_bf16(var[self.K_Q], Hd, Hd, self.wq, Hd, Hd)
                hip.cast_bf16(var[self.K_Q], Hd, Hd, self.wq_t, Hd, Hd, transpose=True)
        self.bias = var[self.B_CELL]
        if self.att == hip.ATT_BAHDANAU:
            self.att_v = var[self.V_ATT]

    # ---------------------------------------------------------------------------------------------
    def _initial_state(self, encoder_state, B, want_zeros=True):
        d = self.hp
        dev = 'cuda'
        if d.pass_hidden_state and d.bottom_only:                         # las/model.py:259-268
            es = encoder_state[0] if isinstance(encoder_state[0], tuple) else encoder_state
            if not hasattr(es, 'c'):
                raise ValueError('pass_hidden_state needs the pyramidal listener (its state is one LSTMStateTuple per '
                                 'direction; the stacked listener returns one per layer, which TF cannot zip either)')
            c0, h0 = es.c, es.h
            if c0.shape[-1] != self.Hd:
                raise ValueError('pass_hidden_state needs decoder_units == encoder_units')
            return c0, h0, True
        if not want_zeros:
            return None, None, False
        z = torch.zeros(B, self.Hd, dtype=torch.float32, device=dev)
        return z, z, False

    def _keys(self, memory, B, Tm):
        keys = torch.empty(B, Tm, self.Hd, dtype=torch.bfloat16, device=memory.device)
        hip.gemm_nt(memory, self.wmemT, keys, B * Tm, self.Hd, self.M, lda=self.M, ldb=self.M, ldc=self.Hd,
                    out_bf16=True)
        return keys

    DEC_STREAM = 1      # generator stream of the decoder cell's input dropout

    def _persist_workspace(self, which, nbytes):
        """Workspace of the one-launch decoder ('fwd' / 'bwd'), kept across steps: the launches zero everything behind
        its 64-byte status header themselves; the header is sticky (LasModel.check_device_status reads and clears it)."""
        cache = self.__dict__.setdefault('_persist_cache', {})
        ws = cache.get(which)
        if ws is None or ws.numel() < nbytes:
            if ws is not None and int(ws[:4].view(torch.int32).item()):
                raise hip.LasError('persistent decoder reported a barrier timeout')
            ws = cache[which] = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
        return ws

    def _step_struct(self, B, Tm, z, tok_ids, tok_stride, c_prev, ldcp, gates, ldg, c_out, ldco, h_out, ldh, h2, ldh2,
                     keys, memory, mem_len, align, align_bf, lda, pq, ldpq, ctx, ldc, ctx2, ldc2):
        s = hip.DecStep()
        s.B, s.Hd, s.M, s.Tm, s.attention = B, self.Hd, self.M, Tm, self.att
        s.z, s.tok_rows, s.tok_ids, s.tok_stride = z, hip.addr(self.tok), tok_ids, tok_stride
        s.bias = hip.addr(self.bias)
        s.c_prev, s.ldcp, s.gates_out, s.ldg, s.c_out, s.ldco = c_prev, ldcp, gates, ldg, c_out, ldco
        s.h_out, s.ldh, s.h_out2, s.ldh2 = h_out, ldh, h2, ldh2
        s.keys, s.values, s.mem_len = hip.addr(keys), hip.addr(memory), hip.addr(mem_len)
        if self.att == hip.ATT_BAHDANAU:
            s.wq, s.att_v = hip.addr(self.wq), hip.addr(self.att_v)
        s.align_out, s.align_bf16, s.lda, s.pq_out, s.ldpq = align, align_bf, lda, pq, ldpq
        s.ctx_out, s.ldc, s.ctx_out2, s.ldc2 = ctx, ldc, ctx2, ldc2
        s.drop_keep, s.feed_width = 1.0, self.V + self.M
        return s

    def forward_train(self, memory, mem_len, encoder_state, targets_inputs, num_steps, parts=4, seed=0, overlap=None,
                      loss_targets=None):
        """TrainingHelper decode (las/model.py:276-296,346-347).  memory [B,T',M] bf16, targets_inputs int32
        [B,>=num_steps]; num_steps = max(target_sequence_length).  Returns logits fp32 [B,U,Vp] (first V valid)."""
        B, Tm, M = memory.shape
        Hd, V, Vp, U = self.Hd, self.V, self.Vp, num_steps
        dev, bf, f32 = memory.device, torch.bfloat16, torch.float32
        lib, st = hip.lib(), hip.stream()
        if self.saved is not None and overlap is not None:
            # a forward_train that no backward() followed left its h W_mem^T product on the side stream: joined and released here
            overlap.join()
        self.saved = None
        c0, h0, passed = self._initial_state(encoder_state, B, want_zeros=False)
        keys = self._keys(memory, B, Tm)
        W = M + Hd
        AH = torch.empty(B, U, W, dtype=bf, device=dev)
        cs = torch.empty(B, U + 1, Hd, dtype=f32, device=dev)
        gates = torch.empty(B, U, 4 * Hd, dtype=f32, device=dev)
        h_all = torch.empty(B, U, Hd, dtype=bf, device=dev)
        Tmp = _r8(Tm)                      # row stride of per-step [T'] vectors (GEMM operand alignment)
        # Dot-product scores (round 4): d(memory)[b] = align[b]^T d(context)[b] + d(scores)[b]^T (h[b] W_mem^T) is ONE batched
        # product over K = 2U when the two left operands and the two right operands sit behind each other per utterance:
        # rows [0, U) of `al2` are the alignments (this pass), rows [U, 2U) d(scores) (the backward pass); rows [0, U) of `cw2`
        # d(context) (the backward pass), rows [U, 2U) hW = h_t W_mem^T (formed beside the loss, off the critical path).  The
        # fp32 alignments share the row stride (struct las_dec_step has one lda for both).
        merged = (self.att == hip.ATT_LUONG and os.environ.get('LAS_DMEM_MERGED', '1') != '0')
        R = 2 if merged else 1
        align = torch.empty(B, R * U, Tmp, dtype=f32, device=dev)
        al2 = torch.empty(B, R * U, Tmp, dtype=bf, device=dev)
        align_bf = al2[:, :U]
        dc_bwd = torch.empty(B, Hd, dtype=f32, device=dev)          # the backward pass's running d(c): cleared here, in the same launch
        # one launch: the first operand row [attention_{-1} = 0 | h_0], c_0, and the alignment buffers' zero padding
        hip.fill_many(zero=[AH[:, 0, :M], align.view(B, -1)[:, :U * Tmp], al2, dc_bwd],
                      copy=[(AH[:, 0, M:], h0.float() if passed else None), (cs[:, 0], c0.float() if passed else None)])
        ctx_all = torch.empty(B, U, M, dtype=bf, device=dev)
        pq_all = torch.empty(B, U, Hd, dtype=f32, device=dev) if self.att == hip.ATT_BAHDANAU else None
        z = torch.empty(B, 4 * Hd, dtype=f32, device=dev)
        d = self.hp
        keep = 1.0 - (d.dropout if d.dropout else 0.0)
        sampling = float(d.sampling_probability or 0.0)
        tin = targets_inputs
        fed = tin
        logits = None
        if sampling > 0.0:
            # scheduled sampling (utils/training_helper.py:48-87): fed[:, t+1] = sampled token or the teacher's
            fed = tin[:, :U].contiguous().clone()
            logits = torch.empty(B, U, Vp, dtype=f32, device=dev)
        ts = fed.stride(0)
        persist = (B <= 4 * lib.las_decoder_persist_max_batch() and Vp <= 1024 and os.environ.get('LAS_DEC_PERSIST', '1') != '0' and      # B: co-residency
                   lib.las_decoder_persist_supported(Hd, M, W, self.att, hip.NORM_SOFTMAX) == 1 and
                   (Hd <= 256 or sampling == 0.0))          # (the in-launch sampling phase is built for decoder_units <= 256)
        if persist:
            # all U steps in one persistent launch (see las_dec_persist in las_hip.h)
            p = hip.DecPersist()
            p.s = self._step_struct(
                B, Tm, 0, hip.addr(fed), ts, hip.addr(cs), (U + 1) * Hd, hip.addr(gates), U * 4 * Hd,
                hip.addr(cs, Hd), (U + 1) * Hd, hip.addr(h_all), U * Hd, hip.addr(AH, W + M) if U > 1 else 0, U * W,
                keys, memory, mem_len, hip.addr(align), hip.addr(align_bf), R * U * Tmp,
                hip.addr(pq_all) if pq_all is not None else 0, U * Hd, hip.addr(ctx_all), U * M,
                hip.addr(AH, W) if U > 1 else 0, U * W)
            if keep < 1.0:
                p.s.drop_keep, p.s.drop_seed, p.s.drop_stream = keep, seed, self.DEC_STREAM
            p.U, p.K_in = U, W
            p.inc_tok, p.inc_cprev, p.inc_gates, p.inc_cout, p.inc_h, p.inc_h2 = 1, Hd, 4 * Hd, Hd, Hd, W
            p.inc_align, p.inc_pq, p.inc_ctx, p.inc_ctx2 = Tmp, Hd, M, W
            p.x, p.ldx, p.inc_x = hip.addr(AH), U * W, W
            p.kT, p.ldk = hip.addr(self.kcT), W
            ws = self._persist_workspace('fwd', lib.las_decoder_persist_workspace_bytes(B, Tm, Hd, M))
            p.workspace = hip.addr(ws)       # status, group flags and the exchange granules (z_t, raw scores)
            if sampling > 0.0:           # logits and the sampled feed are produced inside the launch
                plog = torch.empty(U, B, 4, Vp, dtype=f32, device=dev)
                p.sampling_prob, p.seed = sampling, seed
                p.teacher, p.teacher_stride = hip.addr(tin), tin.stride(0)
                p.wprojT, p.ldw, p.bproj = hip.addr(self.wprojT), M, hip.addr(self.bproj)
                p.logits, p.ld_logits, p.plog, p.V, p.Vp = 0, U * Vp, hip.addr(plog), V, Vp
            # per step and utterance: the cell product [attention, h] K, the scores and the context
            tok = hip.prof_begin('dec_persist_fwd', 2.0 * U * B * (W * 4 * Hd + Tm * Hd + Tm * M))
            hip.check(lib.las_decoder_persist_fwd(C.byref(p), st))
            hip.prof_end(tok)
            self._persist_ws = ws
            logits = None                # the launch only forms the logits it samples from; all of them come from one GEMM below
        for t in range(0 if not persist else U, U):
            hip.gemm_nt(AH[:, t], self.kcT, z, B, 4 * Hd, W, lda=U * W, ldb=W, ldc=4 * Hd)
            last = (t + 1 == U)
            s = self._step_struct(
                B, Tm, hip.addr(z), hip.addr(fed, t), ts, hip.addr(cs, t * Hd), (U + 1) * Hd,
                hip.addr(gates, t * 4 * Hd), U * 4 * Hd, hip.addr(cs, (t + 1) * Hd), (U + 1) * Hd,
                hip.addr(h_all, t * Hd), U * Hd, 0 if last else hip.addr(AH, (t + 1) * W + M), U * W,
                keys, memory, mem_len, hip.addr(align, t * Tmp), hip.addr(al2, t * Tmp), R * U * Tmp,
                hip.addr(pq_all, t * Hd) if pq_all is not None else 0, U * Hd,
                hip.addr(ctx_all, t * M), U * M, 0 if last else hip.addr(AH, (t + 1) * W), U * W)
            if keep < 1.0:
                s.drop_keep, s.drop_seed, s.drop_stream, s.step = keep, seed, self.DEC_STREAM, t
            hip.check(lib.las_decoder_step_fwd(C.byref(s), parts, st))
            if sampling > 0.0:
                hip.gemm_nt(ctx_all[:, t], self.wprojT, logits[:, t], B, Vp, M, lda=U * M, ldb=M, ldc=U * Vp,
                            bias=self.bproj)
                if not last:
                    hip.check(lib.las_sample_tokens(hip.addr(logits, t * Vp), U * Vp, V, hip.addr(tin, t + 1),
                                                    tin.stride(0), hip.addr(fed, t + 1), ts, B, sampling, seed, t, st))
        cw2 = None
        if merged:
            # hW[b, u] = h_t W_mem^T into rows [U, 2U) of cw2, on the second stream: beside the projection, the loss and the
            # decoder's backward launch; the d(memory) product after that launch is the first to read it
            cw2 = torch.empty(B, 2 * U, M, dtype=bf, device=dev)
            with (overlap or ops._NoOverlap()).fork(h_all, cw2):
                hip.gemm_nt(h_all, self.wmem, cw2[:, U:], U, M, Hd, lda=Hd, ldb=Hd, ldc=M, out_bf16=True, batch=B,
                            sa=U * Hd, sb=0, sc=2 * U * M)
        self.fused_loss, dattn_proj = None, None
        if (logits is None and loss_targets is not None and os.environ.get('LAS_PROJ_CE', '1') != '0'
                and lib.las_proj_ce_supported(V, Vp, M) == 1):
            # projection + sequence loss + the product back through the projection in one launch (las_proj_ce): what used to be
            # five small launches between the decoder's forward and backward launches.  loss_targets = (targets_outputs int32
            # [B, >= U], target_sequence_length int32 [B], grad_scale): compute_loss(TRAIN)'s arguments (model_helper.py:24-30).
            tout, tlen, gscale = loss_targets
            if tout.dtype != torch.int32 or tout.stride(1) != 1:
                tout = tout.to(torch.int32).contiguous()
            tlen = tlen.to(torch.int32)
            logits = torch.empty(B, U, Vp, dtype=f32, device=dev)
            dlog = torch.empty(B, U, Vp, dtype=bf, device=dev)
            dattn_proj = torch.empty(B, U, M, dtype=f32, device=dev)
            loss = torch.empty(1, dtype=f32, device=dev)
            need = lib.las_proj_ce_workspace_bytes(B, U)
            if getattr(self, '_projce_ws', None) is None or self._projce_ws.numel() < need:
                if torch.cuda.is_current_stream_capturing():
                    raise hip.LasError('las_proj_ce: new workspace during graph capture (run the step once eagerly first)')
                self._projce_ws = torch.zeros(need, dtype=torch.uint8, device=dev)
            hip.check(lib.las_proj_ce(hip.p(ctx_all), M, hip.p(self.wprojT), hip.p(self.bproj), hip.p(self.wproj), hip.p(tout),
                                      tout.stride(0), hip.p(tlen), B, U, V, Vp, M, float(gscale), hip.p(logits), hip.p(dlog),
                                      hip.p(dattn_proj), M, hip.p(loss), hip.p(self._projce_ws), st))
            self.fused_loss = (loss, dlog)
        if logits is None:
            logits = torch.empty(B, U, Vp, dtype=f32, device=dev)
            hip.gemm_nt(ctx_all, self.wprojT, logits, B * U, Vp, M, lda=M, ldb=M, ldc=Vp, bias=self.bproj)
        self.saved = dict(dattn_proj=dattn_proj, keep=keep, seed=seed, fed=fed, B=B, Tm=Tm, U=U, memory=memory, mem_len=mem_len, keys=keys, AH=AH, cs=cs, gates=gates,
                          h_all=h_all, align=align, align_bf=align_bf, ctx_all=ctx_all, pq_all=pq_all,
                          tin=tin, passed=passed, al2=al2, cw2=cw2, R=R, dc=dc_bwd)
        return logits

    # ---------------------------------------------------------------------------------------------
    def backward(self, dlogits, grads, overlap=None):
        """dlogits bf16 [B,U,Vp] (zero in the pad columns).  Accumulates weight gradients into ``grads`` and
        returns (d_memory fp32 [B,T',M], d_encoder_state or None)."""
        sv = self.saved
        if sv is None:
            # (the running d(c) and the d(scores) rows are cleared by forward_train's fill launch only: a second pass over the
            # same saved forward would start from the first one's leftovers)
            raise hip.LasError('Speller.backward: no saved forward pass (every forward_train is consumed by ONE backward)')
        B, Tm, U = sv['B'], sv['Tm'], sv['U']
        Tmp = _r8(Tm)
        Hd, V, Vp, M = self.Hd, self.V, self.Vp, self.M
        W = M + Hd
        dev, bf, f32 = dlogits.device, torch.bfloat16, torch.float32
        lib, st = hip.lib(), hip.stream()
        BU = B * U
        # d(decoder outputs) through the projection: already formed with the loss when the gradient handed in IS that loss's
        dattn_proj = sv.get('dattn_proj') if (self.fused_loss is not None and dlogits is self.fused_loss[1]) else None
        if dattn_proj is None:
            dattn_proj = torch.empty(B, U, M, dtype=f32, device=dev)
            hip.gemm_nt(dlogits, self.wproj, dattn_proj, BU, M, Vp, lda=Vp, ldb=Vp, ldc=M)
        R, merged = sv['R'], sv['cw2'] is not None
        dc = sv['dc']                       # cleared in the forward pass's fill launch
        dfeed = torch.empty(B, W, dtype=f32, device=dev)
        dz_all = torch.empty(B, U, 4 * Hd, dtype=bf, device=dev)
        if merged:
            # d(scores) behind the alignments, d(context) in front of hW (forward_train): both already zero / in place
            ds_all, ldso = sv['al2'][:, U:], 2 * U * Tmp
            dctx_all, ldds = sv['cw2'][:, :U], 2 * U * M
        else:
            ds_all, ldso = torch.empty(B, U, Tmp, dtype=bf, device=dev), U * Tmp
            hip.fill_many(zero=[ds_all])
            dctx_all, ldds = torch.empty(B, U, M, dtype=bf, device=dev), U * M
        bah = self.att == hip.ATT_BAHDANAU
        if bah:
            dkeys = torch.zeros(B, Tm, Hd, dtype=f32, device=dev)
            dpq_all = torch.empty(B, U, Hd, dtype=bf, device=dev)
            dv = grads[self.V_ATT]
        persist = (B <= 4 * lib.las_decoder_persist_max_batch() and os.environ.get('LAS_DEC_PERSIST', '1') != '0' and
                   lib.las_decoder_persist_bwd_supported(Hd, M, W, self.att, hip.NORM_SOFTMAX) == 1)
        if persist:
            # all U steps in one persistent launch (las_dec_persist_bwd in las_hip.h)
            p = hip.DecPersistBwd()
            s = p.s
            s.B, s.Hd, s.M, s.Tm, s.attention = B, Hd, M, Tm, self.att
            s.dctx_a, s.ldda = hip.addr(dattn_proj), U * M
            s.dctx_save, s.ldds = hip.addr(dctx_all), ldds
            s.dc = hip.addr(dc)
            s.gates, s.ldg = hip.addr(sv['gates']), U * 4 * Hd
            s.c_new, s.ldcn = hip.addr(sv['cs'], Hd), (U + 1) * Hd
            s.c_prev, s.ldcp = hip.addr(sv['cs']), (U + 1) * Hd
            s.align, s.lda = hip.addr(sv['align']), R * U * Tmp
            s.keys, s.values, s.mem_len = hip.addr(sv['keys']), hip.addr(sv['memory']), hip.addr(sv['mem_len'])
            s.dz, s.ldz = hip.addr(dz_all), U * 4 * Hd
            s.ds_out, s.ldso = hip.addr(ds_all), ldso
            s.drop_keep, s.feed_width = 1.0, V + M
            if sv['keep'] < 1.0:
                s.drop_keep, s.drop_seed, s.drop_stream = sv['keep'], sv['seed'], self.DEC_STREAM
            if bah:
                s.pq, s.ldpq = hip.addr(sv['pq_all']), U * Hd
                s.wq_t, s.att_v = hip.addr(self.wq_t), hip.addr(self.att_v)
                s.dkeys_acc, s.dv_acc = hip.addr(dkeys), hip.addr(dv)
                s.dpq_out, s.lddpq = hip.addr(dpq_all), U * Hd
            p.U, p.W = U, W
            p.inc_a, p.inc_save, p.inc_gates, p.inc_c, p.inc_align, p.inc_dz, p.inc_ds = M, M, 4 * Hd, Hd, Tmp, 4 * Hd, Tmp
            p.inc_pq = Hd
            p.kc, p.ldk = hip.addr(self.kc), 4 * Hd
            dfeed_all = torch.empty(1, B, W, dtype=f32, device=dev)      # only step 0's row leaves the launch: d(initial feed)
            ws = self._persist_workspace('bwd', lib.las_decoder_persist_workspace_bytes(B, Tm, Hd, M))
            p.dfeed_all, p.workspace = hip.addr(dfeed_all), hip.addr(ws)
            if self.att == hip.ATT_BAHDANAU:        # d(attention_v) summed over the workgroups in a fixed order (no atomics)
                p.sum_workspace = hip.addr(self._persist_workspace('sum', lib.las_decoder_sum_workspace_bytes(32 * (((B + 7) // 8 + 7) // 8 * 8), Hd)))
            # per step and utterance: dz K^T, d(align) = values . d(context), the d(query) sum over the keys
            tok = hip.prof_begin('dec_persist_bwd', 2.0 * U * B * (W * 4 * Hd + Tm * Hd + Tm * M))
            hip.check(lib.las_decoder_persist_bwd(C.byref(p), st))
            hip.prof_end(tok)
            self._persist_ws_bwd = ws
            dfeed = dfeed_all[0]
        for t in range(U - 1 if not persist else -1, -1, -1):
            first = (t == U - 1)
            s = hip.DecStepBwd()
            s.B, s.Hd, s.M, s.Tm, s.attention = B, Hd, M, Tm, self.att
            s.dctx_a, s.ldda = hip.addr(dattn_proj, t * M), U * M
            s.dctx_b, s.lddb = (0 if first else hip.addr(dfeed)), W
            s.dctx_save, s.ldds = hip.addr(dctx_all, t * M), ldds
            s.dh_rec, s.ldr = (0 if first else hip.addr(dfeed, M)), W
            s.dc = hip.addr(dc)
            s.gates, s.ldg = hip.addr(sv['gates'], t * 4 * Hd), U * 4 * Hd
            s.c_new, s.ldcn = hip.addr(sv['cs'], (t + 1) * Hd), (U + 1) * Hd
            s.c_prev, s.ldcp = hip.addr(sv['cs'], t * Hd), (U + 1) * Hd
            s.align, s.lda = hip.addr(sv['align'], t * Tmp), R * U * Tmp
            s.keys, s.values, s.mem_len = hip.addr(sv['keys']), hip.addr(sv['memory']), hip.addr(sv['mem_len'])
            s.dz, s.ldz = hip.addr(dz_all, t * 4 * Hd), U * 4 * Hd
            s.ds_out, s.ldso = hip.addr(ds_all, t * Tmp), ldso
            s.drop_keep, s.feed_width = 1.0, V + M
            if sv['keep'] < 1.0:
                s.drop_keep, s.drop_seed, s.drop_stream, s.step = sv['keep'], sv['seed'], self.DEC_STREAM, t
            if bah:
                s.pq, s.ldpq = hip.addr(sv['pq_all'], t * Hd), U * Hd
                s.wq_t, s.att_v = hip.addr(self.wq_t), hip.addr(self.att_v)
                s.dkeys_acc, s.dv_acc = hip.addr(dkeys), hip.addr(dv)
                s.dpq_out, s.lddpq = hip.addr(dpq_all, t * Hd), U * Hd
            hip.check(lib.las_decoder_step_bwd(C.byref(s), st))
            hip.gemm_nt(dz_all[:, t], self.kc, dfeed, B, W, 4 * Hd, lda=U * 4 * Hd, ldb=4 * Hd, ldc=W)
        # critical path: d(keys), d(memory) feed the listener's backward
        # (stored, not accumulated: no zero fills in front of them, no atomics -- K is only the U steps; d(keys) of the
        # dot-product scores is only ever an operand of further products: written as bf16 at once)
        dkeys_bf = torch.empty(B * Tm, Hd, dtype=bf, device=dev)
        dmem = torch.empty(B, Tm, M, dtype=f32, device=dev)

        def dkeys_from_scores():            # d(keys)[b] = d(scores)[b]^T h[b] (bf16: only ever an operand of further products)
            hip.gemm_tn(ds_all, sv['h_all'], dkeys_bf, Tm, Hd, U, lda=Tmp, ldb=Hd, ldc=Hd, batch=B, sa=ldso,
                        sb=U * Hd, sc=Tm * Hd, store=True)
        if merged:
            # one batched product over K = 2U: [align | d(scores)]^T [d(context) ; h W_mem^T]; d(keys) itself is only needed
            # for the memory layer's weight gradient and moves to the second stream
            if overlap is not None:
                overlap.join_side()          # hW
            hip.gemm_tn(sv['al2'], sv['cw2'], dmem, Tm, M, 2 * U, lda=Tmp, ldb=M, ldc=M, batch=B, sa=2 * U * Tmp,
                        sb=2 * U * M, sc=Tm * M, store=True)
        else:
            if not bah:
                dkeys_from_scores()
            else:
                hip.cast_bf16(dkeys, B * Tm, Hd, dkeys_bf, B * Tm, Hd, ldd=Hd, lds=Hd)
            hip.gemm_tn(sv['align_bf'], dctx_all, dmem, Tm, M, U, lda=Tmp, ldb=M, ldc=M, batch=B, sa=ldso, sb=ldds,
                        sc=Tm * M, store=True)
            hip.gemm_nt(dkeys_bf, self.wmem, dmem, B * Tm, M, Hd, lda=Hd, ldb=Hd, ldc=M, accumulate=True)
        # weight gradients (off the critical path): cell rows [V, V+M+Hd) from [attention_{t-1}, h_{t-1}], rows
        # [0,V) from the tokens, projection, memory_layer, query_layer
        onehot = torch.empty(BU, Vp, dtype=bf, device=dev)
        fed = sv['fed']
        keep = [sv['AH'], dz_all, onehot, sv['ctx_all'], dlogits, sv['memory'], dkeys_bf, sv['h_all'], sv['al2'], sv['cw2']]
        if bah:
            keep.append(dpq_all)
        with (overlap or ops._NoOverlap()).fork(*keep, beside_chain=True):     # (beside the top listener layer's recurrence)
            if merged:
                dkeys_from_scores()
            # (the one-hot rows of the fed tokens are an operand of the weight gradients only: built on this stream)
            hip.check(lib.las_onehot_bf16(hip.p(fed), fed.stride(0), B, U, V, hip.p(onehot), Vp, sv['keep'], sv['seed'],
                                          self.DEC_STREAM, V + M, hip.stream()))
            gk = grads[self.K_CELL]
            hip.gemm_tn(sv['AH'], dz_all, gk[V:], W, 4 * Hd, BU, lda=W, ldb=4 * Hd, ldc=4 * Hd, split_k=4)
            hip.gemm_tn(onehot, dz_all, gk, V, 4 * Hd, BU, lda=Vp, ldb=4 * Hd, ldc=4 * Hd, split_k=4)
            hip.colsum_bf16(dz_all, BU, 4 * Hd, grads[self.B_CELL], ldx=4 * Hd)
            hip.gemm_tn(sv['ctx_all'], dlogits, grads[self.K_PROJ], M, V, BU, lda=M, ldb=Vp, ldc=V, split_k=4)
            hip.colsum_bf16(dlogits, BU, V, grads[self.B_PROJ], ldx=Vp)
            hip.gemm_tn(sv['memory'], dkeys_bf, grads[self.K_MEM], M, Hd, B * Tm, lda=M, ldb=Hd, ldc=Hd, split_k=8)
            if bah:
                hip.gemm_tn(sv['h_all'], dpq_all, grads[self.K_Q], Hd, Hd, BU, lda=Hd, ldb=Hd, ldc=Hd, split_k=4)
        d_state = None
        if sv['passed']:
            d_state = (dc, dfeed[:, M:])
        self.saved = None
        return dmem, d_state

    # ---------------------------------------------------------------------------------------------
    def forward_greedy(self, memory, mem_len, encoder_state, max_iterations, parts=4):
        """GreedyEmbeddingHelper decode (las/model.py:270-274,337-347).  Returns (logits [B,S,V] fp32, sample_ids
        [B,S] int32, final_sequence_length [B] int32, alignments [B,S,T'] fp32) with S <= max_iterations."""
        d = self.hp
        B, Tm, M = memory.shape
        Hd, V, Vp = self.Hd, self.V, self.Vp
        dev, bf, f32 = memory.device, torch.bfloat16, torch.float32
        lib, st = hip.lib(), hip.stream()
        c0, h0, _ = self._initial_state(encoder_state, B)
        keys = self._keys(memory, B, Tm)
        W = M + Hd
        S = max_iterations
        ah = torch.zeros(2, B, W, dtype=bf, device=dev)
        ah[0, :, M:].copy_(h0)
        cs = torch.empty(2, B, Hd, dtype=f32, device=dev)
        cs[0].copy_(c0)
        gates = torch.empty(B, 4 * Hd, dtype=f32, device=dev)
        h_t = torch.empty(B, Hd, dtype=bf, device=dev)
        align = torch.zeros(B, S, Tm, dtype=f32, device=dev)
        ctx = torch.empty(B, M, dtype=bf, device=dev)
        pq = torch.empty(B, Hd, dtype=f32, device=dev) if self.att == hip.ATT_BAHDANAU else None
        z = torch.empty(B, 4 * Hd, dtype=f32, device=dev)
        logits = torch.zeros(B, S, Vp, dtype=f32, device=dev)
        ids = torch.full((B,), d.sos_id, dtype=torch.int32, device=dev)
        samples = torch.full((B, S), d.eos_id, dtype=torch.int32, device=dev)
        finished = torch.zeros(B, dtype=torch.bool, device=dev)
        final_len = torch.zeros(B, dtype=torch.int32, device=dev)
        steps = 0
        for t in range(S):
            cur, nxt = t & 1, (t + 1) & 1
            hip.gemm_nt(ah[cur], self.kcT, z, B, 4 * Hd, W, lda=W, ldb=W, ldc=4 * Hd)
            s = self._step_struct(
                B, Tm, hip.addr(z), hip.addr(ids), 1, hip.addr(cs[cur]), Hd, hip.addr(gates), 4 * Hd,
                hip.addr(cs[nxt]), Hd, hip.addr(h_t), Hd, hip.addr(ah[nxt], M), W, keys, memory, mem_len,
                hip.addr(align, t * Tm), 0, S * Tm, hip.addr(pq) if pq is not None else 0, Hd,
                hip.addr(ctx), M, hip.addr(ah[nxt]), W)
            hip.check(lib.las_decoder_step_fwd(C.byref(s), parts, st))
            lg = logits[:, t]
            hip.gemm_nt(ctx, self.wprojT, lg, B, Vp, M, lda=M, ldb=M, ldc=S * Vp, bias=self.bproj)
            sample = lg[:, :V].argmax(-1).to(torch.int32)
            samples[:, t] = sample
            final_len = torch.where(finished, final_len, torch.full_like(final_len, t + 1))
            finished = finished | (sample == d.eos_id)
            ids = sample.contiguous()
            steps = t + 1
            if bool(finished.all()):
                break
        return logits[:, :steps, :V], samples[:, :steps], final_len, align[:, :steps]


BasicDecoderOutput = None
FinalBeamSearchDecoderOutput = None


def _output_types():
    global BasicDecoderOutput, FinalBeamSearchDecoderOutput
    if BasicDecoderOutput is None:
        import collections
        BasicDecoderOutput = collections.namedtuple('BasicDecoderOutput', ['rnn_output', 'sample_id'])
        FinalBeamSearchDecoderOutput = collections.namedtuple('FinalBeamSearchDecoderOutput',
                                                              ['predicted_ids', 'beam_search_decoder_output'])
    return BasicDecoderOutput, FinalBeamSearchDecoderOutput


def speller_kind(hparams, binary_outputs=False, binf_embedding=None):
    """Which decoder a las.model.speller(...) call builds (las/model.py:228-257): 'sigmoid' when the caller passes
    binary_outputs (projection = Dense(binf_count), feature-vector inputs), 'binf_projection' when hparams.binf_projection is
    set and the feature map is handed in (attention layer of 2*binf_count, DenseBinfDecoder's fixed map), else 'phones'."""
    if binary_outputs:
        return 'sigmoid'
    if bool(getattr(hparams, 'binf_projection', False)) and binf_embedding is not None:
        return 'binf_projection'
    return 'phones'


def speller(encoder_outputs, encoder_state, decoder_inputs, source_sequence_length, target_sequence_length, mode,
            hparams, binary_outputs=False, binf_embedding=None, transparent_projection=False, *, variables=None,
            module=None, num_steps=None, scope='speller'):
    """las/model.py:205-349, same positional arguments.  Returns (decoder_outputs, final_context_state,
    final_sequence_length) as the reference's dynamic_decode does:

      TRAIN                          BasicDecoderOutput(rnn_output = logits [B,U,.], sample_id = argmax ids); with
                                     hparams.binf_projection the rnn_output is [phone logits | raw 2*binf_count outputs]
                                     (DenseBinfDecoder(concat_cell_outputs=True), las/model.py:251-257); with binary_outputs
                                     the decoder inputs are the targets' FEATURE VECTORS [B,U,binf_count] (model_helper.py:
                                     199-200,222) or their token ids, and the outputs feature logits
      EVAL / PREDICT, beam_width 0   greedy decode (GreedyEmbeddingHelper; the InferenceHelper of :320-336 for binary_outputs
                                     without a map); transparent_projection: rnn_output = the raw cell outputs
                                     (BasicTransparentProjectionDecoder, utils/training_helper.py:156-178)
      PREDICT, beam_width > 0        FinalBeamSearchDecoderOutput(predicted_ids [B,T,K]); decoder_inputs = the partial targets
                                     the search starts from (las/model.py:298-311) or None

    final_context_state is the decoder module (its ``alignment_history`` [B,S,T'] is what get_alignment_history reads).
    variables: {tf name: fp32 CUDA tensor} under ``scope`` (model_helper.param_table), or pass a ready ``module``."""
    Out, BeamOut = _output_types()
    kind = speller_kind(hparams, binary_outputs, binf_embedding)
    beam = int(getattr(hparams, 'beam_width', 0) or 0) if mode == PREDICT else 0
    if kind == 'sigmoid' and binf_embedding is not None and mode != TRAIN:
        raise ValueError('binary_outputs with a feature map outside TRAIN: the reference samples through '
                         'transform_binf_to_phones on binf_count-wide outputs, whose [nf:2nf] half is empty '
                         '(utils/training_helper.py:17-27 from las/model.py:251-257); decode with binf_embedding=None '
                         '(InferenceHelper) or use --binf_projection')
    if module is None:
        from .speller_general import GeneralSpeller
        if kind == 'phones':
            module = make_speller(hparams, variables, encoder_outputs.shape[-1], scope=scope, phones_only=True)
        else:
            module = GeneralSpeller(hparams, variables, encoder_outputs.shape[-1], _ATT[hparams.attention_type],
                                    binf2phone=binf_embedding, sigmoid=(kind == 'sigmoid'), scope=scope)
    if beam > 0 and isinstance(module, Speller):      # the search gathers decoder state between steps: general cell stack
        from .speller_general import GeneralSpeller
        module = GeneralSpeller(hparams, variables if variables is not None else module._variables, module.M,
                                _ATT[hparams.attention_type], scope=scope)
    if mode == TRAIN:
        vec = None
        if kind == 'sigmoid' and decoder_inputs.is_floating_point():      # feature vectors, as model_helper.py:222 passes them
            vec, decoder_inputs = decoder_inputs, None
        n = vec.shape[1] if vec is not None else None
        U = num_steps if num_steps is not None else int(target_sequence_length.max().item())
        if vec is not None:
            logits = module.forward_train(encoder_outputs, source_sequence_length, encoder_state, None, U, input_vectors=vec)
        else:
            logits = module.forward_train(encoder_outputs, source_sequence_length, encoder_state, decoder_inputs, U)
        Vo = getattr(module, 'Vo', module.V)
        out = logits[..., :Vo]
        ids = out.argmax(-1).to(torch.int32)
        if kind == 'binf_projection':       # concat_cell_outputs: model_helper.py:245-246 splits them again
            out = torch.cat([out, module.saved['att'].float()], -1)
        elif kind == 'sigmoid':             # TrainingSigmoidHelper.sample without a map: round(sigmoid(outputs))
            ids = (out > 0).to(torch.float32)
        return Out(out, ids), module, target_sequence_length
    max_len = int(source_sequence_length.max().item())
    max_it = int(round(max_len * hparams.decoding_length_factor))       # las/model.py:270-274
    if beam > 0:
        ids, lens, lps = module.forward_beam(encoder_outputs, source_sequence_length, encoder_state, max_it, beam,
                                             partial_targets=decoder_inputs)
        module.beam_log_probs = lps
        return BeamOut(ids, None), module, lens
    logits, ids, final_len, align = module.forward_greedy(encoder_outputs, source_sequence_length, encoder_state, max_it)
    module.alignment_history = align
    if transparent_projection:
        if kind != 'binf_projection':
            raise ValueError('transparent_projection needs the binf_projection decoder (its raw [lp1 | lp0] outputs)')
        logits = module.last_raw_outputs.float()
    return Out(logits, ids), module, final_len
