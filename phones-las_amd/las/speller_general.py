"""General decoder of las/model.py:145-349 on the HIP kernels: any number of LSTM layers, with the attention either
wrapped around the whole MultiRNNCell stack (default) or around the bottom cell only (``--bottom_only``:
AttentionMultiCell, las/model.py:20-69), optional ``--attention_layer_size`` (Dense(A, no bias) on
[cell_output, context]) and ``--embedding_size`` (trainable target embedding instead of one-hot).

The single-layer / no-attention-layer / one-hot configuration has a fused fast path in ``model.Speller``; this module
composes the same kernels un-fused (LAS_DEC_CELL_ONLY / LAS_DEC_ATTENTION_ONLY) plus skinny GEMMs.  Step t:

  bottom_only:      x0=[emb(y), att_{t-1}] -> cell0 -> h0 -> attention(h0) -> att_t
                    l>=1: [cur, att_{t-1}] -> cell_l -> h_l (cur = att_t for l=1, h_{l-1} above); output = h_top
  default:          x0=[emb(y), att_{t-1}] -> cell0 -> ... -> cell_top -> attention(h_top) -> att_t; output = att_t
  att_t = Dense([query, context]) if attention_layer_size else context.

Input dropout (DropoutWrapper on every cell's input: the GEMM operand is dropped in place, the one-hot token through
the kernel's token scale) and scheduled sampling work as on the fused path; dropout + embedding raises."""
import ctypes as C
import os

import torch

from .. import hip

__all__ = ['GeneralSpeller', 'gather_tree']


def _r8(n):
    return (n + 7) // 8 * 8


def gather_tree(step_ids, parent_ids, max_len, end_token):
    """tf.contrib.seq2seq.gather_tree (beam_search_ops; BeamSearchDecoder.finalize): step_ids / parent_ids [T,B,K]
    (numpy int) -> full beams [T,B,K].  Every final beam is backtracked through its parents from t = max_len[b]-1;
    positions >= max_len[b] and everything after a beam's first end_token are end_token.  Host-side index chasing."""
    import numpy as np
    T, B, K = step_ids.shape
    out = np.full_like(step_ids, end_token)
    for b in range(B):
        L = min(int(max_len[b]), T)
        for k in range(K):
            parent = k
            for t in range(L - 1, -1, -1):
                out[t, b, k] = step_ids[t, b, parent]
                parent = int(parent_ids[t, b, parent])
            seen = False
            for t in range(L):
                if seen:
                    out[t, b, k] = end_token
                elif int(out[t, b, k]) == end_token:
                    seen = True
    return out


class GeneralSpeller:
    NOISE_STREAM = 3                     # generator stream of the monotonic-attention score noise (draw (t*B + b)*Tm + t')

    def __init__(self, hparams, variables, memory_depth, att_code, binf2phone=None, sigmoid=False, scope='speller', binf_var=None):
        """binf2phone [binf_count, V] (0/1, constant): the --binf_projection decoder (las/model.py:179-183,242-257,
        utils/training_helper.py:17-27,122-153): tokens are fed as their binary-feature vectors, the attention layer
        emits A = 2*binf_count values [log p(f=1) | log p(f=0)] and the 'projection' is the fixed map
        logits = lp1 * Mb + lp0 * (1 - Mb); the Dense kernel/bias of projection_layer exist as variables but are not
        applied (inner_projection_layer=False).
        sigmoid: the sigmoid-output decoder (--binary_outputs without --binf_projection; las/model.py:237-241,251-257,
        283-291,320-336): the decoder input of a step is a binary-feature VECTOR (TRAIN: the features of the teacher's
        token, rows of binf2phone^T; inference: the rounded sigmoid of the previous output), the projection is a plain
        Dense(binf_count) and its outputs are feature logits.  binf2phone may then be None (inference without a map).
        scope: variable-name prefix ('speller'; the second decoder of --multitask lives under 'speller_binf',
        model_helper.py:219-227).
        binf_var: name of the TRAINABLE feature map among `variables` (--binf_trainable: model_helper.py:182-184 makes
        binf2phone a variable initialised U(0, 1) instead of a constant; --binf_projection decoders only).  The token feed
        (rows of Mb^T) and the output map [Mb; 1 - Mb] are then rebuilt from it at every refresh(), and backward() adds its
        gradient: d(Mb) = (raw^T dlogits)[:nf] - (raw^T dlogits)[nf:] + (d embedded tokens)^T."""
        d = hparams
        self.binf_var = binf_var
        if binf_var is not None:
            if sigmoid:
                raise ValueError('--binf_trainable with the sigmoid-output decoder: the reference differentiates its sigmoid loss '
                                 'through the TARGETS too (targets_binf is a lookup in the variable, model_helper.py:199); only '
                                 '--binf_projection decoders take a trainable map on the HIP path')
            binf2phone = variables[binf_var]
        self.scope = scope
        self.K_MEM, self.K_PROJ, self.B_PROJ = scope + '/memory_layer/kernel', scope + '/projection_layer/kernel', scope + '/projection_layer/bias'
        self.K_Q, self.V_ATT, self.K_AL = scope + '/query_layer/kernel', scope + '/attention_v', scope + '/attention_layer/kernel'
        self.K_EMB, self.B_SCORE = scope + '/target_embedding', scope + '/attention_score_bias'
        self.sigmoid = bool(sigmoid)
        self.binf = binf2phone if not sigmoid else None      # the binf_projection machinery (fixed output map, A = 2 nf)
        self.feat = binf2phone                               # feature table of the token feed (both binary decoders)
        if binf2phone is not None and not sigmoid and d.bottom_only and d.num_layers > 1 and d.num_units < 2 * int(binf2phone.shape[0]):
            # a multi-layer --bottom_only decoder outputs the TOP CELL's h (AttentionMultiCell, las/model.py:36-69); the reference's
            # transform_binf_to_phones slices its first 2*binf_count columns (utils/training_helper.py:19-21) and fails in the
            # matmul, when the graph is built, if there are fewer
            raise ValueError('binf_projection on a multi-layer --bottom_only decoder reads [lp1 | lp0] from the first 2*binf_count = %d '
                             'columns of the top cell\'s output: decoder_units must be at least that (got %d)'
                             % (2 * int(binf2phone.shape[0]), d.num_units))
        if binf2phone is not None and d.embedding_size:
            raise ValueError('binf_projection with embedding_size > 0: the reference embeds with target_embedding and '
                             'ignores the feature vectors; not implemented on the HIP path')
        self.hp = d
        self.att = att_code
        self.additive = att_code in hip.ATT_ADDITIVE          # v . tanh(keys + Wq h) scores
        self.uses_wq = att_code in hip.ATT_USES_WQ            # query_layer in front of the score
        self.mono = att_code in hip.ATT_MONOTONIC
        self.custom = att_code == hip.ATT_CUSTOM
        self.NL, self.bottom = d.num_layers, bool(d.bottom_only)
        self.V, self.Vp = d.target_vocab_size, _r8(d.target_vocab_size)
        self.Hd, self.M = d.num_units, memory_depth
        self.A = d.attention_layer_size or self.M
        self.has_al = bool(d.attention_layer_size)
        self.emb = bool(d.embedding_size)
        self.E = d.embedding_size if self.emb else self.V
        if self.binf is not None:
            self.nf = int(self.binf.shape[0])
            self.A, self.has_al = 2 * self.nf, True
            self.emb, self.E = True, self.nf          # a constant embedding table: rows of Mb^T
        if self.sigmoid:
            if d.embedding_size:
                raise ValueError('the sigmoid-output decoder with embedding_size > 0 is not implemented on the HIP path')
            self.nf = int(d.binf_count)
            if self.feat is not None and int(self.feat.shape[0]) != self.nf:
                raise ValueError('binf2phone has %d rows, binf_count is %d' % (int(self.feat.shape[0]), self.nf))
            self.emb, self.E = True, self.nf          # the feature vector itself is the decoder input
        # width of the projection layer's output: phones, or binary features for the sigmoid-output decoder
        self.Vo = self.nf if self.sigmoid else self.V
        self.Vop = _r8(self.Vo)
        for n, v in (('decoder_units', self.Hd), ('attention depth', self.A)):      # the token width is zero-padded
            if v % 8:
                raise ValueError('%s must be a multiple of 8 on the HIP path' % n)
        self.P = self.Hd if (self.bottom and self.NL > 1) else self.A
        # Dense token feed (embedding / binary features) under input dropout: the DropoutWrapper mask is element-wise on
        # the embedded token, so it cannot be folded into a per-token row table; the (zero-padded) token vector then
        # travels as the first T0 columns of cell 0's GEMM operand: X_0 = [emb(y) | attention_{t-1} | h_{t-1}].
        self.tokx = self.emb and ((d.dropout or 0.0) > 0 or self.sigmoid)     # (sigmoid: inference feeds arbitrary vectors)
        self.T0 = _r8(self.E) if self.tokx else 0
        Hd, A = self.Hd, self.A
        # GEMM input width of each cell (without the token part of cell 0) and where its pieces sit
        self.win = []
        for l in range(self.NL):
            if l == 0:
                self.win.append(self.T0 + A)               # [(token) att_{t-1}]    + h_0
            elif self.bottom:
                self.win.append((A if l == 1 else Hd) + A)   # [cur, att_{t-1}]     + h_l
            else:
                self.win.append(Hd)                        # [h_{l-1}]              + h_l
        bf, dev = torch.bfloat16, 'cuda'
        self.kT = [torch.empty(4 * Hd, w + Hd, dtype=bf, device=dev) for w in self.win]
        self.kn = [torch.empty(w + Hd, 4 * Hd, dtype=bf, device=dev) for w in self.win]
        # cell 0's rows below the token rows in matrix-core B-fragment order: d(feed)_t = dz_t K^T of the one-launch backward
        self.kn_pk = (torch.empty(-(-(self.win[0] + Hd) // 16) * 16 * 4 * Hd, dtype=bf, device=dev)
                      if self.NL == 1 and not self.tokx else None)
        self.tok = torch.empty(self.V, 4 * Hd, dtype=bf, device=dev)
        self.wmemT = torch.empty(Hd, self.M, dtype=bf, device=dev)
        self.wmem = torch.empty(self.M, Hd, dtype=bf, device=dev)
        self.wprojT = torch.empty(self.Vop, self.P, dtype=bf, device=dev)
        self.wproj = torch.empty(self.P, self.Vop, dtype=bf, device=dev)
        self.bproj = torch.zeros(self.Vop, dtype=torch.float32, device=dev)
        if self.has_al:
            self.walT = torch.empty(A, Hd + self.M, dtype=bf, device=dev)
            self.waln = torch.empty(Hd + self.M, A, dtype=bf, device=dev)
            self.waln_pk = torch.empty((Hd + self.M) * (-(-A // 32) * 32), dtype=bf, device=dev)   # B-fragment image (one-launch backward)
        self.Ep = _r8(self.E)                 # GEMM width of the embedded token (zero padded)
        if self.emb:
            self.emb_bf = torch.zeros(self.V, self.Ep, dtype=bf, device=dev)
            self.k0tokT = torch.zeros(4 * Hd, self.Ep, dtype=bf, device=dev)
            self.k0tok = torch.zeros(self.Ep, 4 * Hd, dtype=bf, device=dev)
        if self.feat is not None:
            self.emb_bf[:, :self.nf].copy_(self.feat.to(device=dev, dtype=torch.float32).t())
        if self.binf is not None:
            Mb = self.binf.to(device=dev, dtype=torch.float32)
            wb = torch.cat([Mb, 1.0 - Mb], 0)                         # [2nf, V]: logits = [lp1 | lp0] Wb, [lp1 | lp0] = the first
            self.wproj.zero_()                                        # 2nf columns of the decoder output (all of them, unless the
            self.wprojT.zero_()                                       # output is the top cell's h of a --bottom_only stack)
            self.wproj[:2 * self.nf, :self.V].copy_(wb)
            self.wprojT[:self.V, :2 * self.nf].copy_(wb.t())
        if self.uses_wq:
            self.wq = torch.empty(Hd, Hd, dtype=bf, device=dev)
            self.wq_t = torch.empty(Hd, Hd, dtype=bf, device=dev)
            self.wq_pk = torch.empty(Hd * Hd, dtype=bf, device=dev) if Hd % 32 == 0 else None      # B-fragment image (one-launch backward)
            self.wq_pkT = torch.empty(Hd * Hd, dtype=bf, device=dev) if Hd % 32 == 0 else None     # ... of the transpose (one-launch forward)
        self._variables = variables
        self.refresh(variables)
        self.saved = None
        self.debug_hook = None

    DEC_STREAM = 1                       # token-scale draws of cell 0 (decoder index scheme of las_dec_step)
    # the sequential backward decoder runs four co-operating workgroups per utterance while that has never timed out in this
    # process (the four parts need CUs of their own at the same time; LasModel.check_device_status clears this when a
    # launch reported status bit 32, and the launches after it take one workgroup per utterance)
    SEQ_FOUR_PARTS = True

    @staticmethod
    def in_stream(l, t):
        """generator stream of the input mask of cell l at step t (element index b*win_l + c)."""
        return 64 + l * 4096 + t

    def cell_names(self, l):
        return ('%s/decoder_cell_%d/lstm_cell/kernel' % (self.scope, l), '%s/decoder_cell_%d/lstm_cell/bias' % (self.scope, l))

    def refresh(self, var):
        Hd, M, V, Vp, E, A = self.Hd, self.M, self.V, self.Vp, self.E, self.A
        Vo, Vop = self.Vo, self.Vop
        if self.binf_var is not None:      # trainable feature map: token feed and output map from the current values
            Mb = var[self.binf_var]
            self.emb_bf[:, :self.nf].copy_(Mb.t())
            wb = torch.cat([Mb, 1.0 - Mb], 0)
            self.wproj[:2 * self.nf, :V].copy_(wb)
            self.wprojT[:V, :2 * self.nf].copy_(wb.t())
        hip.cast_bf16(var[self.K_MEM], M, Hd, self.wmemT, Hd, M, transpose=True)
        hip.cast_bf16(var[self.K_MEM], M, Hd, self.wmem, M, Hd)
        self.bias = []
        for l in range(self.NL):
            k = var[self.cell_names(l)[0]]
            skip = E if l == 0 else 0
            rows = self.win[l] + Hd - (self.T0 if l == 0 else 0)
            assert k.shape == (skip + rows, 4 * Hd), (l, tuple(k.shape), skip + rows)
            if l == 0 and self.tokx:     # rows [0,E) token, zero pad to T0, then attention feed and h
                T0, W0 = self.T0, self.win[0] + Hd
                self.kT[0].zero_()
                self.kn[0].zero_()
                hip.cast_bf16(k, E, 4 * Hd, self.kT[0], 4 * Hd, E, ldd=W0, transpose=True, lds=4 * Hd)
                hip.cast_bf16(k[E:], rows, 4 * Hd, self.kT[0][:, T0:], 4 * Hd, rows, ldd=W0, transpose=True, lds=4 * Hd)
                hip.cast_bf16(k, E, 4 * Hd, self.kn[0], E, 4 * Hd, lds=4 * Hd)
                hip.cast_bf16(k[E:], rows, 4 * Hd, self.kn[0][T0:], rows, 4 * Hd, lds=4 * Hd)
            else:
                hip.cast_bf16(k[skip:], rows, 4 * Hd, self.kT[l], 4 * Hd, rows, transpose=True, lds=4 * Hd)
                hip.cast_bf16(k[skip:], rows, 4 * Hd, self.kn[l], rows, 4 * Hd, lds=4 * Hd)
                if l == 0 and self.kn_pk is not None:
                    hip.pack_mfma_b(k[skip:], rows, 4 * Hd, self.kn_pk, lds=4 * Hd)
            self.bias.append(var[self.cell_names(l)[1]])
        k0 = var[self.cell_names(0)[0]]
        Ep = self.Ep
        if self.emb:
            if self.feat is None and not self.sigmoid:
                hip.cast_bf16(var[self.K_EMB], V, E, self.emb_bf, V, Ep)
            hip.cast_bf16(k0, E, 4 * Hd, self.k0tokT, 4 * Hd, Ep, transpose=True, lds=4 * Hd)
            hip.cast_bf16(k0, E, 4 * Hd, self.k0tok, Ep, 4 * Hd, lds=4 * Hd)
            # rows the cell adds for token v: embedding[v] * K0[:E]  (the embedded feed as a [V,4Hd] table)
            hip.gemm_nt(self.emb_bf, self.k0tokT, self.tok, V, 4 * Hd, Ep, lda=Ep, ldb=Ep, ldc=4 * Hd, out_bf16=True)
        else:
            hip.cast_bf16(k0, V, 4 * Hd, self.tok, V, 4 * Hd, lds=4 * Hd)
        P = self.P
        if self.binf is None:            # binf_projection: the fixed map set up in __init__; kernel/bias are not applied
            hip.cast_bf16(var[self.K_PROJ], P, Vo, self.wprojT, Vop, P, transpose=True)
            hip.cast_bf16(var[self.K_PROJ], P, Vo, self.wproj, P, Vop)
            self.bproj[:Vo].copy_(var[self.B_PROJ])
        if self.has_al:
            hip.cast_bf16(var[self.K_AL], Hd + M, A, self.walT, A, Hd + M, transpose=True)
            hip.cast_bf16(var[self.K_AL], Hd + M, A, self.waln, Hd + M, A)
            hip.pack_mfma_b(var[self.K_AL], Hd + M, A, self.waln_pk)
        if self.uses_wq:
            hip.cast_bf16(var[self.K_Q], Hd, Hd, self.wq, Hd, Hd)
            hip.cast_bf16(var[self.K_Q], Hd, Hd, self.wq_t, Hd, Hd, transpose=True)
            if self.wq_pk is not None:
                hip.pack_mfma_b(var[self.K_Q], Hd, Hd, self.wq_pk)
                hip.pack_mfma_b(var[self.K_Q], Hd, Hd, self.wq_pkT, transpose=True)
        if self.additive:
            self.att_v = var[self.V_ATT]
        if self.mono:
            self.score_bias = var[self.B_SCORE]

    # ------------------------------------------------------------------------------------------------------------------
    def _init_states(self, encoder_state, B):
        """zip(zero_state, encoder_state) of las/model.py:259-268: cell 0 <- encoder fw, cell 1 <- encoder bw."""
        dev = 'cuda'
        z = torch.zeros(B, self.Hd, dtype=torch.float32, device=dev)
        init = [(z, z) for _ in range(self.NL)]
        passed = 0
        if self.hp.pass_hidden_state and self.bottom:
            es = list(encoder_state) if isinstance(encoder_state[0], tuple) else [encoder_state]
            if not hasattr(es[0], 'c'):
                raise ValueError('pass_hidden_state needs the pyramidal listener')
            if self.NL > len(es):
                raise ValueError('pass_hidden_state: decoder_layers (%d) exceeds the encoder states (%d)' % (self.NL, len(es)))
            for l in range(self.NL):
                if es[l].c.shape[-1] != self.Hd:
                    raise ValueError('pass_hidden_state needs decoder_units == encoder_units')
                init[l] = (es[l].c, es[l].h)
            passed = self.NL
        return init, passed

    def _keys(self, memory):
        """keys = memory_layer(memory) (Dense, no bias; las/model.py:168-169), relu'd for CustomAttention (:97)."""
        B, Tm, M = memory.shape
        keys = torch.empty(B, Tm, self.Hd, dtype=torch.bfloat16, device=memory.device)
        hip.gemm_nt(memory, self.wmemT, keys, B * Tm, self.Hd, M, lda=M, ldb=M, ldc=self.Hd, out_bf16=True)
        if self.custom:
            hip.check(hip.lib().las_relu_bf16(hip.p(keys), keys.numel(), hip.stream()))
        return keys

    def _norm(self, train):
        """alignment normaliser: softmax, or monotonic 'parallel' / 'hard' as las/model.py:157-164 selects them."""
        if not self.mono:
            return hip.NORM_SOFTMAX
        if self.att == hip.ATT_BAHDANAU_MONOTONIC and not train:
            return hip.NORM_MONOTONIC_HARD
        return hip.NORM_MONOTONIC_PARALLEL

    def _cell_fwd(self, l, t, sv, z, tok_ids, tok_stride, h2=None):
        """h2: (address, row stride) of a second place for h_t (the next step's operand row): written by the kernel, not by a copy."""
        B, Hd, U = sv['B'], self.Hd, sv['U']
        s = hip.DecStep()
        s.B, s.Hd, s.M, s.Tm, s.attention, s.mode = B, Hd, self.M, sv['Tm'], self.att, hip.DEC_CELL_ONLY
        s.z, s.bias = hip.addr(z), hip.addr(self.bias[l])
        if l == 0 and not self.tokx:
            s.tok_rows, s.tok_ids, s.tok_stride = hip.addr(self.tok), tok_ids, tok_stride
        s.c_prev, s.ldcp = hip.addr(sv['cs'][l], t * Hd), (U + 1) * Hd
        s.gates_out, s.ldg = hip.addr(sv['gates'][l], t * 4 * Hd), U * 4 * Hd
        s.c_out, s.ldco = hip.addr(sv['cs'][l], (t + 1) * Hd), (U + 1) * Hd
        s.h_out, s.ldh = hip.addr(sv['h'][l], t * Hd), U * Hd
        if h2 is not None:
            s.h_out2, s.ldh2 = h2
        s.drop_keep, s.feed_width = 1.0, self.E + self.A
        if l == 0 and sv['keep'] < 1.0 and not self.tokx:   # the one-hot token entry survives with probability keep (scaled 1/keep)
            s.drop_keep, s.drop_seed, s.drop_stream, s.step = sv['keep'], sv['seed'], self.DEC_STREAM, t
        hip.check(hip.lib().las_decoder_step_fwd(C.byref(s), 1, hip.stream()))

    def _attention_fwd(self, t, sv, query, ldq, qc=None):
        """qc: (address, row stride) of this step's [query | context] operand row of the attention layer: the kernel writes both
        halves itself."""
        B, Hd, U, Tm, M = sv['B'], self.Hd, sv['U'], sv['Tm'], self.M
        Tmp = _r8(Tm)
        s = hip.DecStep()
        s.B, s.Hd, s.M, s.Tm, s.attention, s.mode = B, Hd, M, Tm, self.att, hip.DEC_ATTENTION_ONLY
        s.query, s.ldq = query, ldq
        s.keys, s.values, s.mem_len = hip.addr(sv['keys']), hip.addr(sv['memory']), hip.addr(sv['mem_len'])
        if self.uses_wq:
            s.wq = hip.addr(self.wq)
            s.pq_out, s.ldpq = hip.addr(sv['pq'], t * Hd), U * Hd
        if self.additive:
            s.att_v = hip.addr(self.att_v)
        s.align_out, s.align_bf16, s.lda = hip.addr(sv['align'], t * Tmp), hip.addr(sv['align_bf'], t * Tmp), U * Tmp
        s.ctx_out, s.ldc = hip.addr(sv['ctx'], t * M), U * M
        if qc is not None:
            s.h_out2, s.ldh2 = qc[0], qc[1]
            s.ctx_out2, s.ldc2 = qc[0] + 2 * Hd, qc[1]
        s.drop_keep, s.feed_width, s.step = 1.0, self.E + self.A, t
        s.norm = sv.get('norm', hip.NORM_SOFTMAX)
        if self.mono:
            s.score_bias = hip.addr(self.score_bias)
            if t > 0:
                s.prev_align, s.ldpa = hip.addr(sv['align'], (t - 1) * Tmp), U * Tmp
            if sv.get('p') is not None:
                s.p_out, s.ldp = hip.addr(sv['p'], t * Tmp), U * Tmp
            s.noise_scale, s.noise_seed, s.noise_stream = sv.get('noise_scale', 0.0), sv.get('seed', 0), self.NOISE_STREAM
        hip.check(hip.lib().las_decoder_step_fwd(C.byref(s), 4, hip.stream()))

    def forward_train(self, memory, mem_len, encoder_state, targets_inputs, num_steps, parts=4, seed=0, input_vectors=None):
        """input_vectors [B,>=U,nf] (sigmoid-output decoder only): the decoder inputs as feature VECTORS, the form
        las_model_fn hands them to las.model.speller (model_helper.py:199-200,222), instead of token ids."""
        B, Tm, M = memory.shape
        if input_vectors is not None:
            if not self.sigmoid:
                raise ValueError('input_vectors: only the sigmoid-output decoder is fed feature vectors')
            input_vectors = input_vectors.to(torch.bfloat16)
            targets_inputs = torch.zeros(B, num_steps, dtype=torch.int32, device=memory.device)     # (unused placeholder ids)
        Hd, V, Vp, U, A, NL = self.Hd, self.V, self.Vop, num_steps, self.A, self.NL      # Vp: padded projection width
        dev, bf, f32 = memory.device, torch.bfloat16, torch.float32
        Tmp = _r8(Tm)
        if self.sigmoid and self.feat is None and input_vectors is None:
            raise ValueError('training the sigmoid-output decoder needs the binf2phone map (the targets\' feature vectors, '
                             'model_helper.py:199-200)')
        init, passed = self._init_states(encoder_state, B)
        keys = self._keys(memory)
        keep = 1.0 - float(self.hp.dropout or 0.0)
        sampling = float(self.hp.sampling_probability or 0.0)
        sv = dict(B=B, Tm=Tm, U=U, memory=memory, mem_len=mem_len, keys=keys, passed=passed, tin=targets_inputs,
                  keep=keep, seed=seed)
        sv['X'] = [torch.empty(B, U, w + Hd, dtype=bf, device=dev) for w in self.win]
        sv['gates'] = [torch.empty(B, U, 4 * Hd, dtype=f32, device=dev) for _ in range(NL)]
        sv['cs'] = [torch.empty(B, U + 1, Hd, dtype=f32, device=dev) for _ in range(NL)]
        sv['h'] = [torch.empty(B, U, Hd, dtype=bf, device=dev) for _ in range(NL)]
        sv['align'] = torch.empty(B, U, Tmp, dtype=f32, device=dev)
        sv['align_bf'] = torch.empty(B, U, Tmp, dtype=bf, device=dev)
        sv['ctx'] = torch.empty(B, U, M, dtype=bf, device=dev)
        sv['pq'] = torch.empty(B, U, Hd, dtype=f32, device=dev) if self.uses_wq else None
        sv['norm'] = self._norm(True)
        if self.mono:                    # p_choose of every step (backward), sigmoid_noise = 1 for bahdanau_monotonic TRAIN
            sv['p'] = torch.empty(B, U, Tmp, dtype=f32, device=dev)
            sv['noise_scale'] = 1.0 if self.att == hip.ATT_BAHDANAU_MONOTONIC else 0.0
        hip.fill_many(zero=sv['X'] + [sv['align'], sv['align_bf']] + ([sv['p']] if self.mono else []))    # one launch, not one per buffer
        sv['att'] = torch.empty(B, U, A, dtype=bf, device=dev) if self.has_al else sv['ctx']
        sv['qc'] = torch.empty(B, U, Hd + M, dtype=bf, device=dev) if self.has_al else None
        for l in range(NL):
            sv['cs'][l][:, 0].copy_(init[l][0])
            sv['X'][l][:, 0, self.win[l]:].copy_(init[l][1])        # h_{l,-1}; attention_{-1} = 0
        z = torch.empty(B, 4 * Hd, dtype=f32, device=dev)
        tin = targets_inputs
        fed = tin
        logits = None
        if self._persist_ok(B, Tm, keep, sampling, input_vectors):
            return self._forward_train_persist(sv, init, targets_inputs)
        if self._persist2_ok(B, Tm, keep, sampling, input_vectors):
            return self._forward_train_persist(sv, init, targets_inputs, two=True)
        if sampling > 0.0:               # scheduled sampling (utils/training_helper.py:48-87)
            fed = tin[:, :U].contiguous().clone()
            logits = torch.empty(B, U, Vp, dtype=f32, device=dev)
        sv['fed'] = fed
        lib, st = hip.lib(), hip.stream()
        X, h, att = sv['X'], sv['h'], sv['att']
        qlayer = 0 if self.bottom else NL - 1                       # the cell whose output queries the attention
        for t in range(U):
            last = t + 1 == U

            if self.tokx and not (self.sigmoid and sampling > 0.0 and t > 0):
                # embedded token of this step into the operand (dropped with the rest below); the sigmoid-output decoder
                # under scheduled sampling got its input vector from las_sample_features at the end of step t-1
                if input_vectors is not None:
                    X[0][:, t, :self.nf] = input_vectors[:, t, :self.nf]
                else:
                    X[0][:, t, :self.Ep] = self.emb_bf[fed[:, t].long()]

            def run_cell(l):
                Kl = self.win[l] + Hd
                if keep < 1.0:            # DropoutWrapper on this cell's input: drop the GEMM operand in place
                    hip.check(lib.las_dropout_bf16(hip.addr(X[l], t * Kl), U * Kl, hip.addr(X[l], t * Kl), U * Kl, B,
                                                   self.win[l], keep, seed, self.in_stream(l, t), st))
                hip.gemm_nt(X[l][:, t], self.kT[l], z, B, 4 * Hd, Kl, lda=U * Kl, ldb=Kl, ldc=4 * Hd)
                # (h_t also lands in the next step's operand row: the recurrent input, written by the cell kernel)
                self._cell_fwd(l, t, sv, z, hip.addr(fed, t), fed.stride(0),
                               h2=None if last else (hip.addr(X[l], (t + 1) * Kl + self.win[l]), U * Kl))

            run_cell(0)
            if not self.bottom:
                for l in range(1, NL):
                    X[l][:, t, :Hd].copy_(h[l - 1][:, t])
                    run_cell(l)
            # (the attention kernel leaves [query | context] in the attention layer's operand row itself: no copies)
            self._attention_fwd(t, sv, hip.addr(h[qlayer], t * Hd), U * Hd,
                                qc=(hip.addr(sv['qc'], t * (Hd + M)), U * (Hd + M)) if self.has_al else None)
            # Dense(A, no bias) on [query, context].  Where nothing else needs attention_t at once -- one cell, no input dropout
            # (which is applied to the operand row in place), no scheduled sampling -- the product writes it straight into the
            # next step's operand row and `att` is gathered from there once after the loop
            direct = self.has_al and NL == 1 and keep >= 1.0 and sampling == 0.0 and not last
            if self.has_al:                                           # Dense(A, no bias) on [query, context]
                if direct:
                    K0 = self.win[0] + Hd
                    hip.gemm_nt(sv['qc'][:, t], self.walT, X[0][:, t + 1, self.T0:], B, A, Hd + M, lda=U * (Hd + M), ldb=Hd + M,
                                ldc=U * K0, out_bf16=True)
                else:
                    hip.gemm_nt(sv['qc'][:, t], self.walT, att[:, t], B, A, Hd + M, lda=U * (Hd + M), ldb=Hd + M, ldc=U * A,
                                out_bf16=True)
            if self.bottom:
                for l in range(1, NL):
                    wc = A if l == 1 else Hd
                    X[l][:, t, :wc].copy_(att[:, t] if l == 1 else h[l - 1][:, t])
                    # X[l][:, t, wc:wc+A] already holds attention_{t-1} (written at the end of step t-1; zero at t=0)
                    run_cell(l)
            if not last and not direct:
                X[0][:, t + 1, self.T0:self.T0 + A].copy_(att[:, t])
                if self.bottom:
                    for l in range(1, NL):
                        wc = A if l == 1 else Hd
                        X[l][:, t + 1, wc:wc + A].copy_(att[:, t])
            if sampling > 0.0:
                out_t = h[NL - 1][:, t] if (self.bottom and NL > 1) else att[:, t]
                hip.gemm_nt(out_t, self.wprojT, logits[:, t], B, Vp, self.P, lda=out_t.stride(0), ldb=self.P, ldc=U * Vp,
                            bias=self.bproj)
                if not last and self.sigmoid:
                    # ScheduledSigmoidHelper: Bernoulli(sigmoid(logits)) feature draws or the teacher's feature vector
                    if input_vectors is not None:
                        teach = torch.zeros(B, self.Ep, dtype=bf, device=dev)
                        teach[:, :self.nf] = input_vectors[:, t + 1, :self.nf]
                    else:
                        teach = self.emb_bf[tin[:, t + 1].long()]
                    hip.check(lib.las_sample_features(hip.addr(logits, t * Vp), U * Vp, self.nf, hip.p(teach), self.Ep,
                                                      hip.addr(X[0], (t + 1) * (self.win[0] + Hd)), U * (self.win[0] + Hd), B,
                                                      sampling, seed, t, st))
                elif not last:
                    hip.check(lib.las_sample_tokens(hip.addr(logits, t * Vp), U * Vp, V, hip.addr(tin, t + 1), tin.stride(0),
                                                    hip.addr(fed, t + 1), fed.stride(0), B, sampling, seed, t, st))
        if self.has_al and NL == 1 and keep >= 1.0 and sampling == 0.0 and U > 1:
            att[:, :U - 1].copy_(X[0][:, 1:, self.T0:self.T0 + A])       # the steps that wrote into the operand rows (see above)
        out_all = h[NL - 1] if (self.bottom and NL > 1) else att
        sv['out'] = out_all
        if logits is None:
            logits = torch.empty(B, U, Vp, dtype=f32, device=dev)
            hip.gemm_nt(out_all, self.wprojT, logits, B * U, Vp, self.P, lda=self.P, ldb=self.P, ldc=Vp, bias=self.bproj)
        self.saved = sv
        self.last_Tm = Tm            # memory length of the last forward (tests replay the score noise)
        return logits

    # ------------------------------------------------------------------------------------------------------------------
    # one-launch forward (round 3): single cell + attention layer and / or monotonic normaliser (cfg5: --binf_projection with
    # bahdanau_monotonic; --attention_layer_size models).  All U steps in ONE las_decoder_persist_fwd launch; the backward
    # still runs step by step on what this leaves behind.
    def _persist_workspace(self, which, nbytes):
        cache = self.__dict__.setdefault('_persist_cache', {})
        ws = cache.get(which)
        if ws is None or ws.numel() < nbytes:
            if ws is not None and int(ws[:4].view(torch.int32).item()):
                raise hip.LasError('persistent decoder reported a barrier timeout')
            ws = cache[which] = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
        return ws

    def _persist_ok(self, B, Tm, keep, sampling, input_vectors):
        import os
        if os.environ.get('LAS_DEC_PERSIST', '1') == '0' or os.environ.get('LAS_DEC_PERSIST_AL', '1') == '0':
            return False
        if self.NL != 1 or self.tokx or self.sigmoid or keep < 1.0 or sampling > 0.0 or input_vectors is not None:
            return False
        if not (self.has_al or self.mono):
            return False
        lib = hip.lib()
        Kp = (self.win[0] + self.Hd + 63) // 64 * 64
        return (B <= 4 * lib.las_decoder_persist_max_batch() and
                lib.las_decoder_persist_al_supported(self.Hd, self.M, Kp, self.A if self.has_al else 0, self.att, self._norm(True)) == 1)

    def _persist2_ok(self, B, Tm, keep, sampling, input_vectors):
        """Two decoder cells (the reference's default depth) in ONE forward launch (round 4): both wirings, softmax attentions,
        one-hot or embedded tokens (embedding_size > 0: a row table without dropout, the dense token vector in the operand
        row with it), no attention layer; input dropout and scheduled sampling (the reference's defaults: dropout 0.2,
        sampling_probability 0.1) inside the launch.  LAS_DEC_PERSIST2=0: step by step."""
        import os
        if os.environ.get('LAS_DEC_PERSIST', '1') == '0' or os.environ.get('LAS_DEC_PERSIST2', '1') == '0':
            return False
        if self.NL != 2 or self.sigmoid or self.has_al or self.mono or self.custom or self.binf is not None:
            return False
        if input_vectors is not None or self.Vop > 1024 or (keep < 1.0 and (self.win[0] != self.T0 + self.M or self.win[1] % 8)):
            return False
        lib = hip.lib()
        Kp = (self.win[0] + self.Hd + 63) // 64 * 64
        return (B <= 4 * lib.las_decoder_persist_max_batch() and
                lib.las_decoder_persist2_supported(self.Hd, self.M, Kp, self.win[1] + self.Hd, self.att, 1 if self.bottom else 0) == 1)

    def _forward_train_persist(self, sv, init, targets_inputs, two=False):
        B, Tm, U, M = sv['B'], sv['Tm'], sv['U'], self.M
        Hd, A, Vp = self.Hd, self.A, self.Vop
        dev, bf, f32 = sv['memory'].device, torch.bfloat16, torch.float32
        lib, st = hip.lib(), hip.stream()
        Tmp = _r8(Tm)
        W0 = self.win[0] + Hd                                  # [(token) | attention_{t-1} | h_{t-1}]: the cell kernel's rows behind the token rows
        Kp = (W0 + 63) // 64 * 64                              # operand rows of whole 128-byte lines
        T0 = self.T0 if two else 0                             # dense token vector in front (embedding_size > 0 under dropout)
        feed = T0 + (A if self.has_al else M)                  # column of h: behind what is fed back (attention_t, or the context itself)
        Xp = torch.zeros(B, U, Kp, dtype=bf, device=dev)
        Xp[:, 0, feed:feed + Hd].copy_(init[0][1])
        sv['cs'][0][:, 0].copy_(init[0][0])
        if getattr(self, '_kTp', None) is None or self._kTp.shape[1] != Kp:
            self._kTp = torch.zeros(4 * Hd, Kp, dtype=bf, device=dev)
        self._kTp[:, :W0].copy_(self.kT[0])                    # (the image follows the weights: refreshed every step anyway)
        fed = targets_inputs[:, :U].to(torch.int32).contiguous()
        keep, sampling = (sv['keep'], float(self.hp.sampling_probability or 0.0)) if two else (1.0, 0.0)
        if sampling > 0.0 and fed.data_ptr() == targets_inputs.data_ptr():
            fed = fed.clone()                                  # the launch writes the sampled tokens into the fed ids
        sv['fed'] = fed
        p = hip.DecPersist()
        s = p.s
        s.B, s.Hd, s.M, s.Tm, s.attention, s.mode = B, Hd, M, Tm, self.att, 0
        s.tok_rows, s.tok_ids, s.tok_stride = (0 if T0 else hip.addr(self.tok)), hip.addr(fed), fed.stride(0)
        s.bias = hip.addr(self.bias[0])
        s.c_prev, s.ldcp = hip.addr(sv['cs'][0]), (U + 1) * Hd
        s.gates_out, s.ldg = hip.addr(sv['gates'][0]), U * 4 * Hd
        s.c_out, s.ldco = hip.addr(sv['cs'][0], Hd), (U + 1) * Hd
        s.h_out, s.ldh = hip.addr(sv['h'][0]), U * Hd
        s.h_out2, s.ldh2 = (hip.addr(Xp, Kp + feed) if U > 1 else 0), U * Kp
        s.keys, s.values, s.mem_len = hip.addr(sv['keys']), hip.addr(sv['memory']), hip.addr(sv['mem_len'])
        if self.uses_wq:
            s.wq = hip.addr(self.wq)
            s.pq_out, s.ldpq = hip.addr(sv['pq']), U * Hd
        if self.additive:
            s.att_v = hip.addr(self.att_v)
        s.align_out, s.align_bf16, s.lda = hip.addr(sv['align']), hip.addr(sv['align_bf']), U * Tmp
        s.ctx_out, s.ldc = hip.addr(sv['ctx']), U * M
        if not self.has_al:                                    # the context is the feed: straight into the next operand row
            s.ctx_out2, s.ldc2 = (hip.addr(Xp, Kp + T0) if U > 1 else 0), U * Kp
        s.drop_keep, s.feed_width = 1.0, self.E + A
        if keep < 1.0:                                         # the token row's scale: cell 0's draws of the step-by-step path
            s.drop_keep, s.drop_seed, s.drop_stream = keep, sv['seed'], self.DEC_STREAM
            p.win0, p.win1, p.in_stream0, p.in_stream1 = self.win[0], self.win[1], self.in_stream(0, 0), self.in_stream(1, 0)
        if T0:
            # dense token feed: the teacher's embedded tokens into the first T0 columns of every operand row, through the steps'
            # input masks (the launch rewrites a row whose token it samples); the feed's mask window is win0 wide
            Xp[:, :, :self.Ep].copy_(self.emb_bf[fed.long()])
            if keep < 1.0:
                hip.check(lib.las_dropout_bf16_steps(hip.addr(Xp), U * Kp, Kp, B, U, T0, self.win[0], keep, sv['seed'],
                                                     self.in_stream(0, 0), st))
            s.feed_width = self.win[0]
            p.emb, p.ld_emb, p.T0 = hip.addr(self.emb_bf), self.Ep, T0
        s.norm = sv['norm']
        if self.mono:
            s.score_bias = hip.addr(self.score_bias)
            s.p_out, s.ldp = hip.addr(sv['p']), U * Tmp
            s.noise_scale, s.noise_seed, s.noise_stream = sv.get('noise_scale', 0.0), sv.get('seed', 0), self.NOISE_STREAM
        if sampling > 0.0:                                     # the sampled feed is produced inside the launch
            plog = torch.empty(U, B, 4, Vp, dtype=f32, device=dev)
            p.sampling_prob, p.seed = sampling, sv['seed']
            p.teacher, p.teacher_stride = hip.addr(targets_inputs), targets_inputs.stride(0)
            p.wprojT, p.ldw, p.bproj = hip.addr(self.wprojT), self.P, hip.addr(self.bproj)
            p.logits, p.ld_logits, p.plog, p.V, p.Vp = 0, U * Vp, hip.addr(plog), self.V, Vp
        p.U, p.K_in = U, Kp
        if self.uses_wq and self.wq_pkT is not None and Hd in (128, 256) and os.environ.get('LAS_DEC_PQ_MFMA', '1') != '0':
            p.wq_packed = hip.addr(self.wq_pkT)
        p.inc_tok, p.inc_cprev, p.inc_gates, p.inc_cout, p.inc_h, p.inc_h2 = 1, Hd, 4 * Hd, Hd, Hd, Kp
        p.inc_align, p.inc_pq, p.inc_ctx, p.inc_ctx2, p.inc_p = Tmp, Hd, M, Kp, Tmp
        p.x, p.ldx, p.inc_x = hip.addr(Xp), U * Kp, Kp
        p.kT, p.ldk = hip.addr(self._kTp), Kp
        if self.has_al:
            p.walT, p.ld_wal, p.A, p.x_att_off = hip.addr(self.walT), Hd + M, A, 0
            p.att_out, p.ld_att = hip.addr(sv['att']), U * A
        if two:
            # the second cell: its kernel rows in operand order ([h0 | h1] or [attention_t | attention_{t-1} | h1]: self.kT[1] as
            # it is), its states with the initial row in front
            K1 = self.win[1] + Hd
            c1 = torch.empty(B, U + 1, Hd, dtype=f32, device=dev)
            h1 = torch.empty(B, U + 1, Hd, dtype=bf, device=dev)
            hip.fill_many(copy=[(c1[:, 0], init[1][0].float()), (h1[:, 0], init[1][1].float())])
            p.k1T, p.ldk1, p.K1_in, p.wiring = hip.addr(self.kT[1]), K1, K1, 1 if self.bottom else 0
            p.bias1, p.c1, p.gates1, p.h1 = hip.addr(self.bias[1]), hip.addr(c1), hip.addr(sv['gates'][1]), hip.addr(h1)
        ws = self._persist_workspace('fwd', lib.las_decoder_persist_workspace_bytes(B, Tm, Hd, M))
        p.workspace = hip.addr(ws)
        tok = hip.prof_begin('dec_persist_fwd', 2.0 * U * B * (Kp * 4 * Hd + Tm * Hd + Tm * M + (Hd + M) * (A if self.has_al else 0)
                                                               + ((self.win[1] + Hd) * 4 * Hd if two else 0)))
        hip.check(lib.las_decoder_persist_fwd(C.byref(p), st))
        hip.prof_end(tok)
        self._persist_ws = ws
        # what the step-by-step backward reads: the compact operand rows and the attention layer's [query | context] rows
        sv['X'][0] = Xp[:, :, :W0].contiguous()
        if two:
            # ... and the second cell's: its states in the step-by-step layout, its operand rows gathered from the pieces
            sv['cs'][1] = c1
            sv['h'][1] = h1[:, 1:].contiguous()
            if self.bottom:     # [attention_t | attention_{t-1} | h1_{t-1}] (the feed in Xp carries cell 0's mask: from the contexts)
                prev = torch.cat([torch.zeros(B, 1, M, dtype=bf, device=dev), sv['ctx'][:, :U - 1]], 1)
                sv['X'][1] = torch.cat([sv['ctx'], prev, h1[:, :U]], -1)
            else:               # [h0_t | h1_{t-1}]
                sv['X'][1] = torch.cat([sv['h'][0], h1[:, :U]], -1)
            if keep < 1.0:      # the weight-gradient products read the rows as the cells saw them: dropped (cell 0's already are)
                K1 = self.win[1] + Hd
                hip.check(lib.las_dropout_bf16_steps(hip.addr(sv['X'][1]), U * K1, K1, B, U, self.win[1], 0, keep, sv['seed'],
                                                     self.in_stream(1, 0), st))
        if self.has_al:
            sv['qc'][:, :, :Hd].copy_(sv['h'][0])
            sv['qc'][:, :, Hd:].copy_(sv['ctx'])
        out_all = sv['h'][1] if (two and self.bottom) else sv['att']
        sv['out'] = out_all
        logits = torch.empty(B, U, Vp, dtype=f32, device=dev)
        hip.gemm_nt(out_all, self.wprojT, logits, B * U, Vp, self.P, lda=self.P, ldb=self.P, ldc=Vp, bias=self.bproj)
        self.saved = sv
        self.last_Tm = Tm
        return logits

    def _seq_bwd_ok(self, sv):
        import os
        if os.environ.get('LAS_DEC_PERSIST', '1') == '0' or os.environ.get('LAS_DEC_SEQ_BWD', '1') == '0':
            return False
        if self.NL != 1 or self.tokx or self.sigmoid or sv['keep'] < 1.0 or self.debug_hook is not None:
            return False
        if not (self.has_al or self.mono):
            return False
        return hip.lib().las_decoder_seq_bwd_supported(self.Hd, self.M, self.A if self.has_al else 0, self.win[0] + self.Hd, sv['Tm'],
                                                       self.att, sv['norm']) == 1

    def _persist2_bwd_ok(self, sv):
        """Two decoder cells: all U backward steps in ONE las_decoder_persist_bwd launch (round 4) where the forward's one-launch
        kernel applies.  LAS_DEC_PERSIST2_BWD=0: step by step."""
        import os
        if os.environ.get('LAS_DEC_PERSIST', '1') == '0' or os.environ.get('LAS_DEC_PERSIST2', '1') == '0':
            return False
        if os.environ.get('LAS_DEC_PERSIST2_BWD', '1') == '0' or self.debug_hook is not None or sv.get('dreg') is not None:
            return False
        if self.NL != 2 or self.sigmoid or self.has_al or self.mono or self.custom or self.binf is not None:
            return False
        if sv['keep'] < 1.0 and (self.win[0] != self.T0 + self.M or self.win[1] % 8):
            return False
        lib = hip.lib()
        return (sv['B'] <= 4 * lib.las_decoder_persist_max_batch() and
                lib.las_decoder_persist2_bwd_supported(self.Hd, self.M, self.M + self.Hd, self.win[1] + self.Hd, self.att,
                                                       1 if self.bottom else 0) == 1)

    def log_probs_loss(self, loss, weight, grad_scale):
        """loss += weight * compute_log_probs_loss(raw outputs) (model_helper.py:132-146,327-331) on the attention vectors
        of the last forward_train; its gradient joins d(outputs) in backward()."""
        sv = self.saved
        B, U, P = sv['B'], sv['U'], self.P
        # the raw outputs: the attention vectors (P = 2 nf), or the top cell's h of a --bottom_only stack (P = decoder_units) --
        # compute_log_probs_loss halves WHATEVER width it is handed (model_helper.py:137-139: nfeatures = outputs.shape[-1] // 2)
        raw = sv['out'] if (self.bottom and self.NL > 1) else sv['att']
        sv['dreg'] = torch.zeros(B, U, P, dtype=torch.float32, device=loss.device)
        hip.check(hip.lib().las_log_probs_loss(hip.p(raw), P, B * U, P // 2, weight, grad_scale, hip.p(loss),
                                               hip.p(sv['dreg']), P, hip.stream()))

    # ------------------------------------------------------------------------------------------------------------------
    def _cell_bwd(self, l, t, sv, dc, sources, dz):
        B, Hd, U = sv['B'], self.Hd, sv['U']
        s = hip.DecStepBwd()
        s.B, s.Hd, s.M, s.Tm, s.attention, s.mode = B, Hd, self.M, sv['Tm'], self.att, hip.DEC_CELL_ONLY
        s.dc = hip.addr(dc)
        s.gates, s.ldg = hip.addr(sv['gates'][l], t * 4 * Hd), U * 4 * Hd
        s.c_new, s.ldcn = hip.addr(sv['cs'][l], (t + 1) * Hd), (U + 1) * Hd
        s.c_prev, s.ldcp = hip.addr(sv['cs'][l], t * Hd), (U + 1) * Hd
        s.dz, s.ldz = hip.addr(dz, t * 4 * Hd), U * 4 * Hd
        srcs = [x for x in sources if x is not None]
        assert 1 <= len(srcs) <= 3
        fields = (('dh_rec', 'ldr'), ('dh_b', 'ldhb'), ('dh_c', 'ldhc'))
        for (pn, ln), (ptr, ld) in zip(fields, srcs):
            setattr(s, pn, ptr)
            setattr(s, ln, ld)
        s.drop_keep, s.feed_width = 1.0, self.E + self.A
        hip.check(hip.lib().las_decoder_step_bwd(C.byref(s), hip.stream()))

    def backward(self, dlogits, grads, overlap=None):
        sv = self.saved
        B, Tm, U = sv['B'], sv['Tm'], sv['U']
        Hd, V, Vp, M, A, NL, P = self.Hd, self.V, self.Vp, self.M, self.A, self.NL, self.P
        Vo, Vop = self.Vo, self.Vop
        Tmp = _r8(Tm)
        dev, bf, f32 = dlogits.device, torch.bfloat16, torch.float32
        lib, st = hip.lib(), hip.stream()
        BU = B * U
        bah = self.additive
        d_out = torch.empty(B, U, P, dtype=f32, device=dev)
        hip.gemm_nt(dlogits, self.wproj, d_out, BU, P, Vop, lda=Vop, ldb=Vop, ldc=P)
        if sv.get('dreg') is not None:       # gradient of compute_log_probs_loss w.r.t. the raw outputs (binf_projection)
            d_out.add_(sv['dreg'])
        dc = [torch.empty(B, Hd, dtype=f32, device=dev) for _ in range(NL)]
        dx = [[torch.empty(B, w + Hd, dtype=f32, device=dev) for _ in range(2)] for w in self.win]
        dz = [torch.empty(B, U, 4 * Hd, dtype=bf, device=dev) for _ in range(NL)]
        ds_all = torch.empty(B, U, Tmp, dtype=bf, device=dev)
        dctx_all = torch.empty(B, U, M, dtype=bf, device=dev)
        datt = torch.empty(B, A, dtype=f32, device=dev)
        datt_bf = torch.empty(B, U, A, dtype=bf, device=dev) if self.has_al else None
        dqc = torch.empty(B, Hd + M, dtype=f32, device=dev) if self.has_al else None
        dq = torch.empty(B, Hd, dtype=f32, device=dev)
        if bah:
            dkeys = torch.empty(B, Tm, Hd, dtype=f32, device=dev)
        if self.uses_wq:
            dpq_all = torch.empty(B, U, Hd, dtype=bf, device=dev)
        if self.mono:                    # gradient into align_{t-1} through step t's normaliser
            carry = torch.empty(B, Tmp, dtype=f32, device=dev)
        hip.fill_many(zero=dc + [x for pair in dx for x in pair] + [ds_all] + ([dkeys] if bah else []) + ([carry] if self.mono else []))
        qlayer = 0 if self.bottom else NL - 1
        W = [w + Hd for w in self.win]
        dtokx = torch.empty(B, U, self.Ep, dtype=bf, device=dev) if (self.tokx and (self.feat is None or self.binf_var is not None) and not self.sigmoid) else None

        def v(buf, off, ld):              # (address, row stride) of a column window of a [B, ld] fp32 buffer
            return (hip.addr(buf, off), ld)

        seq = self._seq_bwd_ok(sv)
        if seq:
            # all U steps in ONE launch, one workgroup per utterance (las_decoder_seq_bwd): what the loop below does step by step
            q = hip.DecSeqBwd()
            s = q.s
            s.B, s.Hd, s.M, s.Tm, s.attention, s.mode = B, Hd, M, Tm, self.att, 0
            s.dctx_save, s.ldds = hip.addr(dctx_all), U * M
            s.dc = hip.addr(dc[0])
            s.gates, s.ldg = hip.addr(sv['gates'][0]), U * 4 * Hd
            s.c_new, s.ldcn = hip.addr(sv['cs'][0], Hd), (U + 1) * Hd
            s.c_prev, s.ldcp = hip.addr(sv['cs'][0]), (U + 1) * Hd
            s.align, s.lda = hip.addr(sv['align']), U * Tmp
            s.keys, s.values, s.mem_len = hip.addr(sv['keys']), hip.addr(sv['memory']), hip.addr(sv['mem_len'])
            s.dz, s.ldz = hip.addr(dz[0]), U * 4 * Hd
            s.ds_out, s.ldso = hip.addr(ds_all), U * Tmp
            s.drop_keep, s.feed_width = 1.0, self.E + A
            if self.uses_wq:
                s.pq, s.ldpq = hip.addr(sv['pq']), U * Hd
                s.wq_t = hip.addr(self.wq_t)
                s.dpq_out, s.lddpq = hip.addr(dpq_all), U * Hd
            if bah:
                s.att_v = hip.addr(self.att_v)
                s.dkeys_acc, s.dv_acc = hip.addr(dkeys), hip.addr(grads[self.V_ATT])
            s.norm = sv['norm']
            if self.mono:
                s.p, s.ldp = hip.addr(sv['p']), U * Tmp
                s.dalign_carry, s.ldcarry = hip.addr(carry), Tmp
                s.dbias_acc = hip.addr(grads[self.B_SCORE])
            q.U, q.A, q.W0 = U, (A if self.has_al else 0), W[0]
            q.inc_gates, q.inc_c, q.inc_align, q.inc_dz, q.inc_ds, q.inc_save, q.inc_pq = 4 * Hd, Hd, Tmp, 4 * Hd, Tmp, M, Hd
            q.d_out, q.ld_dout, q.inc_dout = hip.addr(d_out), U * P, P
            if self.has_al:
                q.datt_out, q.ld_datt = hip.addr(datt_bf), U * A
                q.waln_packed = hip.addr(self.waln_pk)
                if os.environ.get('LAS_DEC_SEQ_VW', '1') != '0':
                    # VW = values W_c [B, T', A]: d(alignments)_t = VW d(attention_t) inside the launch (no pass over the values)
                    vw = torch.empty(B, Tm, A, dtype=f32, device=dev)
                    hip.gemm_nt(sv['memory'], self.walT[:, Hd:], vw, B * Tm, A, M, lda=M, ldb=Hd + M, ldc=A)
                    q.vw, q.ld_vw = hip.addr(vw), Tm * A
            q.kn_packed = hip.addr(self.kn_pk)
            if bah or self.mono:                    # d(attention_v) / d(score_bias): fixed-order sums instead of atomics
                q.sum_workspace = hip.addr(self._persist_workspace('sum', lib.las_decoder_sum_workspace_bytes(32 * ((B + 7) // 8), Hd + 1)))
                if bah and os.environ.get('LAS_DEC_SEQ_PARTS', '4') != '1' and GeneralSpeller.SEQ_FOUR_PARTS:       # four workgroups per utterance (see las_dec_seq_bwd)
                    q.xchg_workspace = hip.addr(self._persist_workspace('seqx', lib.las_decoder_seq_xchg_bytes(B, Tm, Hd, M, W[0])))
                    if self.uses_wq and self.wq_pk is not None:
                        q.wq_packed = hip.addr(self.wq_pk)
            dfeed0 = torch.empty(B, W[0], dtype=f32, device=dev)
            q.dfeed_out = hip.addr(dfeed0)
            tok = hip.prof_begin('dec_seq_bwd', 2.0 * U * B * (W[0] * 4 * Hd + 2 * Tm * Hd + 2 * Tm * M + (Hd + M) * (A if self.has_al else 0)))
            hip.check(lib.las_decoder_seq_bwd(C.byref(q), st))
            hip.prof_end(tok)
            dx[0][0] = dfeed0                          # step 0's d[feed | h]: the gradient into the initial state
        two = (not seq) and self._persist2_bwd_ok(sv)
        if two:
            # both cells, all U steps in ONE launch (las_dec_persist_bwd with a second cell): what the loop below does step by step
            q = hip.DecPersistBwd()
            s = q.s
            s.B, s.Hd, s.M, s.Tm, s.attention, s.mode = B, Hd, M, Tm, self.att, 0
            if not self.bottom:                        # the context is the output
                s.dctx_a, s.ldda = hip.addr(d_out), U * P
            else:                                      # h1_t is
                q.d_out1, q.ld_dout1, q.inc_dout1 = hip.addr(d_out), U * P, P
            s.dctx_save, s.ldds = hip.addr(dctx_all), U * M
            s.dc = hip.addr(dc[0])
            s.gates, s.ldg = hip.addr(sv['gates'][0]), U * 4 * Hd
            s.c_new, s.ldcn = hip.addr(sv['cs'][0], Hd), (U + 1) * Hd
            s.c_prev, s.ldcp = hip.addr(sv['cs'][0]), (U + 1) * Hd
            s.align, s.lda = hip.addr(sv['align']), U * Tmp
            s.keys, s.values, s.mem_len = hip.addr(sv['keys']), hip.addr(sv['memory']), hip.addr(sv['mem_len'])
            s.dz, s.ldz = hip.addr(dz[0]), U * 4 * Hd
            s.ds_out, s.ldso = hip.addr(ds_all), U * Tmp
            s.drop_keep, s.feed_width = 1.0, self.E + A
            if sv['keep'] < 1.0:
                s.drop_keep, s.drop_seed, s.drop_stream = sv['keep'], sv['seed'], self.DEC_STREAM
                q.win0, q.win1, q.in_stream0, q.in_stream1 = self.win[0], self.win[1], self.in_stream(0, 0), self.in_stream(1, 0)
            if bah:
                s.pq, s.ldpq = hip.addr(sv['pq']), U * Hd
                s.wq_t, s.att_v = hip.addr(self.wq_t), hip.addr(self.att_v)
                s.dkeys_acc, s.dv_acc = hip.addr(dkeys), hip.addr(grads[self.V_ATT])
                s.dpq_out, s.lddpq = hip.addr(dpq_all), U * Hd
                q.sum_workspace = hip.addr(self._persist_workspace('sum', lib.las_decoder_sum_workspace_bytes(32 * (((B + 7) // 8 + 7) // 8 * 8), Hd)))
            T0 = self.T0                               # (dense token vector in front of cell 0's rows: its gradient is one product below)
            q.U, q.W = U, W[0] - T0
            q.inc_a, q.inc_save, q.inc_gates, q.inc_c, q.inc_align, q.inc_dz, q.inc_ds, q.inc_pq = P, M, 4 * Hd, Hd, Tmp, 4 * Hd, Tmp, Hd
            q.kc, q.ldk = hip.addr(self.kn[0], T0 * 4 * Hd), 4 * Hd
            dfeed0 = torch.empty(B, W[0] - T0, dtype=f32, device=dev)
            dfeed1 = torch.empty(B, W[1], dtype=f32, device=dev)
            ws = self._persist_workspace('bwd', lib.las_decoder_persist_workspace_bytes(B, Tm, Hd, M))
            q.dfeed_all, q.workspace = hip.addr(dfeed0), hip.addr(ws)
            q.k1c, q.ldk1, q.W1, q.wiring = hip.addr(self.kn[1]), 4 * Hd, W[1], 1 if self.bottom else 0
            q.gates1, q.c1, q.dz1, q.dc1 = hip.addr(sv['gates'][1]), hip.addr(sv['cs'][1]), hip.addr(dz[1]), hip.addr(dc[1])
            q.dfeed1_all = hip.addr(dfeed1)
            tok = hip.prof_begin('dec_persist_bwd', 2.0 * U * B * ((W[0] + W[1]) * 4 * Hd + Tm * Hd + Tm * M))
            hip.check(lib.las_decoder_persist_bwd(C.byref(q), st))
            hip.prof_end(tok)
            self._persist_ws_bwd = ws
            if T0:
                dfeed0 = torch.cat([torch.zeros(B, T0, dtype=f32, device=dev), dfeed0], 1)
                if dtokx is not None:
                    # gradient w.r.t. the embedded tokens of all steps: dz_0 K_0[token rows]^T, then through the steps' input masks
                    dtok = torch.empty(B, U, self.Ep, dtype=f32, device=dev)
                    hip.gemm_nt(dz[0], self.kn[0], dtok, BU, self.Ep, 4 * Hd, lda=4 * Hd, ldb=4 * Hd, ldc=self.Ep)
                    hip.cast_bf16(dtok, BU, self.Ep, dtokx, BU, self.Ep, ldd=self.Ep, lds=self.Ep)
                    if sv['keep'] < 1.0:
                        hip.check(lib.las_dropout_bf16_steps(hip.addr(dtokx), U * self.Ep, self.Ep, B, U, self.Ep, self.win[0], sv['keep'],
                                                             sv['seed'], self.in_stream(0, 0), st))
            dx[0][0], dx[1][0] = dfeed0, dfeed1       # step 0's products: the gradients into the initial states (their h columns)
        for t in (range(U - 1, -1, -1) if not (seq or two) else ()):
            cur, nxt = t & 1, (t + 1) & 1
            has_next = t + 1 < U

            def cell_and_gemm(l, sources):
                rec = v(dx[l][nxt], self.win[l], W[l]) if has_next else None
                self._cell_bwd(l, t, sv, dc[l], list(sources) + [rec], dz[l])
                hip.gemm_nt(dz[l][:, t], self.kn[l], dx[l][cur], B, W[l], 4 * Hd, lda=U * 4 * Hd, ldb=4 * Hd, ldc=W[l])
                if sv['keep'] < 1.0:      # gradient through this cell's input dropout (mask regenerated), in place
                    hip.check(lib.las_add_masked(None, 0, hip.p(dx[l][cur]), W[l], hip.p(dx[l][cur]), W[l], B, self.win[l],
                                                 sv['keep'], sv['seed'], self.in_stream(l, t), 0, self.win[l], st))

            if self.bottom:
                for l in range(NL - 1, 0, -1):
                    wc_up = (A if l + 1 == 1 else Hd)
                    src = v(d_out, t * P, U * P) if l == NL - 1 else v(dx[l + 1][cur], 0, W[l + 1])
                    cell_and_gemm(l, [src])
            # gradient w.r.t. attention_t
            if self.has_al and NL == 1:
                # d(attention_t) = d(output projection)_t + d(feed into step t+1), as the attention layer's bf16 operand: one launch
                hip.check(lib.las_add_cast_bf16(hip.addr(d_out, t * P), U * P, hip.addr(dx[0][nxt], self.T0) if has_next else None,
                                                W[0], hip.addr(datt_bf, t * A), U * A, B, A, st))
            else:
                if self.bottom and NL > 1:
                    datt.copy_(dx[1][cur][:, :A])
                else:
                    datt.copy_(d_out[:, t])
                if has_next:
                    datt.add_(dx[0][nxt][:, self.T0:self.T0 + A])
                    if self.bottom:
                        for l in range(1, NL):
                            wc = A if l == 1 else Hd
                            datt.add_(dx[l][nxt][:, wc:wc + A])
                if self.has_al:
                    hip.cast_bf16(datt, B, A, datt_bf[:, t], B, A, ldd=U * A, lds=A)
            if self.has_al:
                hip.gemm_nt(datt_bf[:, t], self.waln, dqc, B, Hd + M, A, lda=U * A, ldb=A, ldc=Hd + M)
                dctx_ptr, dctx_ld = hip.addr(dqc, Hd), Hd + M
            else:
                dctx_ptr, dctx_ld = hip.addr(datt), A
            s = hip.DecStepBwd()
            s.B, s.Hd, s.M, s.Tm, s.attention, s.mode = B, Hd, M, Tm, self.att, hip.DEC_ATTENTION_ONLY
            s.dctx_a, s.ldda = dctx_ptr, dctx_ld
            s.dctx_save, s.ldds = hip.addr(dctx_all, t * M), U * M
            s.align, s.lda = hip.addr(sv['align'], t * Tmp), U * Tmp
            s.keys, s.values, s.mem_len = hip.addr(sv['keys']), hip.addr(sv['memory']), hip.addr(sv['mem_len'])
            s.ds_out, s.ldso = hip.addr(ds_all, t * Tmp), U * Tmp
            s.dq_out, s.lddq = hip.addr(dq), Hd
            s.drop_keep, s.feed_width = 1.0, self.E + A
            if self.uses_wq:
                s.pq, s.ldpq = hip.addr(sv['pq'], t * Hd), U * Hd
                s.wq_t = hip.addr(self.wq_t)
                s.dpq_out, s.lddpq = hip.addr(dpq_all, t * Hd), U * Hd
            if bah:
                s.att_v = hip.addr(self.att_v)
                s.dkeys_acc, s.dv_acc = hip.addr(dkeys), hip.addr(grads[self.V_ATT])
            s.norm = sv['norm']
            if self.mono:
                s.p, s.ldp = hip.addr(sv['p'], t * Tmp), U * Tmp
                if t > 0:
                    s.prev_align, s.ldpa = hip.addr(sv['align'], (t - 1) * Tmp), U * Tmp
                s.dalign_carry, s.ldcarry = hip.addr(carry), Tmp
                s.dbias_acc = hip.addr(grads[self.B_SCORE])
            hip.check(lib.las_decoder_step_bwd(C.byref(s), st))
            qsrc = [v(dq, 0, Hd)] + ([v(dqc, 0, Hd + M)] if self.has_al else [])
            if self.bottom:
                cell_and_gemm(0, qsrc)
            else:
                cell_and_gemm(NL - 1, qsrc)
                for l in range(NL - 2, -1, -1):
                    cell_and_gemm(l, [v(dx[l + 1][cur], 0, W[l + 1])])
            if dtokx is not None:         # gradient w.r.t. the embedded token of this step (after its dropout mask)
                hip.cast_bf16(dx[0][cur], B, self.Ep, dtokx[:, t], B, self.Ep, ldd=U * self.Ep, lds=W[0])
            if self.debug_hook is not None:      # diagnostics (scripts/gpu_cfg5_nan.py): look at the step's tensors
                self.debug_hook(t, dict(d_out=d_out[:, t], datt_bf=datt_bf[:, t] if datt_bf is not None else datt, dqc=dqc, dq=dq,
                                        ds=ds_all[:, t], dctx=dctx_all[:, t], dz=dz[0][:, t], dx=dx[0][cur], dc=dc[0],
                                        carry=carry if self.mono else None, dkeys=dkeys if bah else None,
                                        dpq=dpq_all[:, t] if self.uses_wq else None))
        # ---- after the loop: attention tensors (critical path into the listener) ----
        if not bah:
            # dot-product scores: d(keys)[b] = dScore[b]^T Q[b], Q = the queries (relu(Wq h) for CustomAttention)
            qmat = sv['h'][qlayer]
            if self.custom:
                qmat = torch.empty(B, U, Hd, dtype=bf, device=dev)
                hip.cast_bf16(sv['pq'], BU, Hd, qmat, BU, Hd, ldd=Hd, lds=Hd)
            dkeys = torch.empty(B, Tm, Hd, dtype=f32, device=dev)          # (stored: no zero fill, no atomics)
            hip.gemm_tn(ds_all, qmat, dkeys, Tm, Hd, U, lda=Tmp, ldb=Hd, ldc=Hd, batch=B, sa=U * Tmp, sb=U * Hd, sc=Tm * Hd,
                        store=True)
            if self.custom:              # keys = relu(memory_layer(memory))
                hip.check(lib.las_relu_bwd(hip.p(dkeys), hip.p(sv['keys']), dkeys.numel(), st))
        dmem = torch.empty(B, Tm, M, dtype=f32, device=dev)
        hip.gemm_tn(sv['align_bf'], dctx_all, dmem, Tm, M, U, lda=Tmp, ldb=M, ldc=M, batch=B, sa=U * Tmp, sb=U * M, sc=Tm * M,
                    store=True)
        dkeys_bf = torch.empty(B * Tm, Hd, dtype=bf, device=dev)
        hip.cast_bf16(dkeys, B * Tm, Hd, dkeys_bf, B * Tm, Hd, ldd=Hd, lds=Hd)
        hip.gemm_nt(dkeys_bf, self.wmem, dmem, B * Tm, M, Hd, lda=Hd, ldb=Hd, ldc=M, accumulate=True)
        # ---- weight gradients ----
        if self.binf is None:
            hip.gemm_tn(sv['out'], dlogits, grads[self.K_PROJ], P, Vo, BU, lda=P, ldb=Vop, ldc=Vo, split_k=4)
            hip.colsum_bf16(dlogits, BU, Vo, grads[self.B_PROJ], ldx=Vop)
        elif self.binf_var is not None:
            # logits = raw [Mb; 1 - Mb]:  d(Mb) = (raw^T dlogits)[:nf] - (raw^T dlogits)[nf:]
            dwb = torch.zeros(2 * self.nf, Vop, dtype=f32, device=dev)
            hip.gemm_tn(sv['out'], dlogits, dwb, 2 * self.nf, Vop, BU, lda=P, ldb=Vop, ldc=Vop, split_k=4)
            grads[self.binf_var].add_(dwb[:self.nf, :V] - dwb[self.nf:, :V])
        hip.gemm_tn(sv['memory'], dkeys_bf, grads[self.K_MEM], M, Hd, B * Tm, lda=M, ldb=Hd, ldc=Hd, split_k=8)
        if self.uses_wq:
            hip.gemm_tn(sv['h'][qlayer], dpq_all, grads[self.K_Q], Hd, Hd, BU, lda=Hd, ldb=Hd, ldc=Hd, split_k=4)
        if self.has_al:
            hip.gemm_tn(sv['qc'], datt_bf, grads[self.K_AL], Hd + M, A, BU, lda=Hd + M, ldb=A, ldc=A, split_k=4)
        for l in range(NL):
            kn, bn = self.cell_names(l)
            skip = self.E if l == 0 else 0
            if l == 0 and self.tokx:     # operand columns [0,E) are the token rows of the kernel, [T0, ...) the rest
                X0 = sv['X'][0]
                hip.gemm_tn(X0, dz[0], grads[kn], self.E, 4 * Hd, BU, lda=W[0], ldb=4 * Hd, ldc=4 * Hd, split_k=4)
                hip.gemm_tn(X0.view(BU, W[0])[:, self.T0:], dz[0], grads[kn][skip:], W[0] - self.T0, 4 * Hd, BU, lda=W[0],
                            ldb=4 * Hd, ldc=4 * Hd, split_k=4)
            else:
                hip.gemm_tn(sv['X'][l], dz[l], grads[kn][skip:], W[l], 4 * Hd, BU, lda=W[l], ldb=4 * Hd, ldc=4 * Hd, split_k=4)
            hip.colsum_bf16(dz[l], BU, 4 * Hd, grads[bn], ldx=4 * Hd)
        if self.tokx:
            if dtokx is not None:        # d(embedding) = onehot^T d(embedded tokens)
                onehot = torch.empty(BU, Vp, dtype=bf, device=dev)
                fed = sv['fed']
                hip.check(lib.las_onehot_bf16(hip.p(fed), fed.stride(0), B, U, V, hip.p(onehot), Vp, 1.0, 0, 0, 1, st))
                dE = torch.zeros(V, self.Ep, dtype=f32, device=dev)
                hip.gemm_tn(onehot, dtokx, dE, V, self.Ep, BU, lda=Vp, ldb=self.Ep, ldc=self.Ep, split_k=4)
                if self.binf_var is not None:        # the embedding table is Mb^T
                    grads[self.binf_var].add_(dE[:, :self.nf].t())
                else:
                    grads[self.K_EMB].add_(dE[:, :self.E])
            d_state = None
            if sv['passed']:
                d_state = [(dc[l], dx[l][0][:, self.win[l]:]) for l in range(sv['passed'])]
            self.saved = None
            return dmem, d_state
        # token part of cell 0: d(rows) = onehot^T dz_0, then through the embedding if there is one
        onehot = torch.empty(BU, Vp, dtype=bf, device=dev)
        fed = sv['fed']
        hip.check(lib.las_onehot_bf16(hip.p(fed), fed.stride(0), B, U, V, hip.p(onehot), Vp, sv['keep'], sv['seed'],
                                      self.DEC_STREAM, self.E + A, st))
        k0 = grads[self.cell_names(0)[0]]
        if not self.emb:
            hip.gemm_tn(onehot, dz[0], k0, V, 4 * Hd, BU, lda=Vp, ldb=4 * Hd, ldc=4 * Hd, split_k=4)
        else:
            dtok = torch.zeros(V, 4 * Hd, dtype=f32, device=dev)
            hip.gemm_tn(onehot, dz[0], dtok, V, 4 * Hd, BU, lda=Vp, ldb=4 * Hd, ldc=4 * Hd, split_k=4)
            dtok_bf = torch.empty(V, 4 * Hd, dtype=bf, device=dev)
            hip.cast_bf16(dtok, V, 4 * Hd, dtok_bf, V, 4 * Hd)
            hip.gemm_tn(self.emb_bf, dtok_bf, k0, self.E, 4 * Hd, V, lda=self.Ep, ldb=4 * Hd, ldc=4 * Hd)     # emb^T dRows
            if self.binf is None:
                dE = torch.zeros(V, self.Ep, dtype=f32, device=dev) if self.Ep != self.E else grads[self.K_EMB]
                hip.gemm_nt(dtok_bf, self.k0tok, dE, V, self.Ep, 4 * Hd, lda=4 * Hd, ldb=4 * Hd, ldc=self.Ep,
                            accumulate=True)                                                                   # dRows K0^T
                if dE is not grads[self.K_EMB]:
                    grads[self.K_EMB].add_(dE[:, :self.E])
            elif self.binf_var is not None:          # the embedding table is Mb^T: d(Mb) += (dRows K0^T)^T
                dE = torch.zeros(V, self.Ep, dtype=f32, device=dev)
                hip.gemm_nt(dtok_bf, self.k0tok, dE, V, self.Ep, 4 * Hd, lda=4 * Hd, ldb=4 * Hd, ldc=self.Ep, accumulate=True)
                grads[self.binf_var].add_(dE[:, :self.nf].t())
        d_state = None
        if sv['passed']:
            fin = (0) & 1                      # the buffers written by step t = 0
            d_state = [(dc[l], dx[l][fin][:, self.win[l]:]) for l in range(sv['passed'])]
        self.saved = None
        return dmem, d_state

    # ------------------------------------------------------------------------------------------------------------------
    def forward_beam(self, memory, mem_len, encoder_state, max_iterations, beam_width, partial_targets=None):
        """BeamSearchDecoder decode (las/model.py:219-226,298-319,346-347): the batch tiled beam_width times, one
        las_beam_step per decoder step, decoder state gathered by the parent beams, gather_tree at the end.
        partial_targets [B,L] int (features['partial_targets'], model_helper.py:203): the decoder is first run
        teacher-forced over these tokens (get_partial_targets_state, las/model.py:299-307,351-361) and the search
        starts from that state with start_tokens = partial_targets[:, 0], as the reference does.
        Returns (predicted_ids [B,T,K] int32, final lengths [B,K], log-probabilities of the final beams [B,K])."""
        return self.forward_greedy(memory, mem_len, encoder_state, max_iterations, beam_width=beam_width,
                                   partial_targets=partial_targets)

    def forward_greedy(self, memory, mem_len, encoder_state, max_iterations, parts=4, beam_width=0, partial_targets=None):
        """GreedyEmbeddingHelper decode (las/model.py:270-274,337-347) with the general cell stack
        (beam_width > 0: see forward_beam)."""
        d = self.hp
        K = int(beam_width)
        if partial_targets is not None and K <= 0:
            raise ValueError('partial_targets is a beam-search option (las/model.py:298-307)')
        # the sigmoid-output decoder decodes with the InferenceHelper of las/model.py:320-336: the first input is the
        # feature vector of <s> (zeros, 1, 0), a step's sample is round(sigmoid(logits)), fed back as it is, and an utterance
        # ends when its last feature (the one of </s>) is set
        binary = self.sigmoid
        if binary and K > 0:
            raise ValueError('beam search over the sigmoid-output decoder: the reference\'s BeamSearchDecoder needs the phone '
                             'scores of transform_binf_to_phones, which are undefined for binf_count-wide outputs '
                             '(utils/training_helper.py:17-27)')
        if K > 0:                       # tf.contrib.seq2seq.tile_batch of memory, lengths and the encoder state
            B0 = memory.shape[0]
            tile = lambda x: x.repeat_interleave(K, 0).contiguous()
            memory, mem_len = tile(memory), tile(mem_len)
            if isinstance(encoder_state[0], tuple):
                encoder_state = tuple(type(s)(tile(s.c), tile(s.h)) for s in encoder_state)
            else:
                encoder_state = type(encoder_state)(tile(encoder_state.c), tile(encoder_state.h))
        B, Tm, M = memory.shape
        Hd, V, Vp, A, NL = self.Hd, self.Vo, self.Vop, self.A, self.NL       # V / Vp: width of the projection output
        dev, bf, f32 = memory.device, torch.bfloat16, torch.float32
        # steps [0, L) run teacher-forced over the partial targets (no projection, no search), the search follows
        L = 0
        if partial_targets is not None:
            prefix = partial_targets.to(device=dev, dtype=torch.int32).repeat_interleave(K, 0).contiguous()
            L = int(prefix.shape[1])
        S = max_iterations + L
        init, _ = self._init_states(encoder_state, B)
        ids0 = torch.full((B, 1), d.sos_id, dtype=torch.int32, device=dev)
        fed = torch.full((B, max(S, 1)), d.eos_id, dtype=torch.int32, device=dev)
        fed[:, :1] = ids0
        if L > 0:
            fed[:, :L] = prefix
        # reuse the training graph step by step on [B, S] buffers (teacher tokens replaced by the argmax)
        keys = self._keys(memory)
        Tmp = _r8(Tm)
        U = max(S, 1)
        sv = dict(B=B, Tm=Tm, U=U, memory=memory, mem_len=mem_len, keys=keys, keep=1.0, seed=0, norm=self._norm(False))
        sv['X'] = [torch.zeros(B, U, w + Hd, dtype=bf, device=dev) for w in self.win]
        sv['gates'] = [torch.empty(B, U, 4 * Hd, dtype=f32, device=dev) for _ in range(NL)]
        sv['cs'] = [torch.empty(B, U + 1, Hd, dtype=f32, device=dev) for _ in range(NL)]
        sv['h'] = [torch.empty(B, U, Hd, dtype=bf, device=dev) for _ in range(NL)]
        sv['align'] = torch.zeros(B, U, Tmp, dtype=f32, device=dev)
        sv['align_bf'] = torch.zeros(B, U, Tmp, dtype=bf, device=dev)
        sv['ctx'] = torch.empty(B, U, M, dtype=bf, device=dev)
        sv['pq'] = torch.empty(B, U, Hd, dtype=f32, device=dev) if self.uses_wq else None
        att = torch.empty(B, U, A, dtype=bf, device=dev) if self.has_al else sv['ctx']
        qc = torch.empty(B, U, Hd + M, dtype=bf, device=dev) if self.has_al else None
        for l in range(NL):
            sv['cs'][l][:, 0].copy_(init[l][0])
            sv['X'][l][:, 0, self.win[l]:].copy_(init[l][1])
        z = torch.empty(B, 4 * Hd, dtype=f32, device=dev)
        logits = torch.zeros(B, U, Vp, dtype=f32, device=dev)
        samples = torch.full((B, U), d.eos_id, dtype=torch.int32, device=dev)
        if binary:
            samples = torch.zeros(B, U, self.nf, dtype=f32, device=dev)
            sv['X'][0][:, 0, self.nf - 2] = 1.0                       # start_inputs = [0, ..., 0, 1, 0]
        finished = torch.zeros(B, dtype=torch.bool, device=dev)
        final_len = torch.zeros(B, dtype=torch.int32, device=dev)
        X, h = sv['X'], sv['h']
        qlayer = 0 if self.bottom else NL - 1
        steps = 0
        if K > 0:
            i32 = torch.int32
            b_lp = torch.full((B0, K), float('-inf'), dtype=f32, device=dev)
            b_lp[:, 0] = 0.0
            b_fin = torch.zeros(B0, K, dtype=i32, device=dev)
            b_len = torch.zeros(B0, K, dtype=i32, device=dev)
            b_word = torch.zeros(B0, K, dtype=i32, device=dev)
            b_par = torch.zeros(B0, K, dtype=i32, device=dev)
            parents = torch.zeros(B, U, dtype=i32, device=dev)
            base = (torch.arange(B0, device=dev) * K).unsqueeze(1)
        for t in range(S):
            last = t + 1 == S

            if self.tokx and not binary:
                X[0][:, t, :self.Ep] = self.emb_bf[fed[:, t].long()]

            def run_cell(l):
                Kl = self.win[l] + Hd
                hip.gemm_nt(X[l][:, t], self.kT[l], z, B, 4 * Hd, Kl, lda=U * Kl, ldb=Kl, ldc=4 * Hd)
                self._cell_fwd(l, t, sv, z, hip.addr(fed, t), fed.stride(0))
                if not last:
                    X[l][:, t + 1, self.win[l]:].copy_(h[l][:, t])

            run_cell(0)
            if not self.bottom:
                for l in range(1, NL):
                    X[l][:, t, :Hd].copy_(h[l - 1][:, t])
                    run_cell(l)
            self._attention_fwd(t, sv, hip.addr(h[qlayer], t * Hd), U * Hd)
            if self.has_al:
                qc[:, t, :Hd].copy_(h[qlayer][:, t])
                qc[:, t, Hd:].copy_(sv['ctx'][:, t])
                hip.gemm_nt(qc[:, t], self.walT, att[:, t], B, A, Hd + M, lda=U * (Hd + M), ldb=Hd + M, ldc=U * A, out_bf16=True)
            if self.bottom:
                for l in range(1, NL):
                    wc = A if l == 1 else Hd
                    X[l][:, t, :wc].copy_(att[:, t] if l == 1 else h[l - 1][:, t])
                    run_cell(l)
            if t < L:                   # prefix step: only the state moves on
                steps = t + 1
                if not last:
                    X[0][:, t + 1, self.T0:self.T0 + A].copy_(att[:, t])
                    if self.bottom:
                        for l in range(1, NL):
                            wc = A if l == 1 else Hd
                            X[l][:, t + 1, wc:wc + A].copy_(att[:, t])
                    if t + 1 == L:
                        fed[:, L] = prefix[:, 0]
                continue
            out_t = h[NL - 1][:, t] if (self.bottom and NL > 1) else att[:, t]
            hip.gemm_nt(out_t, self.wprojT, logits[:, t], B, Vp, self.P, lda=out_t.stride(0), ldb=self.P, ldc=U * Vp,
                        bias=self.bproj)
            if K > 0:
                hip.check(hip.lib().las_beam_step(hip.addr(logits, t * Vp), U * Vp, hip.p(b_lp), hip.p(b_fin), hip.p(b_len),
                                                  hip.p(b_word), hip.p(b_par), B0, K, V, d.eos_id, hip.stream()))
                sample = b_word.view(-1)
                samples[:, t] = sample
                parents[:, t] = b_par.view(-1)
                finished = b_fin.view(-1) != 0
            elif binary:
                sample = (logits[:, t, :V] > 0).to(f32)                # round(sigmoid(x)): 1 iff x > 0
                samples[:, t] = sample
                final_len = torch.where(finished, final_len, torch.full_like(final_len, t + 1))
                finished = finished | (sample[:, V - 1] > 0.5)          # end_fn: the </s> feature
            else:
                sample = logits[:, t, :V].argmax(-1).to(torch.int32)
                samples[:, t] = sample
                final_len = torch.where(finished, final_len, torch.full_like(final_len, t + 1))
                finished = finished | (sample == d.eos_id)
            steps = t + 1
            if not last and binary:
                X[0][:, t + 1, :V] = sample.to(bf)
            if not last:
                if not binary:
                    fed[:, t + 1] = sample
                X[0][:, t + 1, self.T0:self.T0 + A].copy_(att[:, t])
                if self.bottom:
                    for l in range(1, NL):
                        wc = A if l == 1 else Hd
                        X[l][:, t + 1, wc:wc + A].copy_(att[:, t])
                if K > 0:               # the surviving beams continue from their parents' decoder state
                    flat = (b_par.long() + base).view(-1)
                    for l in range(NL):
                        X[l][:, t + 1] = X[l][:, t + 1][flat]
                        sv['cs'][l][:, t + 1] = sv['cs'][l][:, t + 1][flat]
                    if self.mono:
                        sv['align'][:, t] = sv['align'][:, t][flat]
            if bool(finished.all()):
                break
        if K > 0:
            n = steps - L
            ids = gather_tree(samples[:, L:steps].reshape(B0, K, n).permute(2, 0, 1).cpu().numpy(),
                              parents[:, L:steps].reshape(B0, K, n).permute(2, 0, 1).cpu().numpy(),
                              b_len.max(1).values.cpu().numpy(), d.eos_id)
            return torch.from_numpy(ids).permute(1, 0, 2).contiguous().to(dev), b_len, b_lp
        # raw cell outputs of the decode: what BasicTransparentProjectionDecoder returns as rnn_output (training_helper.py:156-178)
        self.last_raw_outputs = (h[NL - 1] if (self.bottom and NL > 1) else att)[:, :steps]
        return logits[:, :steps, :V], samples[:, :steps], final_len, sv['align'][:, :steps, :Tm]
