"""CPU oracle for the LAS training hot path (TEST INFRASTRUCTURE ONLY).

This file is a from-scratch restatement, in torch-CPU float64 (autograd gives the backward
pass), of what sciforce/phones-las computes on the path features -> listener -> speller ->
loss -> clip/Adam.  It is NOT product code: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product (``phones-las_amd``) never does.

PARITY UNPINNED: the reference holds no test, golden vector or fixture for this path and its
arithmetic lives in un-vendored ``tensorflow==1.15.2`` (requirements.txt:12), which cannot be
installed or run here.  The restatement is therefore cross-validated against
independent installed implementations (``torch.nn.LSTM``, ``F.cross_entropy``, ``F.ctc_loss``,
finite differences; see tests/test_oracle_*.py) and hand-derivable known answers, not against
outputs of the reference itself.

Each function cites the reference file:line it follows; third-party (TF 1.15) semantics are the
ones listed in SURVEY.md Appendix A.

``mxu='bf16'`` emulates the storage points at which the MI355X path keeps GEMM operands in
bfloat16 (weights, layer outputs h_t, attention context, keys); accumulation and every
element-wise op stay in float64 here / float32 on the device.  ``mxu='f64'`` is the exact model.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

DT = torch.float64


def set_dtype(dtype):
    """Switch the arithmetic type of the restatement (float64 for parity checks; float32 to time the
    reference-shaped fp32 CPU path in bench.py's cpu_baseline leg)."""
    global DT
    DT = dtype

UNK_ID, SOS_ID, EOS_ID = 0, 1, 2          # utils/vocab_utils.py:19-21
GRAD_NORM = 2.0                            # model_helper.py:16


# --------------------------------------------------------------------------------------
# quantisation model of the device path
# --------------------------------------------------------------------------------------
def q_bf16(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bfloat16 with a straight-through gradient."""
    r = x.detach().to(torch.float32).to(torch.bfloat16).to(x.dtype)
    return x + (r - x.detach())


class _RoundGrad(torch.autograd.Function):
    """Identity whose BACKWARD rounds the incoming gradient to bfloat16: the device stores dz (every LSTM cell),
    dlogits, d(keys) and the CTC dlogits in bf16 before they enter its backward GEMMs."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.float32).to(torch.bfloat16).to(g.dtype)


def _r(x):
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


class _ContextFn(torch.autograd.Function):
    """context = align . values.  Device backward (decoder.hip dec_step_bwd_kernel): d(align) from the fp32 d(context);
    d(values) = bf16(align)^T bf16(d context) (one batched GEMM after the loop)."""

    @staticmethod
    def forward(ctx, align, values):
        ctx.save_for_backward(align, values)
        return torch.einsum('bt,btm->bm', align, values)

    @staticmethod
    def backward(ctx, g):
        align, values = ctx.saved_tensors
        return torch.einsum('bm,btm->bt', g, values), torch.einsum('bt,bm->btm', _r(align), _r(g))


class _DotScoreFn(torch.autograd.Function):
    """Luong score = query . keys.  Device backward: d(query) from the fp32 d(score); d(keys) = bf16(d score)^T query
    (one batched GEMM after the loop; the query is the cell output, already bf16)."""

    @staticmethod
    def forward(ctx, query, keys):
        ctx.save_for_backward(query, keys)
        return torch.einsum('bh,bth->bt', query, keys)

    @staticmethod
    def backward(ctx, g):
        query, keys = ctx.saved_tensors
        return torch.einsum('bt,bth->bh', g, keys), torch.einsum('bt,bh->bth', _r(g), query)


class _QueryLayerFn(torch.autograd.Function):
    """processed query = query @ Wq.  Device backward: d(query) from the fp32 d(pq); d(Wq) = query^T bf16(d pq)."""

    @staticmethod
    def forward(ctx, query, wq):
        ctx.save_for_backward(query, wq)
        return query @ wq

    @staticmethod
    def backward(ctx, g):
        query, wq = ctx.saved_tensors
        return g @ wq.t(), query.t() @ _r(g)


def _identity(x):
    return x


# q_bf16 carries the device's BACKWARD storage model as attributes: .g rounds a gradient on its way back,
# .bwd selects the device forms of the three attention products above
q_bf16.g = _RoundGrad.apply
q_bf16.bwd = True


def make_q(mxu: str):
    if mxu == 'f64':
        return _identity
    if mxu == 'bf16':
        return q_bf16
    raise ValueError('mxu must be f64 or bf16')


def _g(q):
    return getattr(q, 'g', _identity)


def _bwd(q):
    return getattr(q, 'bwd', False)


# --------------------------------------------------------------------------------------
# hyper-parameters (mirrors utils/params_utils.py:33-77,119-172 for the keys the path reads)
# --------------------------------------------------------------------------------------
@dataclass
class EncoderHP:
    num_layers: int = 3
    num_units: int = 64
    use_pyramidal: bool = True
    unidirectional: bool = False
    dropout: float = 0.0


@dataclass
class DecoderHP:
    num_layers: int = 2
    num_units: int = 128
    target_vocab_size: int = 0
    embedding_size: int = 0
    sampling_probability: float = 0.0
    sos_id: int = SOS_ID
    eos_id: int = EOS_ID
    bottom_only: bool = False
    pass_hidden_state: bool = False
    decoding_length_factor: float = 1.0
    attention_type: str = 'luong'
    attention_layer_size: Optional[int] = None
    dropout: float = 0.0
    binary_outputs: bool = False             # with binf_projection False: the sigmoid-output decoder (feature logits)
    multitask: bool = False                  # --multitask: phone decoder ('speller') + binary decoder ('speller_binf')
    binf_projection: bool = False
    binf_count: int = 0
    binf_projection_reg_weight: float = 1.0
    binf_map: Optional[object] = None        # [binf_count, V] 0/1 matrix (utils.load_binf2phone); a constant ...
    binf_trainable: bool = False             # ... unless --binf_trainable: variable 'binf2phone', U(0,1) init (model_helper.py:182-184)
    max_symbols: int = -1


@dataclass
class HP:
    encoder: EncoderHP = field(default_factory=EncoderHP)
    decoder: DecoderHP = field(default_factory=DecoderHP)
    num_channels: int = 39
    learning_rate: float = 1e-3
    l2_reg_scale: float = 1e-6
    ctc_weight: float = -1.0


# --------------------------------------------------------------------------------------
# parameter table (TF variable shapes; kernel = [D+H, 4H], gate order i,j,f,o)
# --------------------------------------------------------------------------------------
def encoder_out_depth(e: EncoderHP) -> int:
    dirs = 1 if e.unidirectional else 2
    if e.use_pyramidal:
        return dirs * e.num_units * (1 if e.num_layers == 1 else 2)   # las/ops.py:49-87
    return dirs * e.num_units


def attention_depth(hp: HP) -> int:
    d = hp.decoder
    if d.binf_projection:
        return 2 * d.binf_count                                       # las/model.py:180-181
    return d.attention_layer_size or encoder_out_depth(hp.encoder)


def speller_plan(d: DecoderHP):
    """[(scope, kind)] of the decoders las_model_fn builds (model_helper.py:211-227); kind: 'phones', 'binf_projection'
    or 'sigmoid' (binary_outputs without binf_projection).  A single decoder is scoped 'speller' here whatever its kind
    (the reference says 'speller_binf' for a binary one), the second decoder of --multitask 'speller_binf'."""
    binary = d.binary_outputs or d.binf_projection
    kind = 'binf_projection' if d.binf_projection else 'sigmoid'
    if not binary:
        return [('speller', 'phones')]
    if d.multitask:
        return [('speller', 'phones'), ('speller_binf', kind)]
    return [('speller', kind)]


def _speller_table(hp: HP, scope: str, kind: str):
    e, d = hp.encoder, hp.decoder
    M, Hd, V = encoder_out_depth(e), d.num_units, d.target_vocab_size
    A = 2 * d.binf_count if kind == 'binf_projection' else (d.attention_layer_size or M)
    E = d.embedding_size if d.embedding_size else V
    if kind != 'phones' and not d.embedding_size:
        E = d.binf_count                                             # las/model.py:237-243
    Vo = d.binf_count if kind == 'sigmoid' else V                    # DenseBinfDecoder(binf_count) (las/model.py:251-252)
    out = []
    if d.embedding_size:
        out.append((f'{scope}/target_embedding', (V, d.embedding_size), 'glorot'))
    out.append((f'{scope}/memory_layer/kernel', (M, Hd), 'glorot'))
    if d.attention_type in ('bahdanau', 'bahdanau_monotonic', 'custom'):
        out.append((f'{scope}/query_layer/kernel', (Hd, Hd), 'glorot'))
    if d.attention_type in ('bahdanau', 'bahdanau_monotonic'):
        out.append((f'{scope}/attention_v', (Hd,), 'glorot_v'))
    if d.attention_type in ('luong_monotonic', 'bahdanau_monotonic'):
        out.append((f'{scope}/attention_score_bias', (1,), 'zeros'))
    if d.attention_layer_size or kind == 'binf_projection':
        out.append((f'{scope}/attention_layer/kernel', (Hd + M, A), 'glorot'))
    for l in range(d.num_layers):
        if d.bottom_only:                                            # las/model.py:36-69: cell_1 reads [attention_t, attention_{t-1}]
            din = (E + A) if l == 0 else ((A + A) if l == 1 else (Hd + A))
        else:
            din = (E + A) if l == 0 else Hd                         # MultiRNNCell inside the wrapper
        out.append((f'{scope}/decoder_cell_{l}/lstm_cell/kernel', (din + Hd, 4 * Hd), 'lstm'))
        out.append((f'{scope}/decoder_cell_{l}/lstm_cell/bias', (4 * Hd,), 'zeros'))
    P = Hd if (d.bottom_only and d.num_layers > 1) else A
    out.append((f'{scope}/projection_layer/kernel', (P, Vo), 'proj'))
    out.append((f'{scope}/projection_layer/bias', (Vo,), 'zeros'))
    return out


def param_table(hp: HP) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (name, shape, init) list.  init in {lstm, zeros, glorot, proj, emb}."""
    e, d = hp.encoder, hp.decoder
    H = e.num_units
    dirs = ['fw'] if e.unidirectional else ['fw', 'bw']
    out: List[Tuple[str, Tuple[int, ...], str]] = []
    if (d.binary_outputs or d.binf_projection) and d.binf_trainable:             # model_helper.py:181-184
        out.append(('binf2phone', (d.binf_count, d.target_vocab_size), 'uniform01'))
    D = hp.num_channels
    for l in range(e.num_layers):
        for dr in dirs:
            if e.use_pyramidal:
                base = f'listener/bilstm_{l}/{dr}/lstm_cell'
            else:
                base = f'listener/{dr}/multi_rnn_cell/cell_{l}/lstm_cell'
            out.append((base + '/kernel', (D + H, 4 * H), 'lstm'))
            out.append((base + '/bias', (4 * H,), 'zeros'))
        if e.use_pyramidal:
            D = len(dirs) * H * (1 if l == 0 else 2)
        else:
            D = H
    M = encoder_out_depth(e)
    V = d.target_vocab_size
    for scope, kind in speller_plan(d):
        out.extend(_speller_table(hp, scope, kind))
    if hp.ctc_weight > 0:
        out.append(('ctc_logits/kernel', (M, V + 1), 'glorot'))
        out.append(('ctc_logits/bias', (V + 1,), 'zeros'))
    return out


def init_params(hp: HP, seed: int = 4321, bias_scale: float = 0.0) -> Dict[str, torch.Tensor]:
    """U(-0.075,0.075) for LSTM/projection kernels (las/ops.py:12, las/model.py:257), glorot-uniform
    for Dense layers, zeros for biases (``bias_scale`` > 0 perturbs them so tests exercise them)."""
    rng = np.random.default_rng(seed)
    p: Dict[str, torch.Tensor] = {}
    for name, shape, init in param_table(hp):
        if init in ('lstm', 'proj'):
            a = rng.uniform(-0.075, 0.075, size=shape)
        elif init == 'glorot':
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            a = rng.uniform(-lim, lim, size=shape)
        elif init == 'glorot_v':
            lim = math.sqrt(6.0 / (shape[0] + 1))
            a = rng.uniform(-lim, lim, size=shape)
        elif init == 'uniform01':
            a = rng.uniform(0.0, 1.0, size=shape)
        else:
            a = rng.uniform(-bias_scale, bias_scale, size=shape) if bias_scale > 0 else np.zeros(shape)
        p[name] = torch.tensor(a.astype(np.float32).astype(np.float64), dtype=DT)
    return p


# --------------------------------------------------------------------------------------
# listener  (las/ops.py, las/model.py:104-142)
# --------------------------------------------------------------------------------------
def lstm_step(x, c, h, kernel, bias, gq=_identity):
    """tf.nn.rnn_cell.LSTMCell as built by las/ops.py:10-12 (Appendix A.1): gate order i,j,f,o,
    forget_bias 1.0, no peepholes/projection.  gq: the device keeps d(z) in bf16 (storage model of the backward)."""
    z = gq(torch.cat([x, h], 1) @ kernel + bias)
    i, j, f, o = z.chunk(4, dim=1)
    c2 = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return c2, h2


def dynamic_rnn(x, length, kernel, bias, q, reverse=False, in_mask=None):
    """tf.nn.dynamic_rnn with sequence_length (las/ops.py:42-46; Appendix A.3): zero initial state;
    for t >= len: output 0, state carried.  ``reverse`` = the bw half of bidirectional_dynamic_rnn:
    reverse_sequence(x, len) -> rnn -> reverse_sequence back (las/ops.py:35-40).
    ``in_mask`` [B,T,D] optional DropoutWrapper(input_keep_prob) mask already scaled (A.2)."""
    B, T, _ = x.shape
    H = kernel.shape[1] // 4
    kq = q(kernel)
    c = torch.zeros(B, H, dtype=DT)
    h = torch.zeros(B, H, dtype=DT)
    ar = torch.arange(B)
    vals, poss = [], []
    for s in range(T):
        active = (s < length)
        if reverse:
            pos = torch.clamp(length - 1 - s, min=0)
        else:
            pos = torch.full((B,), s, dtype=torch.long)
        xt = x[ar, pos]
        if in_mask is not None:
            xt = q(xt * in_mask[ar, pos])             # the device stores the dropped input in bf16
        c2, h2 = lstm_step(xt, c, h, kq, bias, _g(q))
        h2 = q(h2)                                  # device stores h_t in bf16
        m = active.unsqueeze(1).to(DT)
        c = m * c2 + (1 - m) * c
        h = m * h2 + (1 - m) * h
        vals.append(m * h2)
        poss.append(pos)
    if reverse:
        # inactive rows were clamped to position 0 with a zero value: accumulate is exact
        out = torch.zeros(B, T, H, dtype=DT).index_put(
            (ar.repeat(T), torch.cat(poss)), torch.cat(vals, 0), accumulate=True)
    else:
        out = torch.stack(vals, 1)
    return out, (c, h)


def bilstm(x, length, params, prefix, q, unidirectional=False, in_masks=None):
    """las/ops.py:23-46.  in_masks: optional (mask_fw, mask_bw) realised DropoutWrapper masks [B,T,D] (already
    divided by keep_prob; the fw and bw cells have independent wrappers, las/ops.py:30-34)."""
    mf, mb = in_masks if in_masks is not None else (None, None)
    fw, sfw = dynamic_rnn(x, length, params[prefix + '/fw/lstm_cell/kernel'],
                          params[prefix + '/fw/lstm_cell/bias'], q, in_mask=mf)
    if unidirectional:
        return fw, sfw
    bw, sbw = dynamic_rnn(x, length, params[prefix + '/bw/lstm_cell/kernel'],
                          params[prefix + '/bw/lstm_cell/bias'], q, reverse=True, in_mask=mb)
    return (fw, bw), (sfw, sbw)


def pyramidal_stack(outputs, length):
    """las/ops.py:49-65: pad T to even with zeros, [B,T,C] -> [B,T/2,2C], len -> len//2 + len%2."""
    B, T, C = outputs.shape
    if T % 2:
        outputs = torch.cat([outputs, torch.zeros(B, 1, C, dtype=DT)], 1)
    return outputs.reshape(B, -1, 2 * C), length // 2 + length % 2


def pyramidal_bilstm(x, length, params, e: EncoderHP, q, in_masks=None):
    """las/ops.py:68-87.  Returns ((outputs, lengths), state_of_last_layer).  in_masks: per-layer list of
    (mask_fw, mask_bw) or None."""
    out = x
    state = None
    for l in range(e.num_layers):
        o, state = bilstm(out, length, params, f'listener/bilstm_{l}', q, e.unidirectional,
                          in_masks[l] if in_masks is not None else None)
        out = o if e.unidirectional else torch.cat(o, -1)          # las/ops.py:81
        if l != 0:
            out, length = pyramidal_stack(out, length)             # las/ops.py:83-85
    return (out, length), state


def listener(x, length, params, e: EncoderHP, mxu='f64', in_masks=None):
    """las/model.py:104-142."""
    q = make_q(mxu)
    x = q(x.to(DT))
    if e.use_pyramidal:
        return pyramidal_bilstm(x, length, params, e, q, in_masks)
    dirs = ['fw'] if e.unidirectional else ['fw', 'bw']
    outs, states = [], []
    for dr in dirs:                                                 # MultiRNNCell per direction
        o = x
        st = []
        for l in range(e.num_layers):
            base = f'listener/{dr}/multi_rnn_cell/cell_{l}/lstm_cell'
            o, s = dynamic_rnn(o, length, params[base + '/kernel'], params[base + '/bias'], q,
                               reverse=(dr == 'bw'))
            st.append(s)
        outs.append(o)
        states.append(tuple(st))
    out = outs[0] if e.unidirectional else torch.cat(outs, -1)
    state = states[0] if e.unidirectional else tuple(states)
    return (out, length), state


# --------------------------------------------------------------------------------------
# attention  (las/model.py:145-202; Appendix A.5/A.6)
# --------------------------------------------------------------------------------------
def _safe_cumprod_excl(x):
    # exclusive cumprod along last dim, computed in log space like tf.contrib.seq2seq.safe_cumprod
    tiny = np.finfo(np.float32).tiny
    lx = torch.log(torch.clamp(x, tiny, 1.0))
    cs = torch.cumsum(lx, -1) - lx
    return torch.exp(cs)


def monotonic_attention(p, prev, mode):
    """tf.contrib.seq2seq.monotonic_attention (Appendix A.6)."""
    if mode == 'parallel':
        cp = _safe_cumprod_excl(1 - p)
        return p * cp * torch.cumsum(prev / torch.clamp(cp, 1e-10, 1.0), -1)
    if mode == 'hard':
        p = p * torch.cumsum(prev, -1)
        ones = torch.ones_like(p[:, :1])
        return p * torch.cumprod(torch.cat([ones, 1 - p], -1), -1)[:, :-1]
    raise ValueError(mode)


class Attention:
    """One of the mechanisms selected in las/model.py:153-169 over ``memory`` [B,T',M]."""

    def __init__(self, hp: HP, params, memory, mem_len, q, train=True, noise=None, scope='speller'):
        self.kind = hp.decoder.attention_type
        self.q = q
        # this decoder's variables under the plain 'speller/...' names the methods below use
        self.p = params if scope == 'speller' else {'speller/' + k[len(scope) + 1:]: v for k, v in params.items()
                                                    if k.startswith(scope + '/')}
        params = self.p
        B, Tm, _ = memory.shape
        self.mask = (torch.arange(Tm).unsqueeze(0) < mem_len.unsqueeze(1))       # [B,T']
        self.values = memory * self.mask.unsqueeze(-1).to(DT)
        self.keys = _g(q)(q(self.values @ q(params['speller/memory_layer/kernel'])))   # d(keys): summed over the steps in fp32, one bf16 rounding
        if self.kind == 'custom':
            self.keys = torch.relu(self.keys)                                    # las/model.py:97
        self.train = train
        self.noise = noise        # optional [U,B,T'] N(0,1) draws for bahdanau_monotonic TRAIN
        self.step_i = 0

    def initial_alignments(self, B):
        Tm = self.values.shape[1]
        a = torch.zeros(B, Tm, dtype=DT)
        if 'monotonic' in self.kind:
            a[:, 0] = 1.0
        return a

    def __call__(self, query, prev_align):
        kind, p, q = self.kind, self.p, self.q
        if kind in ('luong', 'luong_monotonic', 'custom'):
            qq = query
            if kind == 'custom':
                wq = q(p['speller/query_layer/kernel'])
                qq = q(torch.relu(_QueryLayerFn.apply(query, wq) if _bwd(q) else query @ wq))   # las/model.py:99
            score = _DotScoreFn.apply(qq, self.keys) if _bwd(q) else torch.einsum('bh,bth->bt', qq, self.keys)
        else:
            wq = q(p['speller/query_layer/kernel'])
            pq = _QueryLayerFn.apply(query, wq) if _bwd(q) else query @ wq
            score = torch.einsum('h,bth->bt', p['speller/attention_v'],
                                 torch.tanh(self.keys + pq.unsqueeze(1)))
        if 'monotonic' in kind:
            score = score + p['speller/attention_score_bias']
            mode = 'parallel'
            if kind == 'bahdanau_monotonic':                                     # las/model.py:159-164
                if self.train:
                    if self.noise is not None:
                        score = score + self.noise[self.step_i]
                else:
                    mode = 'hard'
            self.step_i += 1
            if mode == 'hard':
                pr = (score > 0).to(DT)
            else:
                pr = torch.sigmoid(score)
            pr = pr * self.mask.to(DT)
            return monotonic_attention(pr, prev_align, mode)
        score = score.masked_fill(~self.mask, float('-inf'))
        return torch.softmax(score, -1)


class Speller:
    """Decoder cell of las/model.py:145-202 (AttentionWrapper / AttentionMultiCell) plus the
    projection layer (las/model.py:251-257) and embedding_fn (las/model.py:228-246)."""

    def __init__(self, hp: HP, params, memory, mem_len, enc_state, mxu='f64', train=True,
                 noise=None, scope='speller', kind=None):
        """kind: 'phones' | 'binf_projection' | 'sigmoid' (default: the first decoder of speller_plan); scope: prefix of
        this decoder's variables ('speller_binf' for the binary decoder of a --multitask model)."""
        self.hp, self.d = hp, hp.decoder
        self.kind = kind or speller_plan(hp.decoder)[0][1]
        self.q = make_q(mxu)
        self.att = Attention(hp, params, memory, mem_len, self.q, train, noise, scope)
        self.p_all = params                      # (variables outside the decoder's scope: the trainable binf2phone)
        self.p = self.att.p
        params = self.p
        self.B = memory.shape[0]
        self.A = (2 * hp.decoder.binf_count) if self.kind == 'binf_projection' else \
            (hp.decoder.attention_layer_size or encoder_out_depth(hp.encoder))
        self.train = train
        d = self.d
        Hd = d.num_units
        z = lambda n: torch.zeros(self.B, n, dtype=DT)
        self.cells = [(z(Hd), z(Hd)) for _ in range(d.num_layers)]
        if d.pass_hidden_state and d.bottom_only:                                # las/model.py:259-268
            es = list(enc_state) if isinstance(enc_state[0], tuple) else [enc_state]
            n = min(len(self.cells), len(es))
            for l in range(n):
                c, h = es[l]
                if c.shape[1] != Hd:
                    raise ValueError('pass_hidden_state needs decoder_units == encoder_units')
                self.cells[l] = (c, h)
        self.attention = z(self.A)
        self.align = self.att.initial_alignments(self.B)
        self.align_hist: List[torch.Tensor] = []

    def embed(self, ids):
        d = self.d
        if d.embedding_size:
            return self.q(self.p['speller/target_embedding'])[ids]
        if self.kind in ('binf_projection', 'sigmoid'):                          # las/model.py:237-243
            return self.binf_map().t()[ids]
        return torch.nn.functional.one_hot(ids, d.target_vocab_size).to(DT)

    def binf_map(self):
        """binf_embedding of model_helper.py:181-186: the constant map, or the trainable variable (bf16 as a GEMM operand
        and as the token feed on the device: rounded here too in the bf16 model)."""
        if self.d.binf_trainable:
            return self.q(self.p_all['binf2phone'])
        return torch.as_tensor(self.d.binf_map, dtype=DT)

    def project(self, out):
        """projection_layer of las/model.py:251-257 (DenseBinfDecoder, utils/training_helper.py:122-153); for the
        sigmoid-output decoder a Dense(binf_count) whose outputs are feature logits (binf_to_ipa None in TRAIN)."""
        d, p, q = self.d, self.p, self.q
        if self.kind != 'binf_projection':
            return out @ q(p['speller/projection_layer/kernel']) + p['speller/projection_layer/bias']
        # inner_projection_layer=False: the cell output IS [log p(feature=1) | log p(feature=0)]; the Dense kernel and
        # bias exist as variables but are not applied.  transform_binf_to_phones (:17-27); TRAIN concatenates the input.
        Mb = self.binf_map()
        nf = Mb.shape[0]
        # The output is the 2*binf_count attention vector -- or, for a multi-layer --bottom_only decoder (AttentionMultiCell,
        # las/model.py:36-69,188-200), the TOP CELL's h: transform_binf_to_phones then slices ITS first 2*binf_count columns
        # (utils/training_helper.py:19-21; narrower outputs fail in its matmul when the graph is built).
        if out.shape[1] < 2 * nf:
            raise ValueError('binf_projection: the decoder output (%d wide) must hold [lp1 | lp0] = 2*binf_count = %d columns'
                             % (out.shape[1], 2 * nf))
        logits = out[:, :nf] @ Mb + out[:, nf:2 * nf] @ (1 - Mb)
        return torch.cat([logits, out], 1) if self.train else logits

    def _cell(self, l, x, state):
        k = self.q(self.p[f'speller/decoder_cell_{l}/lstm_cell/kernel'])
        b = self.p[f'speller/decoder_cell_{l}/lstm_cell/bias']
        c2, h2 = lstm_step(x, state[0], state[1], k, b, _g(self.q))
        return c2, self.q(h2)

    def step(self, inputs, in_mask=None):
        """One AttentionWrapper step (Appendix A.5); returns logits [B,V].  in_mask [B, E+A]: realised input
        dropout mask of the (bottom) cell, already divided by keep_prob (A.2)."""
        d, p, q = self.d, self.p, self.q
        old_att = self.attention
        x = torch.cat([inputs, old_att], 1)
        lmask = (lambda l: None)
        if in_mask is not None:
            if isinstance(in_mask, (list, tuple)):           # one mask per decoder cell (each has its own wrapper)
                lmask = (lambda l: in_mask[l])
                x = q(x * in_mask[0])
            else:
                x = q(x * in_mask)
        new_cells = []
        if d.bottom_only:
            c, h = self._cell(0, x, self.cells[0])
            new_cells.append((c, h))
            cell_out = h
        else:
            cur = x
            for l in range(d.num_layers):
                if l > 0 and lmask(l) is not None:
                    cur = q(cur * lmask(l))
                c, h = self._cell(l, cur, self.cells[l])
                new_cells.append((c, h))
                cur = h
            cell_out = cur
        align = self.att(cell_out, self.align)
        ctx = _ContextFn.apply(align, self.att.values) if _bwd(q) else torch.einsum('bt,btm->bm', align, self.att.values)
        if d.attention_layer_size or self.kind == 'binf_projection':
            ctx = q(ctx)
            attention = torch.cat([cell_out, ctx], 1) @ q(p['speller/attention_layer/kernel'])
        else:
            attention = ctx
        attention = q(attention)
        out = attention
        if d.bottom_only:
            cur = out
            for l in range(1, d.num_layers):                                     # las/model.py:54-67
                cur_in = torch.cat([cur, old_att], 1)
                if lmask(l) is not None:
                    cur_in = q(cur_in * lmask(l))
                c, h = self._cell(l, cur_in, self.cells[l])
                new_cells.append((c, h))
                cur = h
            out = cur
        self.cells = new_cells
        self.attention = attention
        self.align = align
        self.align_hist.append(align)
        return self.project(out)


def speller_train(hp: HP, params, memory, mem_len, enc_state, targets_inputs, target_len,
                  mxu='f64', sample_select=None, sample_ids=None, noise=None, in_masks=None, scope='speller', kind=None,
                  sample_vecs=None):
    """las/model.py:276-296,346-347 with TrainingHelper / TrainingSigmoidHelper; optional scheduled sampling with
    externally supplied draws (utils/training_helper.py:48-87): sample_select[t,b] bool, sample_ids[t,b]; for the
    sigmoid-output decoder (ScheduledSigmoidHelper, :89-119) sample_vecs[t,b,nf] holds the Bernoulli feature draws."""
    sp = Speller(hp, params, memory, mem_len, enc_state, mxu, True, noise, scope, kind)
    U = int(target_len.max())
    inp = sp.embed(targets_inputs[:, 0])
    outs = []
    for t in range(U):
        logits = sp.step(inp, in_masks[t] if in_masks is not None else None)
        outs.append(logits)
        if t + 1 < targets_inputs.shape[1]:
            nxt = sp.embed(targets_inputs[:, t + 1])
        else:
            nxt = torch.zeros_like(inp)
        if sample_select is not None:
            sel = sample_select[t].unsqueeze(1).to(DT)
            samp = sample_vecs[t].to(DT) if sample_vecs is not None else sp.embed(sample_ids[t])
            nxt = sel * samp + (1 - sel) * nxt
        inp = nxt
    return torch.stack(outs, 1), sp


def speller_greedy_binary(hp: HP, params, memory, mem_len, enc_state, mxu='f64', scope='speller'):
    """las/model.py:320-336: the sigmoid-output decoder under InferenceHelper -- start_inputs = [0, ..., 0, 1, 0] (the
    features of <s>), sample = round(sigmoid(outputs)) fed back as the next input, finished when the last feature (the
    one of </s>) exceeds 0.5.  Returns (feature logits [B,steps,nf], samples [B,steps,nf], final lengths, speller)."""
    d = hp.decoder
    sp = Speller(hp, params, memory, mem_len, enc_state, mxu, False, None, scope, 'sigmoid')
    B, nf = memory.shape[0], d.binf_count
    max_it = int(round(float(mem_len.max()) * d.decoding_length_factor))
    inp = torch.zeros(B, nf, dtype=DT)
    inp[:, nf - 2] = 1.0
    finished = torch.zeros(B, dtype=torch.bool)
    final_len = torch.zeros(B, dtype=torch.long)
    outs, samples = [], []
    for t in range(max_it):
        logits = sp.step(inp)
        sample = (logits > 0).to(DT)                           # tf.round(tf.sigmoid(x)): 1 iff x > 0
        outs.append(logits)
        samples.append(sample)
        final_len = torch.where(finished, final_len, torch.full_like(final_len, t + 1))
        finished = finished | (sample[:, -1] > 0.5)
        inp = sample
        if bool(finished.all()):
            break
    return torch.stack(outs, 1), torch.stack(samples, 1), final_len, sp


def speller_greedy(hp: HP, params, memory, mem_len, enc_state, mxu='f64', scope='speller', kind=None):
    """las/model.py:270-274,337-347: GreedyEmbeddingHelper, maximum_iterations =
    round(max(len') * decoding_length_factor); returns logits [B,steps,V], ids, final lengths, speller."""
    d = hp.decoder
    sp = Speller(hp, params, memory, mem_len, enc_state, mxu, False, None, scope, kind)
    B = memory.shape[0]
    max_it = int(round(float(mem_len.max()) * d.decoding_length_factor))
    ids = torch.full((B,), d.sos_id, dtype=torch.long)
    finished = torch.zeros(B, dtype=torch.bool)
    final_len = torch.zeros(B, dtype=torch.long)
    outs, samples = [], []
    for t in range(max_it):
        logits = sp.step(sp.embed(ids))
        sample = logits.argmax(-1)
        outs.append(logits)
        samples.append(sample)
        newly = (sample == d.eos_id) | finished
        final_len = torch.where(finished, final_len, torch.full_like(final_len, t + 1))
        finished = newly
        ids = sample
        if bool(finished.all()):
            break
    if not outs:
        return torch.zeros(B, 0, d.target_vocab_size, dtype=DT), torch.zeros(B, 0, dtype=torch.long), final_len, sp
    return torch.stack(outs, 1), torch.stack(samples, 1), final_len, sp


def gather_tree(step_ids, parent_ids, max_len, end_token):
    """tf.contrib.seq2seq.gather_tree (beam_search_ops): step_ids/parent_ids [T,B,K] -> full beams [T,B,K].
    Backtracks every final beam through its parents from t = max_len[b]-1; positions >= max_len[b] and everything
    after a beam's first end_token are end_token."""
    T, B, K = step_ids.shape
    out = torch.full_like(step_ids, end_token)
    for b in range(B):
        L = min(int(max_len[b]), T)
        for k in range(K):
            if L <= 0:
                continue
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


def speller_beam(hp: HP, params, memory, mem_len, enc_state, beam_width, mxu='f64', partial_targets=None):
    """las/model.py:219-226,298-319: tf.contrib.seq2seq.BeamSearchDecoder (length_penalty_weight 0, no coverage
    penalty) over the tiled batch, then gather_tree.  partial_targets [B,L] (features['partial_targets'],
    model_helper.py:203): get_partial_targets_state (las/model.py:299-307,351-361) first runs the decoder
    teacher-forced over the tiled tokens (TrainingHelper, full length for every row) and the search starts from that
    final state with start_tokens = partial_targets[:, 0].  Returns (predicted_ids [B,T,K], scores [B,T,K] per step,
    final lengths [B,K])."""
    d = hp.decoder
    B, K, V = memory.shape[0], beam_width, d.target_vocab_size
    tile = lambda x: x.repeat_interleave(K, 0)
    if isinstance(enc_state[0], tuple):
        st = tuple((tile(c), tile(h)) for c, h in enc_state)
    else:
        st = (tile(enc_state[0]), tile(enc_state[1]))
    sp = Speller(hp, params, tile(memory), tile(mem_len), st, mxu, False)
    max_it = int(round(float(mem_len.max()) * d.decoding_length_factor))
    ids = torch.full((B * K,), d.sos_id, dtype=torch.long)
    if partial_targets is not None:
        prefix = tile(partial_targets.long())
        for t in range(prefix.shape[1]):
            sp.step(sp.embed(prefix[:, t]))           # outputs dropped: only final_context_state is kept
        ids = prefix[:, 0]
    log_probs = torch.full((B, K), float('-inf'), dtype=DT)
    log_probs[:, 0] = 0.0
    finished = torch.zeros(B, K, dtype=torch.bool)
    lengths = torch.zeros(B, K, dtype=torch.long)
    NEG = torch.finfo(torch.float32).min
    step_ids, parents, step_scores = [], [], []
    for t in range(max_it):
        logits = sp.step(sp.embed(ids)).view(B, K, V)
        lp = torch.log_softmax(logits, -1)
        fin_row = torch.full((V,), NEG, dtype=DT)
        fin_row[d.eos_id] = 0.0
        lp = torch.where(finished.unsqueeze(-1), fin_row.view(1, 1, V), lp)       # _mask_probs
        total = log_probs.unsqueeze(-1) + lp
        scores, idx = torch.topk(total.view(B, K * V), K, dim=-1)                 # length penalty 0: score = log prob
        word, beam = idx % V, idx // V
        prev_fin = torch.gather(finished, 1, beam)
        lengths = torch.gather(lengths, 1, beam) + (~prev_fin).long()
        finished = prev_fin | (word == d.eos_id)
        log_probs = scores
        flat = (beam + torch.arange(B).unsqueeze(1) * K).view(-1)
        sp.cells = [(c[flat], h[flat]) for c, h in sp.cells]
        sp.attention, sp.align = sp.attention[flat], sp.align[flat]
        ids = word.view(-1)
        step_ids.append(word)
        parents.append(beam)
        step_scores.append(scores)
        if bool(finished.all()):
            break
    if not step_ids:
        z = torch.zeros(B, 0, K, dtype=torch.long)
        return z, z.to(DT), lengths
    sid, par = torch.stack(step_ids, 0), torch.stack(parents, 0)                  # [T,B,K]
    pred = gather_tree(sid, par, lengths.max(1).values, d.eos_id)
    return pred.permute(1, 0, 2), torch.stack(step_scores, 0).permute(1, 0, 2), lengths


# --------------------------------------------------------------------------------------
# losses  (model_helper.py:20-146)
# --------------------------------------------------------------------------------------
def sequence_loss(logits, targets, weights):
    """tf.contrib.seq2seq.sequence_loss defaults: sum(w*CE)/(sum(w)+1e-12) (Appendix A.8)."""
    lp = torch.log_softmax(logits, -1)
    ce = -lp.gather(-1, targets.unsqueeze(-1)).squeeze(-1)
    return (ce * weights).sum() / (weights.sum() + 1e-12)


def compute_loss_train(logits, targets, target_len):
    """model_helper.py:24-30."""
    B, U, V = logits.shape
    Ut = targets.shape[1]
    if U < Ut:
        logits = torch.cat([logits, torch.zeros(B, Ut - U, V, dtype=DT)], 1)
    w = (torch.arange(Ut).unsqueeze(0) < target_len.unsqueeze(1)).to(DT)
    return sequence_loss(logits, targets, w)


def compute_loss_eval(logits, targets, final_len, target_len, eos_id=EOS_ID):
    """model_helper.py:54-76: pad the shorter of (logits, targets) and weigh max(len) steps."""
    B, _, V = logits.shape
    max_ts, max_fs = int(target_len.max()), int(final_len.max())
    L = max(max_ts, max_fs)
    logits = logits[:, :max_fs]
    if targets.shape[1] < L:
        targets = torch.cat([targets, torch.full((B, L - targets.shape[1]), eos_id, dtype=torch.long)], 1)
    if logits.shape[1] < L:
        logits = torch.cat([logits, torch.zeros(B, L - logits.shape[1], V, dtype=DT)], 1)
    sl = torch.maximum(target_len, final_len)
    w = (torch.arange(L).unsqueeze(0) < sl.unsqueeze(1)).to(DT)
    return sequence_loss(logits[:, :L], targets[:, :L], w)


def sequence_loss_sigmoid(logits, targets, weights):
    """model_helper.py:81-95: mean over the features of sigmoid_cross_entropy_with_logits, weighted over the steps,
    divided by (sum of the weights + 1e-12)."""
    ce = torch.nn.functional.binary_cross_entropy_with_logits(logits, targets.to(logits.dtype), reduction='none').mean(-1)
    return (ce * weights).sum() / (weights.sum() + 1e-12)


def compute_loss_sigmoid_train(logits, targets_binf, target_len):
    """model_helper.py:102-105."""
    U = logits.shape[1]
    w = (torch.arange(U).unsqueeze(0) < target_len.unsqueeze(1)).to(DT)
    return sequence_loss_sigmoid(logits, targets_binf[:, :U], w)


def compute_loss_sigmoid_eval(logits, targets_binf, final_len, target_len):
    """model_helper.py:106-128 with the targets' FEATURE vectors (the reference passes the integer ids at :336-337, which
    its own reshape cannot take): cut the logits at the longest decoded length, zero-pad both to the longer, weigh
    max(target_len, final_len) steps per utterance."""
    B, _, nf = logits.shape
    max_ts, max_fs = int(target_len.max()), int(final_len.max())
    L = max(max_ts, max_fs)
    logits = logits[:, :max_fs]
    if targets_binf.shape[1] < L:
        targets_binf = torch.cat([targets_binf, torch.zeros(B, L - targets_binf.shape[1], nf, dtype=targets_binf.dtype)], 1)
    if logits.shape[1] < L:
        logits = torch.cat([logits, torch.zeros(B, L - logits.shape[1], nf, dtype=DT)], 1)
    w = (torch.arange(L).unsqueeze(0) < torch.maximum(target_len, final_len).unsqueeze(1)).to(DT)
    return sequence_loss_sigmoid(logits[:, :L], targets_binf[:, :L], w)


def ctc_loss_dense(logits, labels, label_len, logit_len, blank=0):
    """tf.nn.ctc_loss_v2 with dense labels, blank index 0 (model_helper.py:355-357; Appendix A.8).
    Log-space alpha recursion; returns per-example negative log likelihood [B]."""
    B, T, C = logits.shape
    lp = torch.log_softmax(logits, -1)
    losses = []
    NEG = torch.tensor(-1e30, dtype=DT)
    for b in range(B):
        L = int(label_len[b]); Tb = int(logit_len[b])
        ext = [blank]
        for s in labels[b, :L].tolist():
            ext += [int(s), blank]
        S = len(ext)
        alpha = [NEG] * S
        alpha[0] = lp[b, 0, ext[0]]
        if S > 1:
            alpha[1] = lp[b, 0, ext[1]]
        for t in range(1, Tb):
            new = []
            for s in range(S):
                terms = [alpha[s]]
                if s >= 1:
                    terms.append(alpha[s - 1])
                if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                    terms.append(alpha[s - 2])
                new.append(torch.logsumexp(torch.stack(terms), 0) + lp[b, t, ext[s]])
            alpha = new
        fin = [alpha[S - 1]] + ([alpha[S - 2]] if S > 1 else [])
        losses.append(-torch.logsumexp(torch.stack(fin), 0))
    return torch.stack(losses)


def ctc_greedy_decode(logits, logit_len):
    """tf.nn.ctc_greedy_decoder: blank = last class, merge repeated (model_helper.py:351-353)."""
    B, T, C = logits.shape
    best = logits.argmax(-1)
    res = []
    for b in range(B):
        prev, seq = -1, []
        for t in range(int(logit_len[b])):
            k = int(best[b, t])
            if k != prev and k != C - 1:
                seq.append(k)
            prev = k
        res.append(seq)
    return res


# --------------------------------------------------------------------------------------
# metric  (utils/metrics_utils.py:8-41)
# --------------------------------------------------------------------------------------
def _levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def dense_to_sparse_merge(row, eos_id):
    """utils/metrics_utils.py:8-28: append EOS, keep an element iff it differs from its successor
    (= last of each run), is before the first EOS and is not -1."""
    row = list(row)
    cat = row + [eos_id]
    first_eos = cat.index(eos_id)
    out = []
    for i, v in enumerate(row):
        if (cat[i + 1] - cat[i]) != 0 and i < first_eos and v != -1:
            out.append(v)
    return out


def edit_distance(hyp, truth, eos_id=EOS_ID, mapping=None):
    """utils/metrics_utils.py:31-41 -> tf.edit_distance(normalize=True) per row."""
    res = []
    for h, t in zip(hyp, truth):
        h, t = list(h), list(t)
        if mapping is not None:
            h = [mapping[i] for i in h]
            t = [mapping[i] for i in t]
        hs, ts = dense_to_sparse_merge(h, eos_id), dense_to_sparse_merge(t, eos_id)
        dist = _levenshtein(hs, ts)
        if len(ts) == 0:
            res.append(float('inf') if len(hs) else 0.0)
        else:
            res.append(dist / len(ts))
    return res


# --------------------------------------------------------------------------------------
# whole model + train op  (model_helper.py:165-444)
# --------------------------------------------------------------------------------------
def model_loss(hp: HP, params, batch, mxu='f64', stochastic=None):
    """Forward of las_model_fn in TRAIN mode (model_helper.py:205-227,319-357).  batch keys:
    encoder_inputs [B,T,F], source_sequence_length, targets_inputs, targets_outputs,
    target_sequence_length.  Returns (audio_loss, aux dict)."""
    x = batch['encoder_inputs']
    st = stochastic or {}
    (mem, mem_len), state = listener(x, batch['source_sequence_length'], params, hp.encoder, mxu,
                                     st.get('enc_masks'))
    gq = _g(make_q(mxu))
    loss, aux = None, {'memory': mem, 'memory_len': mem_len, 'state': state}
    for scope, kind in speller_plan(hp.decoder):                                 # model_helper.py:211-227,319-342
        logits, sp = speller_train(hp, params, mem, mem_len, state, batch['targets_inputs'],
                                   batch['target_sequence_length'], mxu, sample_select=st.get('sample_select'),
                                   sample_ids=st.get('sample_ids'), noise=st.get('att_noise'), in_masks=st.get('dec_masks'),
                                   scope=scope, kind=kind, sample_vecs=st.get('sample_vecs'))
        if kind == 'sigmoid':
            logits = gq(logits)
            tb = torch.as_tensor(hp.decoder.binf_map, dtype=DT).t()[batch['targets_outputs']]
            l_ = compute_loss_sigmoid_train(logits, tb, batch['target_sequence_length'])
            aux.setdefault('logits_binf', logits)
        elif kind == 'binf_projection':                                          # model_helper.py:251-253
            V = hp.decoder.target_vocab_size
            raw, logits = logits[..., V:], logits[..., :V]
            l_ = compute_loss_train(logits, batch['targets_outputs'], batch['target_sequence_length'])
            aux['ce_binf'] = l_
            reg = compute_log_probs_loss(raw)                                    # model_helper.py:327-331
            aux['log_probs_loss'] = reg
            l_ = l_ + reg * hp.decoder.binf_projection_reg_weight
            aux.setdefault('logits_binf', logits)
        else:
            logits = gq(logits)                          # the loss kernel writes d(logits) in bf16
            l_ = compute_loss_train(logits, batch['targets_outputs'], batch['target_sequence_length'])
        if loss is None:
            aux['logits'], aux['ce'] = logits, (aux['ce_binf'] if kind == 'binf_projection' else l_)
        loss = l_ if loss is None else loss + l_
    gq_ = gq
    if hp.ctc_weight > 0:
        q = make_q(mxu)
        cl = gq_(mem @ q(params['ctc_logits/kernel']) + params['ctc_logits/bias'])    # CTC d(logits) in bf16
        ctc = ctc_loss_dense(cl, batch['targets_outputs'], batch['target_sequence_length'], mem_len).mean()
        aux['ctc'] = ctc
        aux['ctc_logits'] = cl
        loss = loss + ctc * hp.ctc_weight
    return loss, aux


def compute_log_probs_loss(outputs):
    """model_helper.py:132-146: pushes [lp1 | lp0] towards normalised log-probabilities; mean over EVERY element
    (padded steps included); the stabilising constant carries no gradient."""
    nf = outputs.shape[-1] // 2
    a, b = outputs[..., :nf], outputs[..., nf:2 * nf]
    c = (-(a + b) / 2).detach()
    loss = torch.abs((torch.exp(a + c) + torch.exp(b + c)) / torch.exp(c) - 1)
    loss = loss + torch.relu(a) + torch.relu(b)
    return loss.mean()


def l2_term(params, scale):
    """tf.contrib.layers.l2_regularizer over ALL trainable vars (model_helper.py:411-413)."""
    return scale * sum((v * v).sum() for v in params.values()) / 2.0


def train_step(hp: HP, params, adam_m, adam_v, step, batch, mxu='f64', n_replicas=1,
               beta1=0.9, beta2=0.999, eps=1e-8, stochastic=None):
    """model_helper.py:403-417 (+405-406 for n_replicas>1 on ONE replica's shard): loss(+L2) ->
    grads -> clip_by_norm(g,2) per tensor -> Adam (TF form).  Returns dict with new params/m/v,
    loss, raw and clipped grads.  ``step`` is the 1-based Adam step t."""
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    audio, aux = model_loss(hp, leaf, batch, mxu, stochastic)
    loss = audio + l2_term(leaf, hp.l2_reg_scale)
    (loss / n_replicas).backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    clipped = {}
    for k, g in grads.items():
        n = torch.sqrt((g * g).sum())
        clipped[k] = g * GRAD_NORM / torch.maximum(n, torch.tensor(GRAD_NORM, dtype=DT))
    return {'loss': loss.detach(), 'audio_loss': audio.detach(), 'grads': grads, 'clipped': clipped,
            'aux': aux}


def adam_apply(params, adam_m, adam_v, grads, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update (Appendix A.8): epsilon outside the bias correction."""
    lr_t = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    new_p, new_m, new_v = {}, {}, {}
    for k in params:
        g = grads[k]
        m = beta1 * adam_m[k] + (1 - beta1) * g
        v = beta2 * adam_v[k] + (1 - beta2) * g * g
        new_p[k] = params[k] - lr_t * m / (torch.sqrt(v) + eps)
        new_m[k], new_v[k] = m, v
    return new_p, new_m, new_v


# --------------------------------------------------------------------------------------
# synthetic workload of SURVEY.md §8(d)
# --------------------------------------------------------------------------------------
def synthetic_batch(B=64, T=800, F=40, V=64, U=80, ragged=False, seed=1234):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T, F)).astype(np.float32)
    if ragged:
        src_len = np.array([T - 8 * (i % 26) for i in range(B)], dtype=np.int64)
        tgt_len = np.array([U - (i % 17) for i in range(B)], dtype=np.int64)
    else:
        src_len = np.full(B, T, dtype=np.int64)
        tgt_len = np.full(B, U, dtype=np.int64)
    src_len = np.maximum(src_len, 1)
    tgt_len = np.maximum(tgt_len, 1)
    tin = np.full((B, U), EOS_ID, dtype=np.int64)
    tout = np.full((B, U), EOS_ID, dtype=np.int64)
    for b in range(B):
        x[b, src_len[b]:] = 0.0
        n = tgt_len[b] - 1
        y = rng.integers(3, V, size=n)
        tin[b, 0] = SOS_ID
        tin[b, 1:n + 1] = y
        tout[b, :n] = y
        tout[b, n] = EOS_ID
    return {
        'encoder_inputs': torch.tensor(x.astype(np.float64)),
        'source_sequence_length': torch.tensor(src_len),
        'targets_inputs': torch.tensor(tin),
        'targets_outputs': torch.tensor(tout),
        'target_sequence_length': torch.tensor(tgt_len),
    }
