"""MI355X host side of the reference's ``model_helper.py``: ``las_model_fn`` (model_helper.py:165-444),
the losses (model_helper.py:20-78) and the train op (model_helper.py:403-417) over liblas_hip.so.

TF keeps variables in the graph; here ``LasModel`` owns them as ONE flat fp32 buffer (plus flat gradient
and Adam slot buffers) with per-tensor views named like the TF variables.  The train op is
L2 -> per-tensor clip_by_norm(2) -> [RCCL all-reduce(sum) when data-parallel] -> TF-form Adam,
i.e. CrossShardOptimizer's order (model_helper.py:405-417).
"""
import collections
import copy
import math
import os

import numpy as np
import torch

from . import dp, hip
from .las import model as las_model
from .las.ops import TRAIN, EVAL, PREDICT
from .utils import metrics_utils

__all__ = ['las_model_fn', 'LasModel', 'param_table', 'compute_loss', 'compute_loss_sigmoid', 'EstimatorSpec', 'GRAD_NORM']

GRAD_NORM = 2            # model_helper.py:16
EstimatorSpec = collections.namedtuple('EstimatorSpec', ['mode', 'loss', 'train_op', 'predictions', 'eval_metric_ops'])
EstimatorSpec.__new__.__defaults__ = (None, None, None, None)


def _enc_depth(e):
    dirs = 1 if e.unidirectional else 2
    if e.use_pyramidal:
        return dirs * e.num_units * (1 if e.num_layers == 1 else 2)
    return dirs * e.num_units


SUPPORTED_UNITS = (64, 128, 256, 512, 1024)      # what the recurrent / decoder kernels are built for (las/ops.py lstm_cell); 1024
# (round 5): chains of 32 workgroups, the decoder on its step-by-step launches -- the reference takes any width (las/ops.py:10-12)


def physical_units(n):
    """The width the kernels run a `n`-unit LSTM at: the next supported size.  The extra units have all-zero weights and
    biases, which keeps them at c = h = 0 for ever and their gradients at exactly 0 (see physical_params)."""
    for s in SUPPORTED_UNITS:
        if n <= s:
            return s
    raise ValueError('num_units %d: the HIP kernels go up to %d units' % (n, SUPPORTED_UNITS[-1]))


def physical_params(params):
    """The hparams the kernels run at.  The reference takes any --encoder_units / --decoder_units (las/ops.py:10-12); the
    kernels are built for SUPPORTED_UNITS, so other widths run zero-padded to the next one: a unit whose kernel columns,
    kernel rows and bias are zero has i = o = 1/2, j = 0, so c_t = f c_{t-1} = 0 and h_t = 0 from a zero state, every
    product it enters contributes 0.0 exactly, and its dz (hence every gradient element that touches it) is 0.0 exactly;
    L2, the per-tensor norms, the clip and Adam (m = v = 0) leave the padding at zero.  Returns `params` itself when
    nothing needs padding."""
    He, Hd = params.encoder.num_units, params.decoder.num_units
    if physical_units(He) == He and physical_units(Hd) == Hd:
        return params
    phys = copy.deepcopy(params)
    phys.encoder.set_hparam('num_units', physical_units(He))
    phys.decoder.set_hparam('num_units', physical_units(Hd))
    return phys


def _mem_atoms(e):
    """The listener's output depth as encoder-unit blocks: [fw | bw] per frame, two frames when pyramid-stacked."""
    dirs = 1 if e.unidirectional else 2
    if e.use_pyramidal:
        return ['H'] * (dirs * (1 if e.num_layers == 1 else 2))
    return ['H'] * dirs


def param_layout(params):
    """Ordered [(tf_variable_name, axes, initializer)]: every axis is a list of atoms, an int (a fixed extent) or 'H' /
    'D' (one block of encoder / decoder units).  param_table() turns the atoms into a shape for given unit counts; the
    same walk at the logical and the physical unit counts gives the index map of a zero-padded model."""
    e, d = params.encoder, params.decoder
    dirs = ['fw'] if e.unidirectional else ['fw', 'bw']
    out = []
    if bool(getattr(d, 'binary_outputs', False)) and bool(getattr(d, 'binf_trainable', False)):
        # --binf_trainable: the feature map is a variable, U(0, 1) initialised, created before the listener (model_helper.py:182-184)
        out.append(('binf2phone', ([d.binf_count], [d.target_vocab_size]), 'uniform01'))
    D = [params.num_channels]
    for l in range(e.num_layers):
        for dr in dirs:
            base = ('listener/bilstm_%d/%s/lstm_cell' % (l, dr)) if e.use_pyramidal else \
                ('listener/%s/multi_rnn_cell/cell_%d/lstm_cell' % (dr, l))
            out.append((base + '/kernel', (D + ['H'], ['H'] * 4), 'lstm'))
            out.append((base + '/bias', (['H'] * 4,), 'zeros'))
        D = ['H'] * (len(dirs) * (1 if l == 0 else 2)) if e.use_pyramidal else ['H']
    M, V = _mem_atoms(e), d.target_vocab_size
    for scope, kind in speller_plan(d):
        out.extend(_speller_layout(d, M, scope, kind))
    if params.ctc_weight > 0:
        out.append(('ctc_logits/kernel', (M, [V + 1]), 'glorot'))
        out.append(('ctc_logits/bias', ([V + 1],), 'zeros'))
    return out


def _extent(axis, H, Hd):
    return sum(H if a == 'H' else Hd if a == 'D' else int(a) for a in axis)


def param_table(params):
    """Ordered [(tf_variable_name, shape, initializer)] of every trainable variable the model_fn creates."""
    H, Hd = params.encoder.num_units, params.decoder.num_units
    return [(name, tuple(_extent(a, H, Hd) for a in axes), init) for name, axes, init in param_layout(params)]


def pad_index_maps(logical, physical):
    """{name: [per axis: int64 array, logical index -> physical index]} between the same model at its logical unit counts
    and at the widths the kernels run (physical_params)."""
    Hl, Dl = logical.encoder.num_units, logical.decoder.num_units
    Hp, Dp = physical.encoder.num_units, physical.decoder.num_units
    maps = {}
    for name, axes, _ in param_layout(logical):
        per_axis = []
        for axis in axes:
            idx, o = [], 0
            for a in axis:
                nl, npad = (Hl, Hp) if a == 'H' else (Dl, Dp) if a == 'D' else (int(a), int(a))
                idx.append(np.arange(o, o + nl, dtype=np.int64))
                o += npad
            per_axis.append(np.concatenate(idx))
        maps[name] = per_axis
    return maps


def speller_plan(d):
    """[(variable scope, kind)] of the decoders las_model_fn builds (model_helper.py:211-227): kind 'phones' (softmax over
    the vocabulary), 'binf_projection' (--binary_outputs --binf_projection: DenseBinfDecoder's fixed feature-to-phone map)
    or 'sigmoid' (--binary_outputs alone: feature logits).  A binary model has ONE decoder unless --multitask adds the
    phone decoder in front of it.  Scopes: the reference names the binary decoder 'speller_binf' in every case; here a
    single decoder always lives under 'speller' (TF checkpoints cannot be loaded either way) and only the second decoder
    of a multitask model under 'speller_binf'."""
    binary = bool(getattr(d, 'binary_outputs', False))
    kind = 'binf_projection' if getattr(d, 'binf_projection', False) else 'sigmoid'
    if not binary:
        return [('speller', 'phones')]
    if getattr(d, 'multitask', False):
        return [('speller', 'phones'), ('speller_binf', kind)]
    return [('speller', kind)]


def tf_variable_name_map(params):
    """{product variable name: name of the same tensor in the reference's TF-1 graph}, for a checkpoint importer.  Layouts are
    identical (kernel [D+H, 4H], gate order i, j, f, o, zero bias with the forget bias added in the cell); only names differ.

    VERIFIED against SURVEY A.1 (the names the reference's graph gives the listener, `las/ops.py:23-46`, `las/model.py:113-141`):
      pyramidal      listener/bilstm_{l}/{fw,bw}/lstm_cell/X   ->  listener/bilstm_{l}/bidirectional_rnn/{fw,bw}/lstm_cell/X
      (unidirectional: tf.nn.dynamic_rnn's default scope)      ->  listener/bilstm_{l}/rnn/lstm_cell/X
      stacked        listener/{fw,bw}/multi_rnn_cell/cell_{l}/lstm_cell/X
                                                               ->  listener/bidirectional_rnn/{fw,bw}/multi_rnn_cell/cell_{l}/lstm_cell/X
      (unidirectional)                                         ->  listener/rnn/multi_rnn_cell/cell_{l}/lstm_cell/X
    The `fw_cell` / `bw_cell` scopes of `las/ops.py:30,33` leave no trace: cells create their variables at their first call, inside
    the rnn scope.  `ctc_logits/*` is a tf.layers.dense under its own name (`model_helper.py:350`).
    NOT VERIFIED (TensorFlow is not installed here and the reference ships no checkpoint): the speller.  Its variables are
    created inside `dynamic_decode`'s `decoder` scope by AttentionWrapper; the values returned for them follow TF-1.15's scoping
    rules as written in tf.contrib.seq2seq and are marked by UNVERIFIED_TF_NAMES -- an importer must check them against
    `tf.train.list_variables(checkpoint)` before trusting them."""
    e, d = params.encoder, params.decoder
    rnn = 'rnn' if e.unidirectional else 'bidirectional_rnn'
    out = {}
    mech = {'luong': 'luong_attention', 'bahdanau': 'bahdanau_attention', 'luong_monotonic': 'luong_monotonic_attention',
            'bahdanau_monotonic': 'bahdanau_monotonic_attention', 'custom': 'CustomAttention'}[d.attention_type]
    for name, _, _ in param_layout(params):
        parts = name.split('/')
        if parts[0] == 'listener':
            if e.use_pyramidal:                    # listener/bilstm_l/<dir>/lstm_cell/X
                mid = [rnn] if e.unidirectional else [rnn, parts[2]]
                out[name] = '/'.join(parts[:2] + mid + parts[3:])
            else:                                  # listener/<dir>/multi_rnn_cell/cell_l/lstm_cell/X
                mid = [rnn] if e.unidirectional else [rnn, parts[1]]
                out[name] = '/'.join(parts[:1] + mid + parts[2:])
        elif parts[0] in ('speller', 'speller_binf'):
            scope, rest = parts[0], parts[1:]
            if rest[0] == 'memory_layer':          # built with the mechanism, before the decoder scope exists
                tf_name = [scope, 'memory_layer', rest[1]]
            elif rest[0] in ('query_layer', 'attention_v', 'attention_score_bias'):
                tf_name = [scope, 'decoder', 'attention_wrapper', mech] + rest
            elif rest[0] == 'attention_layer':
                tf_name = [scope, 'decoder', 'attention_wrapper', 'attention_layer', rest[1]]
            elif rest[0].startswith('decoder_cell_'):
                l = int(rest[0].rsplit('_', 1)[1])
                if d.bottom_only:                  # AttentionMultiCell: cell 0 inside the wrapper, the others as cell_{l-1} of the multi cell
                    cell = ['attention_multi_cell'] + (['attention_wrapper'] if l == 0 else ['cell_%d' % (l - 1)]) if d.num_layers > 1 \
                        else ['attention_wrapper']
                else:
                    cell = ['attention_wrapper'] + (['multi_rnn_cell', 'cell_%d' % l] if d.num_layers > 1 else [])
                tf_name = [scope, 'decoder'] + cell + rest[1:]
            elif rest[0] == 'projection_layer':
                tf_name = [scope, 'decoder', 'projection_layer'] + rest[1:]
            else:                                  # target_embedding: tf.get_variable in the speller's own scope
                tf_name = [scope] + rest
            out[name] = '/'.join(tf_name)
        else:                                      # binf2phone, ctc_logits/*
            out[name] = name
    return out


def UNVERIFIED_TF_NAMES(params):
    """The entries of tf_variable_name_map() that could not be checked against the reference (see there)."""
    return sorted(n for n in tf_variable_name_map(params) if n.split('/')[0] in ('speller', 'speller_binf'))


def _speller_layout(d, M, scope, kind):
    """param_layout's rows of one decoder; M: the memory depth as atoms (_mem_atoms)."""
    V = d.target_vocab_size
    out = []
    A = [2 * d.binf_count] if kind == 'binf_projection' else ([d.attention_layer_size] if d.attention_layer_size else list(M))
    E = d.embedding_size if d.embedding_size else V
    if kind != 'phones' and not d.embedding_size:
        E = d.binf_count            # embedding_fn = rows of binf2phone^T / the feature vector itself (las/model.py:237-243)
    Vo = d.binf_count if kind == 'sigmoid' else V          # DenseBinfDecoder(binf_count units) (las/model.py:251-252)
    if d.embedding_size:
        out.append((scope + '/target_embedding', ([V], [d.embedding_size]), 'glorot'))
    out.append((scope + '/memory_layer/kernel', (list(M), ['D']), 'glorot'))
    if d.attention_type in ('bahdanau', 'bahdanau_monotonic', 'custom'):
        out.append((scope + '/query_layer/kernel', (['D'], ['D']), 'glorot'))
    if d.attention_type in ('bahdanau', 'bahdanau_monotonic'):
        out.append((scope + '/attention_v', (['D'],), 'glorot_v'))
    if d.attention_type in ('luong_monotonic', 'bahdanau_monotonic'):
        out.append((scope + '/attention_score_bias', ([1],), 'zeros'))
    if d.attention_layer_size or kind == 'binf_projection':
        out.append((scope + '/attention_layer/kernel', (['D'] + list(M), A), 'glorot'))
    for l in range(d.num_layers):
        if d.bottom_only:       # AttentionMultiCell: cell_1 reads [attention_t, attention_{t-1}], upper cells [h_{l-1}, attention_{t-1}]
            din = ([E] + A) if l == 0 else ((A + A) if l == 1 else (['D'] + A))
        else:
            din = ([E] + A) if l == 0 else ['D']
        out.append((scope + '/decoder_cell_%d/lstm_cell/kernel' % l, (din + ['D'], ['D'] * 4), 'lstm'))
        out.append((scope + '/decoder_cell_%d/lstm_cell/bias' % l, (['D'] * 4,), 'zeros'))
    P = ['D'] if (d.bottom_only and d.num_layers > 1) else A      # AttentionMultiCell with upper layers outputs h_top
    out.append((scope + '/projection_layer/kernel', (P, [Vo]), 'proj'))
    out.append((scope + '/projection_layer/bias', ([Vo],), 'zeros'))
    return out


def _speller_table(d, M, scope, kind):
    """[(name, shape, init)] of one decoder on a memory of depth M (an int)."""
    Hd = d.num_units
    return [(n, tuple(_extent(a, M, Hd) for a in axes), i) for n, axes, i in _speller_layout(d, ['H'], scope, kind)]


def _init_array(shape, init, rng):
    if init in ('lstm', 'proj'):                       # las/ops.py:12, las/model.py:257
        return rng.uniform(-0.075, 0.075, size=shape)
    if init == 'glorot':                               # tf.layers.Dense default kernel initializer
        lim = math.sqrt(6.0 / (shape[0] + shape[1]))
        return rng.uniform(-lim, lim, size=shape)
    if init == 'glorot_v':
        lim = math.sqrt(6.0 / (shape[0] + 1))
        return rng.uniform(-lim, lim, size=shape)
    if init == 'uniform01':                            # tf.random_uniform_initializer(0., 1.): the trainable binf2phone map
        return rng.uniform(0.0, 1.0, size=shape)
    return np.zeros(shape)


class Variables:
    """Flat fp32 parameter / gradient / Adam-slot buffers with TF-named per-tensor views."""

    def __init__(self, table, device='cuda', logical_table=None, index_maps=None):
        """table: the tensors as the kernels see them.  logical_table / index_maps (pad_index_maps): set when the model
        runs zero-padded to wider LSTMs than its hparams say; initialize / load / logical() then speak the logical shapes."""
        self.table = list(table)
        self.logical_table = list(logical_table) if logical_table is not None else None
        self.index_maps = index_maps
        offs = [0]
        for _, shape, _ in self.table:
            n = int(np.prod(shape))
            offs.append(offs[-1] + (n + 3) // 4 * 4)          # keep every tensor 16-byte aligned
        self.offsets = offs
        self.total = offs[-1]
        f32 = torch.float32
        self.flat = torch.zeros(self.total, dtype=f32, device=device)
        # gradient buffer with a 4-float head: grad_store[0] is the step's "a persistent kernel timed out" flag
        # (las_status_collect).  It is all-reduced together with the gradients, so every replica skips the Adam update
        # when any of them saw a timeout (las_adam_update's skip_flag).
        self.grad_store = torch.zeros(self.total + 4, dtype=f32, device=device)
        self.grad = self.grad_store[4:]
        self.skip_flag = self.grad_store[:1]
        self.m = torch.zeros(self.total, dtype=f32, device=device)
        self.v = torch.zeros(self.total, dtype=f32, device=device)
        self.seg = torch.tensor(offs, dtype=torch.int64, device=device)
        self.device = device
        self.sumsq = torch.zeros(len(self.table), dtype=f32, device=device)
        self.param_sumsq = torch.zeros(2, dtype=f32, device=device)      # sum theta^2 per bucket, by-product of the norms pass
        self.buckets = [self._bucket(0, len(self.table), 0)]             # one exchange bucket = everything
        self.params = self._views(self.flat)
        self.grads = self._views(self.grad)
        self.mask = None
        if self.index_maps is not None:
            # 1.0 where a logical element lives, 0.0 on the padding (weight noise must not wake the padded units up)
            self.mask = torch.zeros(self.total, dtype=f32, device=device)
            for name, view in self._views(self.mask).items():
                self._scatter(view, name, torch.ones(1, dtype=f32, device=device).expand(*self._logical_shape(name)))

    def _bucket(self, lo, hi, slot):
        """Tensors lo..hi-1 as a contiguous piece of the flat buffers: element range, boundaries relative to its start."""
        o = self.offsets
        rel = torch.tensor([x - o[lo] for x in o[lo:hi + 1]], dtype=torch.int64, device=self.device)
        return dict(lo=lo, hi=hi, begin=o[lo], end=o[hi], seg=rel, slot=slot)

    def split_buckets(self, first_of_last):
        """Two exchange buckets in the order the backward pass completes them: tensors [first_of_last, end) first (top
        listener layer, speller, CTC head), then [0, first_of_last).  At most two (param_sumsq has two slots)."""
        n = len(self.table)
        if 0 < first_of_last < n:
            self.buckets = [self._bucket(first_of_last, n, 0), self._bucket(0, first_of_last, 1)]
        return self.buckets

    def exchange_view(self, bucket=None):
        """The piece of the gradient storage that is all-reduced for `bucket` (None: everything).  The timeout flag in
        front of the gradients travels with the piece that starts at element 0 -- of two buckets the LAST to leave, so
        the flag covers every kernel of the step."""
        if bucket is None:
            return self.grad_store
        if bucket['begin'] == 0:
            return self.grad_store[:4 + bucket['end']]
        return self.grad[bucket['begin']:bucket['end']]

    def _views(self, flat):
        d = collections.OrderedDict()
        for (name, shape, _), o in zip(self.table, self.offsets):
            d[name] = flat[o:o + int(np.prod(shape))].view(*shape)
        return d

    def _logical_shape(self, name):
        return tuple(len(i) for i in self.index_maps[name])

    def _index(self, name):
        """Broadcastable index tensors of the logical elements inside the physical tensor."""
        maps = self.index_maps[name]
        n = len(maps)
        return tuple(torch.as_tensor(m, device=self.device).view(*[(-1 if k == a else 1) for k in range(n)])
                     for a, m in enumerate(maps))

    def _scatter(self, view, name, value):
        view.zero_()
        view[self._index(name)] = value.to(device=view.device, dtype=view.dtype)

    def initialize(self, seed=4321):
        """The initial values are drawn at the LOGICAL shapes (a zero-padded model starts from the same weights as an
        un-padded one of its hparams would)."""
        rng = np.random.default_rng(seed)
        for name, shape, init in (self.logical_table or self.table):
            a = torch.from_numpy(_init_array(shape, init, rng).astype(np.float32))
            if self.index_maps is None:
                self.params[name].copy_(a)
            else:
                self._scatter(self.params[name], name, a)

    def load(self, tensors):
        """tensors: {name: array} at the logical shapes (or, for a padded model, at the physical ones)."""
        for name in self.params:
            t = torch.as_tensor(tensors[name]).to(torch.float32)
            if tuple(t.shape) == tuple(self.params[name].shape):
                self.params[name].copy_(t)
            elif self.index_maps is not None and tuple(t.shape) == self._logical_shape(name):
                self._scatter(self.params[name], name, t)
            else:
                raise ValueError('%s: got shape %s, the model holds %s' % (name, tuple(t.shape), tuple(self.params[name].shape)))

    def logical(self, which='params'):
        """{name: tensor} of `which` ('params' | 'grads' | 'm' | 'v') at the logical shapes (copies for a padded model)."""
        views = {'params': self.params, 'grads': self.grads}.get(which)
        if views is None:
            views = self._views(getattr(self, which))
        if self.index_maps is None:
            return views
        return collections.OrderedDict((n, t[self._index(n)]) for n, t in views.items())

    def padding_is_zero(self, which='params'):
        """True when every padded element of `which` is exactly 0.0 (always True for an un-padded model)."""
        if self.mask is None:
            return True
        flat = {'params': self.flat, 'grads': self.grad}.get(which)
        if flat is None:
            flat = getattr(self, which)
        return bool((flat * (1.0 - self.mask)).abs().max().item() == 0.0)

    def num_parameters(self):
        return sum(int(np.prod(s)) for _, s, _ in (self.logical_table or self.table))


def compute_loss(logits, targets, final_sequence_length, target_sequence_length, mode, eos_id, grad_scale=1.0,
                 want_grad=False, vocab=None):
    """model_helper.py:20-78.  logits fp32 [B,U,ldl] on the device; returns (loss scalar tensor, dlogits or None).
    TRAIN: weights = sequence_mask(target_len) (model_helper.py:24-30).  EVAL: logits/targets padded to the longer
    of (target_len, final_len) with zeros / EOS (model_helper.py:54-76)."""
    B, U, ldl = logits.shape
    V = vocab if vocab is not None else ldl
    dev = logits.device
    if mode != TRAIN:
        max_ts = int(target_sequence_length.max().item())
        max_fs = int(final_sequence_length.max().item())
        L = max(max_ts, max_fs)
        lg = torch.zeros(B, L, ldl, dtype=torch.float32, device=dev)
        n = min(max_fs, U)
        lg[:, :n] = logits[:, :n]
        tg = torch.full((B, L), eos_id, dtype=torch.int32, device=dev)
        n = min(L, targets.shape[1])
        tg[:, :n] = targets[:, :n]
        lens = torch.maximum(target_sequence_length.to(torch.int32), final_sequence_length.to(torch.int32))
        logits, targets, target_sequence_length, U = lg, tg, lens, L
    tg = targets[:, :U].to(torch.int32).contiguous()
    if tg.shape[1] < U:
        raise ValueError('targets shorter than the decoded length')
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dlogits = torch.empty(B, U, ldl, dtype=torch.bfloat16, device=dev) if want_grad else None
    hip.fill_many(zero=[loss] + ([dlogits] if want_grad else []))
    tlen32 = target_sequence_length.to(torch.int32)     # named: a temporary would be recycled before the launch
    hip.check(hip.lib().las_seq_ce_loss(hip.p(logits), ldl, hip.p(tg), hip.p(tlen32),
                                        B, U, V, float(grad_scale), hip.p(loss), hip.p(dlogits), ldl, hip.stream()))
    return loss, dlogits


def compute_loss_sigmoid(logits, targets_binf, final_sequence_length, target_sequence_length, mode, nf, grad_scale=1.0,
                         want_grad=False):
    """model_helper.py:98-130 over sequence_loss_sigmoid (:81-95).  logits fp32 [B,U,ldl] (first nf columns: feature
    logits), targets_binf bf16 0/1 [B,Ut,ldt].  TRAIN: weights = sequence_mask(target_len).  EVAL: logits cut at the
    longest decoded length, both padded with zeros to the longer of (targets, decoded), weights over max(target_len,
    final_len) per utterance.  Returns (loss [1], dlogits bf16 or None)."""
    B, U, ldl = logits.shape
    dev = logits.device
    if mode != TRAIN:
        max_ts = int(target_sequence_length.max().item())
        max_fs = int(final_sequence_length.max().item())
        L = max(max_ts, max_fs)
        lg = torch.zeros(B, L, ldl, dtype=torch.float32, device=dev)
        n = min(max_fs, U)
        lg[:, :n] = logits[:, :n]
        tg = torch.zeros(B, L, targets_binf.shape[-1], dtype=torch.bfloat16, device=dev)
        n = min(L, targets_binf.shape[1])
        tg[:, :n] = targets_binf[:, :n]
        lens = torch.maximum(target_sequence_length.to(torch.int32), final_sequence_length.to(torch.int32))
        logits, targets_binf, target_sequence_length, U = lg, tg, lens, L
    tg = targets_binf[:, :U].contiguous()
    if tg.shape[1] < U:
        raise ValueError('targets shorter than the decoded length')
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    dlogits = torch.zeros(B, U, ldl, dtype=torch.bfloat16, device=dev) if want_grad else None
    tlen32 = target_sequence_length.to(torch.int32)     # named: a temporary would be recycled before the launch
    hip.check(hip.lib().las_seq_sigmoid_loss(hip.p(logits), ldl, hip.p(tg), tg.shape[-1], hip.p(tlen32), B, U, nf,
                                             float(grad_scale), hip.p(loss), hip.p(dlogits), ldl, hip.stream()))
    return loss, dlogits


class CtcHead:
    """Dense(M -> V+1) on the encoder outputs + tf.nn.ctc_loss_v2 (model_helper.py:347-358): blank index 0, labels =
    targets_outputs incl. </s> with label_length = target_sequence_length, logit_length = reduced source length,
    mean over the batch, times ctc_weight."""
    K, Bn = 'ctc_logits/kernel', 'ctc_logits/bias'

    def __init__(self, params, variables, M):
        self.w = float(params.ctc_weight)
        self.M, self.C = M, params.decoder.target_vocab_size + 1
        self.Cp = (self.C + 7) // 8 * 8
        bf = torch.bfloat16
        self.wT = torch.empty(self.Cp, M, dtype=bf, device='cuda')
        self.wn = torch.empty(M, self.Cp, dtype=bf, device='cuda')
        self.bias = torch.zeros(self.Cp, dtype=torch.float32, device='cuda')
        self.refresh(variables)

    def refresh(self, var):
        hip.cast_bf16(var[self.K], self.M, self.C, self.wT, self.Cp, self.M, transpose=True)
        hip.cast_bf16(var[self.K], self.M, self.C, self.wn, self.M, self.Cp)
        self.bias[:self.C].copy_(var[self.Bn])

    def forward(self, memory, mem_len, targets, target_len, loss_out, grad_scale):
        B, Tm, M = memory.shape
        dev = memory.device
        lib = hip.lib()
        self.logits = torch.empty(B, Tm, self.Cp, dtype=torch.float32, device=dev)
        hip.gemm_nt(memory, self.wT, self.logits, B * Tm, self.Cp, M, lda=M, ldb=M, ldc=self.Cp, bias=self.bias)
        U = targets.shape[1]
        ws = torch.empty(lib.las_ctc_workspace_bytes(B, Tm, self.Cp, U), dtype=torch.uint8, device=dev)
        self.dlogits = torch.empty(B, Tm, self.Cp, dtype=torch.bfloat16, device=dev)
        self.per_example = torch.empty(B, dtype=torch.float32, device=dev)
        tg = targets.to(torch.int32).contiguous()
        tl = target_len.to(torch.int32)                 # named: a temporary would be recycled before the launch
        hip.check(lib.las_ctc_loss(hip.p(self.logits), self.Cp, hip.p(tg), U, hip.p(tl),
                                   hip.p(mem_len), B, Tm, self.C, U, 0, self.w / B, self.w / B * grad_scale, hip.p(ws),
                                   hip.p(loss_out), hip.p(self.per_example), hip.p(self.dlogits), hip.stream()))
        self.memory = memory

    def backward(self, dmem, grads):
        B, Tm, M = self.memory.shape
        hip.gemm_nt(self.dlogits, self.wn, dmem, B * Tm, M, self.Cp, lda=self.Cp, ldb=self.Cp, ldc=M, accumulate=True)
        hip.gemm_tn(self.memory, self.dlogits, grads[self.K], M, self.C, B * Tm, lda=M, ldb=self.Cp, ldc=self.C, split_k=4)
        hip.colsum_bf16(self.dlogits, B * Tm, self.C, grads[self.Bn], ldx=self.Cp)


def ctc_greedy_decode(logits, logit_len):
    """tf.nn.ctc_greedy_decoder (model_helper.py:351-353): blank = LAST class, merge repeated; host-side metric."""
    best = logits.argmax(-1).cpu().tolist()
    C = logits.shape[-1]
    out = []
    for b, row in enumerate(best):
        prev, seq = -1, []
        for t in range(int(logit_len[b])):
            k = row[t]
            if k != prev and k != C - 1:
                seq.append(k)
            prev = k
        out.append(seq)
    return out


class LasModel:
    """Variables + listener + speller + train op: what tf.estimator.Estimator(model_fn=las_model_fn) holds."""

    def __init__(self, params, seed=4321, world_size=1, process_group=None, binf2phone=None, rank=0):
        """binf2phone: the [binf_count, V] 0/1 matrix of utils.load_binf2phone for the binary-feature decoders
        (model_helper.py:181-187: a constant unless --binf_trainable).  rank: this replica's index in a data-parallel
        job; it enters the seed of the dropout / sampling draws (replicas draw independently, as CrossShardOptimizer's
        do) but not the weight-noise seed (the weights stay identical)."""
        if not torch.cuda.is_available():
            raise hip.LasError('no HIP device visible: the LAS path has no CPU fallback')
        hip.lib()
        # any --encoder_units / --decoder_units: widths the kernels are not built for run zero-padded to the next one that
        # is (physical_params); self.params is what the modules run at, self.hparams what the caller asked for
        self.hparams = params
        params = physical_params(params)
        self.params = params
        self.padded = params is not self.hparams
        d = params.decoder
        binary = bool(getattr(d, 'binary_outputs', False))
        self.binf_projection = binary and bool(getattr(d, 'binf_projection', False))
        self.sigmoid = binary and not self.binf_projection       # feature-logit outputs (a12 / a14 of SURVEY.md 8a)
        self.binf_trainable = binary and bool(getattr(d, 'binf_trainable', False))
        if binary:
            if getattr(d, 'binf_sampling', False):
                # not a gap of this implementation: the flag has no well-formed graph in the reference
                raise ValueError(
                    '--binf_sampling: the reference cannot build this graph in any flag combination -- with --binf_projection '
                    'las.model.speller gets binf_embedding=None and its embedding_fn transposes it (las/model.py:241-243); '
                    'without, las_model_fn applies transform_binf_to_phones to binf_count-wide logits whose [nf:2nf] half is '
                    'empty (model_helper.py:247-249, utils/training_helper.py:17-27).  Drop the flag.')
            if self.binf_trainable and not self.binf_projection:
                raise ValueError('--binf_trainable needs --binf_projection on the HIP path: for the sigmoid-output decoder the '
                                 'reference differentiates the loss through its TARGETS as well (targets_binf is a lookup in the '
                                 'variable, model_helper.py:199)')
            if self.binf_projection and binf2phone is None:
                raise ValueError('binf_projection needs the binf2phone matrix (--binf_map)')
            if binf2phone is not None:
                binf2phone = torch.as_tensor(np.asarray(binf2phone), dtype=torch.float32)
                if tuple(binf2phone.shape) != (d.binf_count, d.target_vocab_size):
                    raise ValueError('binf2phone must be [binf_count=%d, target_vocab_size=%d], got %s'
                                     % (d.binf_count, d.target_vocab_size, tuple(binf2phone.shape)))
        if (self.padded and self.binf_projection and d.bottom_only and d.num_layers > 1
                and physical_units(self.hparams.decoder.num_units) != self.hparams.decoder.num_units):
            # the raw outputs are then the top cell's h, and compute_log_probs_loss splits them at HALF their width
            # (model_helper.py:137-139): a zero-padded h would move that split
            raise ValueError('binf_projection on a multi-layer --bottom_only decoder: decoder_units must be one of the widths the '
                             'kernels are built for (64, 128, 256, 512, 1024), not a zero-padded one (got %d)' % self.hparams.decoder.num_units)
        if self.padded:
            self.vars = Variables(param_table(params), logical_table=param_table(self.hparams),
                                  index_maps=pad_index_maps(self.hparams, params))
        else:
            self.vars = Variables(param_table(params))
        self.vars.initialize(seed)
        self.listener = las_model.Listener(params.encoder, self.vars.params, params.num_channels)
        # the decoders of model_helper.py:211-227: [(module, kind)]; self.speller is the first one (the only one unless
        # --multitask), self.speller_binf the binary decoder of a multitask model
        self.spellers = []
        for scope, kind in speller_plan(d):
            mod = las_model.make_speller(d, self.vars.params, _enc_depth(params.encoder),
                                         binf2phone=binf2phone if kind != 'phones' else None, scope=scope,
                                         phones_only=(kind == 'phones'),
                                         binf_var='binf2phone' if (self.binf_trainable and kind != 'phones') else None)
            self.spellers.append((mod, kind))
        self.speller = self.spellers[0][0]
        self.speller_binf = self.spellers[1][0] if len(self.spellers) > 1 else None
        self.ctc = CtcHead(params, self.vars.params, _enc_depth(params.encoder)) if params.ctc_weight > 0 else None
        self.global_step = 0
        self.rng_seed = (seed * 2654435761 + 12345) & 0x7fffffff      # base of the dropout / sampling draws
        self.noise_seed = self.rng_seed                               # weight noise: the same on every replica
        self.rng_seed = (self.rng_seed ^ ((rank * 0x9E3779B9) & 0x7fffffff)) & 0x7fffffff
        self.step_dev = torch.ones(1, dtype=torch.int32, device='cuda')       # Adam t = global_step + 1
        self.world_size = world_size
        self.process_group = process_group
        self._images_stale = False
        # LAS_SERIAL=1 (diagnostics, A/B timing): no side streams -- every product runs where it is issued, on the one stream
        self.overlap = las_model.ops.Overlap() if os.environ.get('LAS_SERIAL', '0') != '1' else las_model.ops._NoOverlap()
        self.tail_buckets = self._tail_buckets()

    def _tail_buckets(self):
        """Single replica, pyramidal listener: [everything above the bottom listener layer] and [the bottom layer (and
        whatever the table holds in front of it)] as two pieces of the flat buffers.  The bottom layer's weight-gradient
        products are the last thing a backward pass computes and nothing hides them; apply_gradients() runs norms + clip +
        Adam of the first piece beside them and only the second piece after them.  LAS_TAIL_OVERLAP=0: one pass, as before."""
        e = self.params.encoder
        if (self.world_size != 1 or self.process_group is not None or not e.use_pyramidal
                or os.environ.get('LAS_TAIL_OVERLAP', '1') == '0'):
            return None
        names = [n for n, _, _ in self.vars.table]
        lead = next(i for i, n in enumerate(names) if n.startswith('listener/'))
        k = lead + 2 * (1 if e.unidirectional else 2)
        if not all(n.startswith('listener/bilstm_0/') for n in names[lead:k]) or k >= len(names):
            return None
        return [self.vars._bucket(k, len(names), 0), self.vars._bucket(0, k, 1)]

    # -- weights --------------------------------------------------------------------------------
    def load_variables(self, tensors):
        self.vars.load(tensors)
        self.refresh_images()

    def refresh_images(self, beside_first_layer=False):
        """Rebuild the bf16 operand images from the fp32 master weights.  beside_first_layer: only the bottom listener
        layer's images now; returns (start, wait) for Listener.forward: start() -- called between that layer's input
        projection and its recurrence -- enqueues everything else on the second stream, where it runs beside the
        recurrence (which leaves a third of the CUs idle); wait() makes the current stream wait for it in front of the
        second layer.  Otherwise everything on the current stream; returns None."""
        hooks = None
        if beside_first_layer and len(self.listener.layers) > 1 and os.environ.get('LAS_REFRESH_BESIDE', '1') != '0':
            self.listener.refresh(self.vars.params, layers=[0])
            done = []

            def start():
                with self.overlap.fork(beside_chain=True):
                    self.listener.refresh(self.vars.params, layers=range(1, len(self.listener.layers)))
                    for mod, _ in self.spellers:
                        mod.refresh(self.vars.params)
                    if self.ctc is not None:
                        self.ctc.refresh(self.vars.params)
                    done.append(self.overlap.mark())

            hooks = (start, lambda: done[0].wait())
        else:
            self.listener.refresh(self.vars.params)
            for mod, _ in self.spellers:
                mod.refresh(self.vars.params)
            if self.ctc is not None:
                self.ctc.refresh(self.vars.params)
        self._images_stale = False
        return hooks

    # -- forward / backward ---------------------------------------------------------------------
    def forward_train(self, features, labels, num_steps=None):
        """Forward of las_model_fn in TRAIN mode.  Returns (audio_loss [1], logits [B,U,Vp], dlogits)."""
        images_ready = self.refresh_images(beside_first_layer=True) if self._images_stale else None
        x = features['encoder_inputs']
        src_len = features['source_sequence_length']
        tin, tout, tlen = labels['targets_inputs'], labels['targets_outputs'], labels['target_sequence_length']
        step_seed = (self.rng_seed + 7919 * self.global_step) & 0x7fffffff   # fresh draws every optimiser step
        self.last_seed = step_seed
        (mem, mem_len), state = self.listener.forward(x, src_len, TRAIN, seed=step_seed, after_first_layer=images_ready)
        U = num_steps if num_steps is not None else int(tlen.max().item())
        loss, logits, dlogits = None, None, []
        for mod, kind in self.spellers:          # audio_loss = sum of the decoders' losses (model_helper.py:337-342)
            extra = {'overlap': self.overlap} if isinstance(mod, las_model.Speller) else {}
            if extra and kind == 'phones':
                # the fused speller forms the loss itself when it can (las_proj_ce: projection + loss + product back in one launch)
                extra['loss_targets'] = (tout, tlen, 1.0 / self.world_size)
            lg = mod.forward_train(mem, mem_len, state, tin, U, seed=step_seed, **extra)
            if extra and getattr(mod, 'fused_loss', None) is not None:
                l_, dl = mod.fused_loss
            elif kind == 'sigmoid':
                # compute_loss_sigmoid against the feature vectors of the targets (model_helper.py:199,333-335)
                l_, dl = compute_loss_sigmoid(lg, mod.emb_bf[tout[:, :U].long()], None, tlen, TRAIN, nf=mod.nf,
                                              grad_scale=1.0 / self.world_size, want_grad=True)
            else:
                l_, dl = compute_loss(lg, tout, None, tlen, TRAIN, self.params.decoder.eos_id,
                                      grad_scale=1.0 / self.world_size, want_grad=True, vocab=mod.V)
            if kind == 'binf_projection':   # + compute_log_probs_loss(raw outputs) * reg weight (model_helper.py:327-331)
                mod.log_probs_loss(l_, float(self.params.decoder.binf_projection_reg_weight), 1.0 / self.world_size)
            loss = l_ if loss is None else loss + l_
            logits = lg if logits is None else logits
            dlogits.append(dl)
        dlogits = dlogits[0] if len(dlogits) == 1 else dlogits
        if self.ctc is not None:        # audio_loss += ctc_loss * ctc_weight (model_helper.py:347-358)
            self.ctc.forward(mem, mem_len, tout, tlen, loss, 1.0 / self.world_size)
        return loss, logits, dlogits

    def backward(self, dlogits, join=True):
        """join=False: the side streams are left running (the bottom layer's weight-gradient products); the caller joins --
        apply_gradients(joined=False) does, after it has updated the tensors that are already final."""
        self.backward_top(dlogits, layers=None, join=join)

    def backward_top(self, dlogits, layers=None, join=True):
        """Backward of the speller (+ CTC head) and of the top `layers` listener layers (None: all of them).
        Returns the number of listener layers still to do (backward_rest)."""
        g = self.vars.grads
        if isinstance(dlogits, (list, tuple)):      # --multitask: both decoders read the same memory and encoder state
            dmem, d_state = None, None
            for (mod, _), dl in zip(self.spellers, dlogits):
                dm, ds_ = mod.backward(dl, g, self.overlap)
                dmem = dm if dmem is None else dmem.add_(dm)
                if ds_ is not None:
                    ds_ = ds_ if isinstance(ds_, list) else [ds_]
                    d_state = ds_ if d_state is None else [(a[0] + b[0], a[1] + b[1]) for a, b in zip(d_state, ds_)]
        else:
            dmem, d_state = self.speller.backward(dlogits, g, self.overlap)
        if self.ctc is not None:
            self.ctc.backward(dmem, g)
        ds = None
        if d_state is not None:
            nd = 1 if self.params.encoder.unidirectional else 2
            H = self.params.encoder.num_units
            dc = torch.empty(nd, dmem.shape[0], H, dtype=torch.float32, device=dmem.device)
            dh = torch.empty_like(dc)
            per_layer = d_state if isinstance(d_state, list) else [d_state]     # decoder cell l <- encoder direction l
            pairs = []
            for l in range(nd):                      # directions without a decoder cell on top get zeros
                dcl, dhl = per_layer[l] if l < len(per_layer) else (None, None)
                pairs += [(dc[l], dcl.float() if dcl is not None else None), (dh[l], dhl.float() if dhl is not None else None)]
            hip.fill_many(copy=pairs)
            ds = (dc, dh)
        self.listener.backward_begin(dmem, ds)
        return self.backward_rest(layers, join=join)

    def backward_rest(self, layers=None, defer_last=False, join=True):
        n = self.params.encoder.num_layers if layers is None else layers
        left = self.listener.backward_layers(n, self.vars.grads, self.overlap, defer_last=defer_last)
        if left == 0 and join:
            self.overlap.join()
        return left

    def gradient_norms(self, bucket=None, acc=None):
        """grad += l2 * theta (gradient of the L2 term, model_helper.py:411-413) and per-tensor ||grad||^2; the same pass
        leaves sum theta^2 (the value of the L2 term) in vars.param_sumsq.  bucket: one of vars.buckets (default: all).
        acc=True: the accumulators are already clear (collect_status(zero_norms=True)): the pass that only adds."""
        v, p = self.vars, self.params
        lib = hip.lib()
        # (after collect_status(zero_norms=True) the accumulators are already clear: the pass that only adds)
        zeroed = self.__dict__.pop('_norms_zeroed', False)
        fn = lib.las_grad_l2_norms_acc if (zeroed if acc is None else acc) else lib.las_grad_l2_norms
        if getattr(v, 'norm_ws', None) is None:     # fixed-order sums of the workgroups' partial norms (no fp32 atomics)
            v.norm_ws = torch.zeros(lib.las_grad_l2_norms_ws_bytes(len(v.table), v.total), dtype=torch.uint8, device=v.flat.device)
        for b in (v.buckets if bucket is None else [bucket]):
            hip.check(fn(hip.addr(v.grad, b['begin']), hip.addr(v.flat, b['begin']), hip.p(b['seg']),
                         b['hi'] - b['lo'], b['end'] - b['begin'],
                         float(p.l2_reg_scale) / self.world_size, hip.addr(v.sumsq, b['lo']),
                         hip.addr(v.param_sumsq, b['slot']), hip.p(v.norm_ws), v.norm_ws.numel(), hip.stream()))

    def clip_gradients(self, bucket=None, norms=True):
        """L2 gradient + per-tensor clip_by_norm(GRAD_NORM) on the flat buffers (model_helper.py:411-416)."""
        v = self.vars
        if norms:
            self.gradient_norms(bucket)
        for b in (v.buckets if bucket is None else [bucket]):
            hip.check(hip.lib().las_grad_clip(hip.addr(v.grad, b['begin']), hip.p(b['seg']), b['hi'] - b['lo'],
                                              b['end'] - b['begin'], hip.addr(v.sumsq, b['lo']), float(GRAD_NORM), hip.stream()))

    def all_reduce_gradients(self, bucket=None, async_op=False):
        """CrossShardOptimizer's cross-replica sum (model_helper.py:405-406): RCCL all-reduce of the clipped gradients,
        the whole flat buffer or one bucket of it.  async_op: returns the work handle (RCCL's own stream; the caller
        waits on it before the Adam update), so that the rest of the backward pass runs beside the exchange."""
        if self.world_size <= 1 and self.process_group is None:
            return None
        flat = self.vars.exchange_view(bucket)
        if async_op:
            return torch.distributed.all_reduce(flat, group=self.process_group, async_op=True)
        dp.all_reduce_sum_(flat, self.process_group)
        return None

    def adam_update(self):
        """tf.train.AdamOptimizer.apply_gradients + global_step increment (model_helper.py:404,417)."""
        v, p = self.vars, self.params
        lib, st = hip.lib(), hip.stream()
        hip.check(lib.las_adam_update(hip.p(v.flat), hip.p(v.m), hip.p(v.v), hip.p(v.grad), v.total,
                                      float(p.learning_rate), 0.9, 0.999, 1e-8, 0, hip.p(self.step_dev), hip.p(v.skip_flag), st))
        hip.check(lib.las_counter_add_unless(hip.p(self.step_dev), 1, hip.p(v.skip_flag), st))   # a withheld update does not consume a step
        self._images_stale = True

    def clip_adam_update(self, bucket=None, count=True):
        """Single replica: the clip and the Adam update in one pass over the buffers (after gradient_norms).  bucket: one
        piece of the flat buffers (tail_buckets); count=False: the step counter stays (another piece of this step follows)."""
        v, p = self.vars, self.params
        lib, st = hip.lib(), hip.stream()
        if bucket is None:
            hip.check(lib.las_clip_adam_update(hip.p(v.flat), hip.p(v.m), hip.p(v.v), hip.p(v.grad), hip.p(v.seg), len(v.table),
                                               v.total, hip.p(v.sumsq), float(GRAD_NORM), float(p.learning_rate), 0.9, 0.999, 1e-8,
                                               0, hip.p(self.step_dev), hip.p(v.skip_flag), st))
        else:
            b = bucket
            hip.check(lib.las_clip_adam_update(hip.addr(v.flat, b['begin']), hip.addr(v.m, b['begin']), hip.addr(v.v, b['begin']),
                                               hip.addr(v.grad, b['begin']), hip.p(b['seg']), b['hi'] - b['lo'], b['end'] - b['begin'],
                                               hip.addr(v.sumsq, b['lo']), float(GRAD_NORM), float(p.learning_rate), 0.9, 0.999,
                                               1e-8, 0, hip.p(self.step_dev), hip.p(v.skip_flag), st))
        if count:
            hip.check(lib.las_counter_add_unless(hip.p(self.step_dev), 1, hip.p(v.skip_flag), st))   # a withheld update does not consume a step
        self._images_stale = True

    def apply_gradients(self, joined=True, update_tail=True):
        """The train op after the backward pass.  joined=False (after backward(join=False), single replica with
        tail_buckets): norms + clip + Adam of everything above the bottom listener layer run while that layer's
        weight-gradient products are still in flight on the side streams; then the join and the same for the bottom layer.
        update_tail=False leaves that last clip + Adam launch to the caller (apply_tail(): bench.py's second graph)."""
        if self.world_size == 1 and self.process_group is None:
            tb = self.tail_buckets
            ev = self.listener.before_bottom_grads
            if not joined and (tb is None or ev is None):
                self.overlap.join()
                joined = True
            self._tail_split = not (joined or tb is None)
            if not self._tail_split:
                self.collect_status(zero_norms=True)
                self.gradient_norms()
                if update_tail:
                    self.clip_adam_update()
                return
            # the bottom layer's recurrence has been enqueued on this stream; the side streams hold its products
            ev.wait()                                           # ... and, before them, every other weight gradient
            self.collect_status(zero_norms=True)
            self.__dict__.pop('_norms_zeroed', None)
            self.gradient_norms(tb[0], acc=True)
            self.clip_adam_update(tb[0], count=False)
            self.overlap.join()
            self.gradient_norms(tb[1], acc=True)
            if update_tail:
                self.clip_adam_update(tb[1])
        else:
            if not joined:
                self.overlap.join()
            self.collect_status(zero_norms=True)
            self.clip_gradients()
            self.all_reduce_gradients()
            self.adam_update()

    def apply_tail(self):
        """The launch apply_gradients(update_tail=False) left out."""
        if getattr(self, '_tail_split', False):
            self.clip_adam_update(self.tail_buckets[1])
        else:
            self.clip_adam_update()

    # -- timeouts of the persistent kernels -----------------------------------------------------------------------
    def _status_tensors(self):
        """Workspaces whose first word is a sticky timeout status: the recurrent kernels' (one per batch shape) and the
        one-launch decoder's."""
        dev = torch.cuda.current_device()
        out = [ws for (B, H, nd, d), ws in las_model.ops._WORKSPACES.items() if d == dev]
        for mod, _ in self.spellers:               # every decoder's one-launch workspaces
            out += [ws for ws in getattr(mod, '_persist_cache', {}).values()]
        return out

    def collect_status(self, zero_norms=False):
        """vars.skip_flag = 1 if any persistent kernel launched so far reported a timeout (the status words are sticky),
        else 0: enqueued once per step after the backward pass, before the gradients are exchanged.  The Adam kernels do
        nothing when the (all-reduced) flag is set, so parameters never see the invalid gradients of such a step; the
        host raises at its next check_device_status().  zero_norms: the same launch clears the per-tensor ||g||^2 and
        sum(theta^2) accumulators, and the NEXT gradient_norms() call only adds into them (no memsets of its own): for the
        flows that take all norms after this point (not the two-bucket exchange, whose first bucket is clipped earlier)."""
        ws = self._status_tensors()
        key = tuple(t.data_ptr() for t in ws)
        if getattr(self, '_status_key', None) != key:
            if torch.cuda.is_current_stream_capturing():
                raise hip.LasError('collect_status: new workspace during graph capture (run the step once eagerly first)')
            self._status_ptrs = torch.tensor(list(key) or [0], dtype=torch.int64).cuda()
            self._status_key = key
        if zero_norms:
            v = self.vars
            hip.check(hip.lib().las_train_op_begin(hip.p(self._status_ptrs), len(key), hip.p(v.skip_flag), hip.p(v.sumsq),
                                                   v.sumsq.numel(), hip.p(v.param_sumsq), v.param_sumsq.numel(), hip.stream()))
            self._norms_zeroed = True
            return
        hip.check(hip.lib().las_status_collect(hip.p(self._status_ptrs), len(key), hip.p(self.vars.skip_flag), hip.stream()))

    # -- data-parallel step with the exchange overlapped with the backward pass ----------------------------------
    def enable_exchange_overlap(self):
        """Two exchange buckets: [top listener layer, speller, CTC head] and [the lower listener layers].  The first is
        clipped and handed to RCCL while the lower layers' backward is still running (backward_and_exchange)."""
        e = self.params.encoder
        per_layer = 2 * (1 if e.unidirectional else 2)
        self.exchange_overlap = True
        lead = next(i for i, (n, _, _) in enumerate(self.vars.table) if n.startswith('listener/'))   # (a trainable binf2phone comes first)
        return self.vars.split_buckets(lead + (e.num_layers - 1) * per_layer)

    def backward_exchange_begin(self, dlogits, exchange=True):
        """First half of the overlapped step: speller backward, the top listener layer and the RECURRENCE of the next
        one (the top layer's weight-gradient products run beside it), then norms + clip + all-reduce of bucket 0.
        Returns the pending work handles.  exchange=False: only the compute part (bench.py captures it in a HIP graph and
        issues the all-reduce itself between the graphs)."""
        v = self.vars
        if len(v.buckets) < 2:
            self.backward(dlogits)
            self.collect_status()
            self.clip_gradients()
            return [self.all_reduce_gradients(async_op=True)] if exchange else []
        left = self.backward_top(dlogits, layers=1)
        if left > 0:
            # recurrence and dX of the next layer; its own weight-gradient products wait for backward_exchange_end
            left = self.backward_rest(layers=1, defer_last=True)
        if left > 0:
            self.overlap.join()                         # = the top layer's (and the speller's) weight gradients
        self.clip_gradients(v.buckets[0])
        return [self.all_reduce_gradients(v.buckets[0], async_op=True)] if exchange else []

    def backward_exchange_end(self, pending, exchange=True):
        """Second half: the remaining layers, bucket 1, then wait for both exchanges."""
        v = self.vars
        if len(v.buckets) >= 2:
            if self.listener._bwd is not None:
                self.backward_rest()
            self.collect_status()                       # the flag leaves with bucket 1 (it starts at element 0)
            self.clip_gradients(v.buckets[1])
            if exchange:
                pending = list(pending) + [self.all_reduce_gradients(v.buckets[1], async_op=True)]
        for w in pending:
            if w is not None:
                w.wait()

    def maybe_add_noise(self):
        """model_helper.py:418-432: every `add_noise` steps (and not at step 0) add N(0, noise_std) to every variable
        whose name ends in 'kernel'."""
        p = self.params
        n = int(getattr(p, 'add_noise', 0) or 0)
        if n <= 0 or self.global_step == 0 or self.global_step % n:
            return
        lib, st = hip.lib(), hip.stream()
        for i, (name, shape, _) in enumerate(self.vars.table):
            if name.endswith('kernel'):
                t = self.vars.params[name]
                hip.check(lib.las_add_noise(hip.p(t), t.numel(), float(p.noise_std),
                                            (self.noise_seed + 7919 * self.global_step) & 0x7fffffff, 1000 + i, st))
        if self.vars.mask is not None:
            self.vars.flat.mul_(self.vars.mask)        # the padded units of a zero-padded model stay at zero
        self._images_stale = True

    def l2_loss(self, from_norms=False):
        """scale * sum(theta^2) / 2 (model_helper.py:411-413).  from_norms: take sum theta^2 from the last gradient_norms()
        pass (the parameters of this step, before the update) instead of another pass over the parameters."""
        if from_norms:
            return self.vars.param_sumsq.sum(0, keepdim=True) * (0.5 * float(self.params.l2_reg_scale))
        out = torch.zeros(1, dtype=torch.float32, device='cuda')
        hip.check(hip.lib().las_sumsq(hip.p(self.vars.flat), self.vars.total, hip.p(out), hip.stream()))
        return out * (0.5 * float(self.params.l2_reg_scale))

    def total_loss(self, audio_loss, out=None):
        """audio loss + L2 term (from the sums of the last gradient_norms() pass) in one launch; out: a [1] fp32 tensor."""
        if out is None:
            out = torch.empty(1, dtype=torch.float32, device='cuda')
        v = self.vars
        hip.check(hip.lib().las_total_loss(hip.p(audio_loss), hip.p(v.param_sumsq), v.param_sumsq.numel(),
                                           0.5 * float(self.params.l2_reg_scale), hip.p(out), hip.stream()))
        return out

    def train_step(self, features, labels, num_steps=None):
        """One optimiser step; returns the loss (audio loss + L2 term) as a device scalar tensor."""
        self.vars.grad.zero_()
        audio_loss, logits, dlogits = self.forward_train(features, labels, num_steps)
        self.last_train_logits = logits          # (train_edit_distance: the TRAIN-mode metric of model_helper.py:299-317,435-439)
        if getattr(self, 'exchange_overlap', False):
            self.backward_exchange_end(self.backward_exchange_begin(dlogits))
            self.adam_update()
        else:
            tail = self.tail_buckets is not None
            self.backward(dlogits, join=not tail)
            self.apply_gradients(joined=not tail)
        loss = self.total_loss(audio_loss)
        self.maybe_add_noise()
        self.global_step += 1         # (the weight images are stale now: the next forward rebuilds them, see refresh_images)
        return loss

    def train_edit_distance(self, labels):
        """Mean normalised edit distance of the LAST train step's teacher-forced outputs against its targets: what the
        reference's LoggingTensorHook prints beside the loss every 10 iterations (model_helper.py:299-317: sample ids =
        argmax of the first decoder's logits -- phone logits, or for a lone sigmoid-output decoder its feature logits, as
        model_helper.py:251 writes it -- `utils.edit_distance(..., eos_id, mapping)`; :435-439: the last batch's mean).
        Synchronises (one argmax + a [B, U] int copy): call it at logging steps only."""
        logits = getattr(self, 'last_train_logits', None)
        if logits is None:
            return None
        width = getattr(self.speller, 'Vo', self.speller.V)
        ids = logits[..., :width].argmax(-1).to(torch.int32).cpu().numpy()
        tout = labels['targets_outputs']
        tout = (tout.cpu().numpy() if torch.is_tensor(tout) else np.asarray(tout))[:, :ids.shape[1]]
        return float(np.mean(metrics_utils.edit_distance(ids, tout, self.params.decoder.eos_id, self.params.mapping)))

    # -- inference ------------------------------------------------------------------------------
    def predict(self, features, transparent_projection=False):
        """PREDICT branch of las_model_fn (model_helper.py:253-297), greedy or beam search.  Keys follow the reference:
        a phone decoder gives 'sample_ids', 'alignment'; a binary decoder 'logits_binf', 'sample_ids_phones_binf',
        'alignment_binf' (a --multitask model both sets); 'probs' as model_helper.py:281-295 chooses it.  On top of
        those: 'logits' and 'final_sequence_length' of the decoder that defines 'probs' (EVAL uses them), and -- for a
        single binary decoder -- 'sample_ids' / 'alignment' aliases of its phone ids and alignments, so that infer.py
        works without --use_phones_from_binf.
        transparent_projection (BasicTransparentProjectionDecoder, utils/training_helper.py:156-178, binf_projection
        decoders): 'logits_binf' holds the RAW cell outputs [log p(f=1) | log p(f=0)], the phone ids come from
        transform_binf_to_phones of them (the same ids: the projection is that map) and 'probs' is the normalised
        per-feature probability p1 / (p1 + p0) (model_helper.py:287-293)."""
        if self._images_stale:
            self.refresh_images()
        if transparent_projection and not self.binf_projection:
            raise ValueError('transparent_projection needs a --binf_projection decoder: model_helper.py:247-248 applies '
                             'transform_binf_to_phones to its raw 2*binf_count-wide outputs')
        x, src_len = features['encoder_inputs'], features['source_sequence_length']
        (mem, mem_len), state = self.listener.forward(x, src_len, PREDICT)
        max_it = int(round(int(mem_len.max().item()) * self.params.decoder.decoding_length_factor))
        # model_helper.py:259-268: concat of the directions' states; a bare LSTMStateTuple as it is; anything else
        # (the per-layer tuples of the stacked listener) leaves 'embedding' out of the predictions
        emb = None
        if hasattr(state, 'c') and hasattr(state, 'h') and torch.is_tensor(state.c):
            emb = torch.stack([state.c, state.h], 1)
        elif all(hasattr(s, 'c') and torch.is_tensor(s.c) for s in state):
            emb = torch.stack([torch.cat([s.c for s in state], 1), torch.cat([s.h for s in state], 1)], 1)
        enc_out = mem
        if self.padded and self.params.encoder.num_units != self.hparams.encoder.num_units:
            cols = self.vars.index_maps[speller_plan(self.params.decoder)[0][0] + '/memory_layer/kernel'][0]
            enc_out = mem[..., torch.as_tensor(cols, device=mem.device)]        # the logical units of the zero-padded listener
        out = {'encoder_out': enc_out, 'source_length': mem_len}
        if emb is not None:
            out['embedding'] = emb
        beam_width = int(getattr(self.params.decoder, 'beam_width', 0) or 0)
        if beam_width > 0:              # model_helper.py:231-236: predicted_ids [B,T,K] instead of logits
            for mod, kind in self.spellers:
                bs = self._beam_speller() if mod is self.speller and kind == 'phones' else mod
                ids, lens, lps = bs.forward_beam(mem, mem_len, state, max_it, beam_width,
                                                 partial_targets=features.get('partial_targets'))
                key = 'sample_ids' if kind == 'phones' else 'sample_ids_phones_binf'
                out[key] = ids
                if key not in ('sample_ids',) and 'sample_ids' not in out and len(self.spellers) == 1:
                    out['sample_ids'] = ids
                if 'beam_lengths' not in out:
                    out['beam_lengths'], out['beam_log_probs'] = lens, lps
            return out
        primary = None
        out['_decoders'] = []                # (kind, logits, final_sequence_length) per decoder, for evaluate()
        for mod, kind in self.spellers:
            logits, ids, final_len, align = mod.forward_greedy(mem, mem_len, state, max_it)
            out['_decoders'].append((kind, logits, final_len))
            if kind == 'phones':
                out['sample_ids'], out['alignment'] = ids, align
                out['probs'] = torch.softmax(logits, -1)                             # model_helper.py:284-285
                primary = (logits, final_len)
                continue
            out['alignment_binf'] = align
            if kind == 'sigmoid':            # feature logits; ids = argmax over them as model_helper.py:251 writes it
                out['logits_binf'] = logits
                out['sample_ids_phones_binf'] = logits.argmax(-1).to(torch.int32)
                out['sample_features_binf'] = ids                                    # the decoded 0/1 vectors [B,S,nf]
                probs = torch.sigmoid(logits)                                        # model_helper.py:282-283
            elif transparent_projection:
                raw = mod.last_raw_outputs.float()                                   # [B,S,2 nf]
                out['logits_binf'] = raw
                out['sample_ids_phones_binf'] = ids                                  # argmax(transform_binf_to_phones(raw))
                e = torch.exp(raw - raw.max(-1, keepdim=True).values)
                nf = raw.shape[-1] // 2
                probs = e[..., :nf] / (e[..., :nf] + e[..., nf:2 * nf])              # model_helper.py:287-293
            else:
                out['logits_binf'] = logits
                out['sample_ids_phones_binf'] = ids
                probs = torch.softmax(logits, -1)                                    # model_helper.py:294-295
            if primary is None:
                out['probs'] = probs
                primary = (logits, final_len)
            if len(self.spellers) == 1:
                out['sample_ids'], out['alignment'] = out['sample_ids_phones_binf'], align
        out['logits'], out['final_sequence_length'] = primary
        return out

    def evaluate(self, features, labels):
        """EVAL branch: free-running greedy decode, the padded losses of every decoder (model_helper.py:54-76,98-130,
        319-342) and the edit distance of the first decoder's phone ids (:299-309)."""
        pred = self.predict(features)
        tout, tlen = labels['targets_outputs'], labels['target_sequence_length']
        eos = self.params.decoder.eos_id
        loss = None
        for (mod, kind), (_, logits, final_len) in zip(self.spellers, pred['_decoders']):
            if kind == 'sigmoid':
                if mod.feat is None:
                    raise ValueError('evaluating the sigmoid-output decoder needs the binf2phone map (target feature vectors)')
                # (the reference passes the integer targets here, model_helper.py:336-337, which its reshape to
                # [-1, binf_count] cannot take; the evident intent -- the targets' feature vectors, as in TRAIN -- is used)
                l_, _ = compute_loss_sigmoid(logits.contiguous(), mod.emb_bf[tout.long()], final_len, tlen, EVAL, nf=mod.nf)
            else:
                l_, _ = compute_loss(logits.contiguous(), tout, final_len, tlen, EVAL, eos)
            loss = l_ if loss is None else loss + l_
        ed = metrics_utils.edit_distance(pred['sample_ids'], tout, eos, self.params.mapping)
        return loss, ed, pred

    def check_device_status(self):
        """The persistent kernels (recurrent layers, one-launch decoder) bound every inter-workgroup wait and flag a
        timeout in their workspace instead of hanging; results are then invalid.  The status words are sticky (no launch
        clears them), so this reports ANY launch since the last check.  Call at logging points, before a checkpoint
        is written and at the end of inference (it synchronises): raises LasError and clears the words."""
        try:
            las_model.ops.check_all_lstm_status()
            for mod, _ in self.spellers:
                for ws in getattr(mod, '_persist_cache', {}).values():
                    st = int(ws[:4].view(torch.int32).item())
                    if st:
                        if st & 32:          # the four-parts sequential backward: one workgroup per utterance from here on
                            from .las.speller_general import GeneralSpeller
                            if GeneralSpeller.SEQ_FOUR_PARTS:
                                GeneralSpeller.SEQ_FOUR_PARTS = False
                                import sys
                                print('phones_las_amd: the four-parts backward decoder timed out; it runs one workgroup per utterance '
                                      'for the rest of this process (LAS_DEC_SEQ_PARTS=1 has the same effect)', file=sys.stderr)
                        raise hip.LasError('persistent decoder reported a barrier timeout (status %d)' % st)
        except hip.LasError:
            for ws in self._status_tensors():        # read: the sticky words start over
                ws[:4].zero_()
            raise

    def read_and_clear_status(self):
        """Non-raising form of check_device_status(): synchronises, returns the non-zero status words of the persistent
        kernels' workspaces (empty list: no timeout since the last read) and clears them.  bench.py calls it after every
        candidate of its untimed probe, so that a form that timed out there is dropped instead of poisoning the timed run
        through the sticky words."""
        torch.cuda.synchronize()
        bad = []
        for ws in self._status_tensors():
            st = int(ws[:4].view(torch.int32).item())
            if st:
                bad.append(st)
                ws[:4].zero_()
        return bad

    def _beam_speller(self):
        """Beam search runs on the general cell stack (it gathers the decoder state between steps); the fused
        single-cell speller gets a GeneralSpeller twin over the same variables."""
        from .las.speller_general import GeneralSpeller
        if isinstance(self.speller, GeneralSpeller):
            return self.speller
        if getattr(self, '_beam_twin', None) is None:
            self._beam_twin = GeneralSpeller(self.params.decoder, self.vars.params, _enc_depth(self.params.encoder),
                                             las_model._ATT[self.params.decoder.attention_type])
        else:
            self._beam_twin.refresh(self.vars.params)
        return self._beam_twin


def las_model_fn(features, labels, mode, config, params, binf2phone=None, run_name=None,
                 transparent_projection=False, *, model=None):
    """model_helper.py:165-444.  ``model`` is the LasModel that holds the variables (an Estimator would own it);
    when omitted a freshly initialised one is built.  Returns an EstimatorSpec whose ``train_op`` is a callable
    that applies one optimiser step (TF returns a graph op)."""
    if model is None:
        model = LasModel(params, binf2phone=binf2phone)
    if mode == PREDICT:
        return EstimatorSpec(mode, predictions=model.predict(features, transparent_projection=transparent_projection))
    if mode == EVAL:
        loss, ed, pred = model.evaluate(features, labels)
        return EstimatorSpec(mode, loss=loss, predictions=pred,
                             eval_metric_ops={'edit_distance': float(np.mean(ed))})
    model.vars.grad.zero_()
    audio_loss, logits, dlogits = model.forward_train(features, labels)
    loss = audio_loss + model.l2_loss()
    sample_ids = logits[..., :getattr(model.speller, 'Vo', model.speller.V)].argmax(-1).to(torch.int32)

    def train_op():
        model.backward(dlogits)
        model.apply_gradients()
        model.maybe_add_noise()
        model.refresh_images()
        model.global_step += 1

    return EstimatorSpec(mode, loss=loss, train_op=train_op, predictions={'sample_ids': sample_ids, 'logits': logits})
