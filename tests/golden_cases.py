"""The benchmarked / BASELINE configurations as parity cases (TEST INFRASTRUCTURE; imports neither the oracle nor
/root/reference).  tests/golden/make_golden.py (build container, has the oracle) and tests/test_gpu_golden_shapes.py
(GPU box, no oracle in the loop) build IDENTICAL hyper-parameters, weights and batches from this module; only the
oracle's expected outputs travel as fixtures (tests/golden/shape_<case>.npz).

Cases (BASELINE.json configs / SURVEY.md 8(d)):
  metricM_dense   cfg2 = the headline config: 3 x pBiLSTM-256 + Luong + 1x256 decoder, F=40, V=64, T=800, U=80, B=4
  metricM_ragged  the same model, B=8, SURVEY 8(d)'s ragged lengths (800-8*(i%26), 80-(i%17))
  metricM_ragged64  the same ragged lengths at bench.py's `metric-M-ragged` shape, B=64 (round 4)
  metricM_bench   the same model at bench.py's exact shape: B=64 dense (16 slices of 4 utterances per direction, 8 decoder
                  groups); the fixture keeps logits, losses, states and gradients, not the encoder memory
  metricL_ctc     cfg3/cfg4: 4 x pBiLSTM-512 + Bahdanau + CTC head (ctc_weight 0.3), F=80, T=384 (T'=48 >= U for CTC), U=24, B=4, ragged
  cfg5_binf       cfg5: --binary_outputs --binf_projection with the reference's misc/binf_map.csv (nf=40, V=197) +
                  bahdanau_monotonic (sigmoid_noise 1 in TRAIN: the score noise comes from the device's counter-based
                  generator, restated in numpy below), 3 x pBiLSTM-256, F=39, T=64, U=12, B=4
  cfg1_timit      cfg1: 2 x pBiLSTM-128 + Luong, F=39, V=64, B=4, T=300 (TIMIT's ~3 s), U=40, ragged
  cfg5_full       cfg5 at T=800 / U=80 (T'=200), B=4, ragged
  metricL_full    cfg3/cfg4 at T=800 / U=80, B=16 ragged, CTC head
  dec512_groups   (and _luong) 512-unit decoders on M=2048, T'=100, U=80, B=16 ragged: two decoder groups
"""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden')

CASES = {
    'metricM_dense': dict(F=40, L=3, H=256, Hd=256, V=64, att='luong', T=800, U=80, B=4, ragged='dense'),
    'metricM_ragged': dict(F=40, L=3, H=256, Hd=256, V=64, att='luong', T=800, U=80, B=8, ragged='survey'),
    'metricM_bench': dict(F=40, L=3, H=256, Hd=256, V=64, att='luong', T=800, U=80, B=64, ragged='dense', memory=False),
    # round 4: bench.py's `metric-M-ragged` at its exact shape (SURVEY 8(d)'s ragged lengths over all 64 utterances: every
    # slice of four utterances mixes lengths, the lean steps end at the slice's shortest and the masked general step takes over)
    'metricM_ragged64': dict(F=40, L=3, H=256, Hd=256, V=64, att='luong', T=800, U=80, B=64, ragged='survey', memory=False),
    'metricL_ctc': dict(F=80, L=4, H=512, Hd=512, V=64, att='bahdanau', T=384, U=24, B=4, ragged='mixed', ctc=0.3),
    'cfg5_binf': dict(F=39, L=3, H=256, Hd=256, V=197, att='bahdanau_monotonic', T=64, U=12, B=4, ragged='mixed',
                      binf='binf_map.csv', binf_reg=1.0),
    'cfg1_timit': dict(F=39, L=2, H=128, Hd=128, V=64, att='luong', T=300, U=40, B=4, ragged='mixed'),
    # round 3: cfg5 AT ITS STATED SIZE (T=800 -> T'=200 frames, U=80: the monotonic 'parallel' chain runs past the point
    # where the exclusive cumprod drops below its 1e-10 clip, which the T=64 case never reaches)
    'cfg5_full': dict(F=39, L=3, H=256, Hd=256, V=197, att='bahdanau_monotonic', T=800, U=80, B=4, ragged='mixed',
                      binf='binf_map.csv', binf_reg=1.0, memory=False),
    # round 3: the 512-unit one-launch decoders of metric-L / cfg3 / cfg4 over SEVERAL groups at the benchmarked decoder
    # shape (M=2048, T'=100, U=80; B=16 = two groups of 8); the listener is kept short (2 layers, T=200) so that the oracle
    # finishes in minutes
    'dec512_groups': dict(F=40, L=2, H=512, Hd=512, V=64, att='bahdanau', T=200, U=80, B=16, ragged='mixed', memory=False),
    # round 3: cfg3 / cfg4 at their stated size: 4 x pBiLSTM-512 over T=800 (T'=100), Bahdanau, 1x512 decoder, CTC head,
    # B=16 ragged (lstm_fwd/bwd_kernel<512> over 800-step chains, two decoder groups)
    'metricL_full': dict(F=80, L=4, H=512, Hd=512, V=64, att='bahdanau', T=800, U=80, B=16, ragged='mixed', ctc=0.3,
                         memory=False),
    'dec512_groups_luong': dict(F=40, L=2, H=512, Hd=512, V=64, att='luong', T=200, U=80, B=16, ragged='mixed', memory=False),
}
SEED_PARAMS, SEED_BATCH = 4321, 1234


def binf_matrix(name):
    """The [nf, V] 0/1 matrix the REFERENCE's utils.ipa_utils.load_binf2phone returned for misc/<name> (fixture
    tests/golden/binf_maps.json, produced by make_golden.py from the reference's own code)."""
    m = json.load(open(os.path.join(GOLDEN, 'binf_maps.json')))['maps'][name]
    return np.array([[int(ch) for ch in row] for row in m['rows']], dtype=np.float32)


def product_params(case):
    """params for LasModel (phones-las_amd/utils/params_utils.py), the reference's "LAS architecture" flags."""
    from phones_las_amd.utils import params_utils as pu
    c = CASES[case]
    hp = pu.get_default_hparams()
    kv = dict(num_channels=c['F'], encoder_layers=c['L'], encoder_units=c['H'], use_pyramidal=True, unidirectional=False,
              decoder_layers=1, decoder_units=c['Hd'], target_vocab_size=c['V'], attention_type=c['att'], bottom_only=True,
              pass_hidden_state=True, dropout=0.0, sampling_probability=0.0, learning_rate=1e-3, l2_reg_scale=1e-6,
              ctc_weight=c.get('ctc', -1.0))
    if c.get('binf'):
        nf = binf_matrix(c['binf']).shape[0]
        kv.update(binary_outputs=True, binf_projection=True, binf_count=nf, binf_projection_reg_weight=c['binf_reg'])
    for k, v in kv.items():
        hp.set_hparam(k, v)
    return pu.get_encoder_decoder_hparams(hp)


def weights(case):
    """name -> float32 array: the product's own initialisers (model_helper._init_array, seed 4321: U(-0.075, 0.075) LSTM /
    projection kernels, glorot Dense kernels) with the zero biases replaced by U(-0.1, 0.1) draws and the monotonic
    score bias at 0.3, so that every bias path carries signal."""
    from phones_las_amd import model_helper as mh
    rng = np.random.default_rng(SEED_PARAMS)
    rb = np.random.default_rng(SEED_PARAMS + 1)
    out = {}
    for name, shape, init in mh.param_table(product_params(case)):
        a = mh._init_array(shape, init, rng).astype(np.float32)
        if init == 'zeros':
            a = rb.uniform(-0.1, 0.1, size=shape).astype(np.float32)
            if name.endswith('attention_score_bias'):
                a = np.full(shape, 0.3, dtype=np.float32)
        out[name] = a
    return out


def batch(case):
    """SURVEY.md 8(d) synthetic batch (numpy): x ~ N(0,1), frames beyond the length zeroed, tokens in [3, V)."""
    c = CASES[case]
    B, T, F, V, U = c['B'], c['T'], c['F'], c['V'], c['U']
    rng = np.random.default_rng(SEED_BATCH)
    x = rng.standard_normal((B, T, F)).astype(np.float32)
    if c['ragged'] == 'dense':
        src = np.full(B, T)
        tgt = np.full(B, U)
    elif c['ragged'] == 'survey':
        src = np.array([T - 8 * (i % 26) for i in range(B)])
        tgt = np.array([U - (i % 17) for i in range(B)])
    else:       # a full-length row, a short one, odd lengths in between
        src = np.array([T, max(1, T // 5), T - 3, (2 * T) // 3 + 1][:B] + [T - 7 * i for i in range(4, B)])
        tgt = np.array([U, max(1, U // 4), U - 1, U // 2 + 1][:B] + [U - i for i in range(4, B)])
    tin = np.full((B, U), 2, dtype=np.int64)
    tout = np.full((B, U), 2, dtype=np.int64)
    for b in range(B):
        x[b, src[b]:] = 0.0
        n = int(tgt[b]) - 1
        y = rng.integers(3, V, size=n)
        tin[b, 0] = 1
        tin[b, 1:n + 1] = y
        tout[b, :n] = y
    return {'encoder_inputs': x, 'source_sequence_length': src.astype(np.int64), 'targets_inputs': tin,
            'targets_outputs': tout, 'target_sequence_length': tgt.astype(np.int64)}


# ---------------------------------------------------------------------------------------------------------------------
# numpy restatement of the device's counter-based generator (phones-las_amd/csrc/las_common.h: las_mix32, las_uniform,
# las_normal).  The uniforms are bit-exact; the normals differ from the device by its fast log / cos (~1e-6).
# ---------------------------------------------------------------------------------------------------------------------
def _mix32(x):
    x = x.astype(np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def device_uniform(seed, stream, idx):
    idx = np.asarray(idx, dtype=np.uint64)
    h = _mix32(np.array([(seed ^ ((stream * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFFFFFF], dtype=np.uint64))
    h = _mix32(h ^ (idx & 0xFFFFFFFF))
    h = _mix32((h + ((idx >> 32) * 0x85EBCA6B & 0xFFFFFFFF) + 0x632BE5AB) & 0xFFFFFFFF)
    return ((h >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def device_normal(seed, stream, n):
    i = np.arange(n, dtype=np.uint64)
    u1 = np.maximum(device_uniform(seed, stream, 2 * i), np.float32(1e-12))
    u2 = device_uniform(seed, stream, 2 * i + 1)
    return (np.sqrt(-2.0 * np.log(u1.astype(np.float64))) * np.cos(6.2831853 * u2.astype(np.float64))).astype(np.float32)


NOISE_STREAM = 3        # speller_general.GeneralSpeller.NOISE_STREAM: draw (t*B + b)*Tm + t'


def first_step_seed(seed=SEED_PARAMS):
    """LasModel's seed of the stochastic draws at global_step 0 (model_helper.LasModel.__init__ / forward_train)."""
    return (seed * 2654435761 + 12345) & 0x7fffffff


def monotonic_noise(case):
    """[U, B, T'] score noise the device draws for bahdanau_monotonic in TRAIN mode at the first optimiser step."""
    c = CASES[case]
    Tm = int(math.ceil(c['T'] / 2 ** (c['L'] - 1)))
    U = int(batch(case)['target_sequence_length'].max())
    return device_normal(first_step_seed(), NOISE_STREAM, U * c['B'] * Tm).reshape(U, c['B'], Tm)


def grad_sample(name, n):
    """Indices of the gradient elements a fixture keeps for tensor ``name`` with n elements (at most 4096, evenly spread)."""
    stride = max(1, -(-n // 4096))
    return np.arange(0, n, stride)
