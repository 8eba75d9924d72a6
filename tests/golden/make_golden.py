#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/ (run in the build container only; needs /root/reference).

  oracle_small.npz   seeded inputs + expected loss/logits/gradient norms of the oracle (f64 and bf16 storage
                     models) for a small LAS model.  The reference itself cannot run here (TF 1.15), so these pin
                     the ORACLE; they are what the GPU parity tests compare against on the GPU box.
  shape_<case>.npz   expected outputs of the oracle at the BENCHMARKED / BASELINE shapes (tests/golden_cases.py: metric-M
                     dims with T=800 / U=80, metric-L dims + CTC, cfg5 with the reference's real binf_map.csv, cfg1):
                     loss, logits, encoder memory, final encoder states, per-tensor gradient norms and a sample of the
                     gradient elements, for the exact float64 model ('f64') and for the device's storage model ('bf16':
                     bf16 GEMM operands forward AND backward).  The GPU tests compare the HIP path with these without the
                     oracle in the loop.  `python tests/golden/make_golden.py shapes [case ...]` regenerates them.
  binf_maps.json     outputs of the REFERENCE's own utils/ipa_utils.load_binf2phone / get_mapping (imported from
                     /root/reference under a stub `tensorflow` module and a dummy `espeak-ng`) on the
                     reference's misc/ data files, plus the small input CSV so the product loader can be checked
                     without /root/reference.
"""
import json
import os
import stat
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'


def oracle_small():
    from oracle import las_oracle as O
    from tests.helpers import make_hparams, make_batch
    out = {}
    for att in ('luong', 'bahdanau'):
        ohp, _ = make_hparams(att=att)
        op = O.init_params(ohp, bias_scale=0.1)
        batch = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
        for mxu in ('f64', 'bf16'):
            r = O.train_step(ohp, op, None, None, 1, batch, mxu=mxu)
            key = '%s_%s' % (att, mxu)
            out[key + '_loss'] = np.array(float(r['loss']))
            out[key + '_ce'] = np.array(float(r['aux']['ce'].detach()))
            out[key + '_logits'] = r['aux']['logits'].detach().numpy().astype(np.float32)
            out[key + '_gradnorm'] = np.array([float(r['grads'][n].norm()) for n in op])
            out[key + '_memory'] = r['aux']['memory'].detach().numpy().astype(np.float32)
    out['x'] = batch['encoder_inputs'].numpy().astype(np.float32)
    out['targets_inputs'] = batch['targets_inputs'].numpy()
    out['targets_outputs'] = batch['targets_outputs'].numpy()
    np.savez_compressed(os.path.join(HERE, 'oracle_small.npz'), **out)


def reference_binf():
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    sys.dont_write_bytecode = True
    tmp = tempfile.mkdtemp()
    exe = os.path.join(tmp, 'espeak-ng')
    with open(exe, 'w') as f:
        f.write('#!/bin/sh\nexit 0\n')
    os.chmod(exe, os.stat(exe).st_mode | stat.S_IEXEC)
    os.environ['PATH'] = tmp + os.pathsep + os.environ['PATH']
    np.int = int
    pkg = types.ModuleType('utils')
    pkg.__path__ = [os.path.join(REF, 'utils')]
    sys.modules['utils'] = pkg
    sys.modules['tensorflow'] = types.ModuleType('tensorflow')
    import importlib
    importlib.import_module('utils.vocab_utils')
    ipa = importlib.import_module('utils.ipa_utils')
    res = {'maps': {}}
    for name in sorted(os.listdir(os.path.join(REF, 'misc'))):
        if not name.startswith('binf_map') or not name.endswith('.csv'):
            continue
        df = ipa.load_binf2phone(os.path.join(REF, 'misc', name))
        vals = df.values.astype(int)
        res['maps'][name] = {
            'shape': list(vals.shape), 'index': [str(i) for i in df.index], 'columns': [str(c) for c in df.columns],
            'rows': [''.join(str(int(v)) for v in row) for row in vals],
        }
    res['input_csv_binf_map_arpabet'] = open(os.path.join(REF, 'misc', 'binf_map_arpabet.csv')).read()
    nv, im = ipa.get_mapping(os.path.join(REF, 'misc', 'phones.60-48-39.map'), os.path.join(REF, 'misc', 'timit-61.txt'))
    res['timit_mapping'] = {'new_vocab': nv, 'int_mapping': im,
                            'input_map': open(os.path.join(REF, 'misc', 'phones.60-48-39.map')).read(),
                            'input_vocab': open(os.path.join(REF, 'misc', 'timit-61.txt')).read()}
    with open(os.path.join(HERE, 'binf_maps.json'), 'w') as f:
        json.dump(res, f)
    for m in ('utils', 'tensorflow', 'utils.vocab_utils', 'utils.ipa_utils'):
        sys.modules.pop(m, None)


def oracle_hp(case):
    from oracle import las_oracle as O
    from tests import golden_cases as G
    c = G.CASES[case]
    binf = G.binf_matrix(c['binf']) if c.get('binf') else None
    return O.HP(encoder=O.EncoderHP(num_layers=c['L'], num_units=c['H']), num_channels=c['F'],
                decoder=O.DecoderHP(num_layers=1, num_units=c['Hd'], target_vocab_size=c['V'], attention_type=c['att'],
                                    bottom_only=True, pass_hidden_state=True, binf_projection=binf is not None,
                                    binf_count=0 if binf is None else int(binf.shape[0]), binf_map=binf,
                                    binf_projection_reg_weight=c.get('binf_reg', 1.0)),
                learning_rate=1e-3, l2_reg_scale=1e-6, ctc_weight=c.get('ctc', -1.0))


def shape_case(case):
    """One fixture: the oracle's train step (loss + autograd gradients, before the clip) on tests/golden_cases' inputs."""
    import time
    from oracle import las_oracle as O
    from phones_las_amd import model_helper as mh
    from tests import golden_cases as G
    c = G.CASES[case]
    ohp = oracle_hp(case)
    w = G.weights(case)
    assert [n for n, _, _ in O.param_table(ohp)] == list(w) == [n for n, _, _ in mh.param_table(G.product_params(case))]
    op = {k: torch.tensor(v.astype(np.float64)) for k, v in w.items()}
    nb = G.batch(case)
    batch = {k: torch.tensor(v.astype(np.float64) if k == 'encoder_inputs' else v) for k, v in nb.items()}
    stochastic = None
    if c['att'] == 'bahdanau_monotonic':
        stochastic = {'att_noise': torch.tensor(G.monotonic_noise(case).astype(np.float64))}
    out = {}
    for mxu in ('f64', 'bf16'):
        t0 = time.time()
        r = O.train_step(ohp, op, None, None, 1, batch, mxu=mxu, stochastic=stochastic)
        aux = r['aux']
        out[mxu + '_audio_loss'] = np.array(float(r['audio_loss']))
        out[mxu + '_loss'] = np.array(float(r['loss']))
        out[mxu + '_ce'] = np.array(float(aux['ce'].detach()))
        out[mxu + '_logits'] = aux['logits'].detach().numpy().astype(np.float32)
        if c.get('memory', True):
            out[mxu + '_memory'] = aux['memory'].detach()[:, ::4].numpy().astype(np.float16)      # every 4th frame
        st = aux['state']
        out[mxu + '_state_c'] = np.stack([s[0].detach().numpy() for s in st]).astype(np.float32)
        out[mxu + '_state_h'] = np.stack([s[1].detach().numpy() for s in st]).astype(np.float32)
        # gradients of the audio loss (the L2 term's l2*theta is removed: the device adds it in its norms pass)
        names = list(op)
        g = {n: (r['grads'][n] - ohp.l2_reg_scale * op[n]).numpy() for n in names}
        out[mxu + '_gradnorm'] = np.array([float(np.linalg.norm(g[n])) for n in names])
        for i, n in enumerate(names):
            flat = g[n].reshape(-1)
            out['%s_grad_%02d' % (mxu, i)] = flat[G.grad_sample(n, flat.size)].astype(np.float32)
        if 'ctc' in aux:
            out[mxu + '_ctc'] = np.array(float(aux['ctc'].detach()))
        if 'log_probs_loss' in aux:
            out[mxu + '_log_probs_loss'] = np.array(float(aux['log_probs_loss'].detach()))
        print('  %s %s: %.1f s, audio loss %.6f' % (case, mxu, time.time() - t0, float(r['audio_loss'])), flush=True)
    out['names'] = np.array(list(op))
    out['x_checksum'] = np.array(float(np.abs(nb['encoder_inputs']).sum()))
    np.savez_compressed(os.path.join(HERE, 'shape_%s.npz' % case), **out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'shapes':
        from tests import golden_cases as G
        for case in (sys.argv[2:] or list(G.CASES)):
            shape_case(case)
    else:
        reference_binf()
        oracle_small()
    print('golden fixtures written to', HERE)
