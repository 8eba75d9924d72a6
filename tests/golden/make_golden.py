#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/ (run in the build container only; needs /root/reference).

  oracle_small.npz   seeded inputs + expected loss/logits/gradient norms of the oracle (f64 and bf16 storage
                     models) for a small LAS model.  The reference itself cannot run here (TF 1.15), so these pin
                     the ORACLE; they are what the GPU parity tests compare against on the GPU box.
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


if __name__ == '__main__':
    reference_binf()
    oracle_small()
    print('golden fixtures written to', HERE)
