"""Trains the ORACLE (oracle/las_oracle.py: exact f64 model and its bf16 storage model) on the toy corpus of
tests/twin_corpus.py for STEPS optimiser steps and writes the loss curve checkpoints and the final greedy PER on the held-out
utterances to tests/golden/convergence_twin.json -- the fixture tests/test_gpu_convergence.py holds the device against.

    TWIN_OUT=part.json python tests/golden/make_convergence_twin.py <f64|bf16> <init seed> [...]    (build container, CPU; ~5 min per run)
    python tests/golden/make_convergence_twin.py merge part*.json                                 -> tests/golden/convergence_twin.json

The only available stand-in for north_star's "matched TIMIT PER": the reference's TF-1 graph cannot run here and no corpus
ships with it, so the question the twin answers is whether the device's bf16 operand storage (3e-2 worst-case gradient error
against the exact model) changes WHERE training ends up."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import las_oracle as O          # noqa: E402
from tests import twin_corpus as TC         # noqa: E402

OUT = os.environ.get('TWIN_OUT') or os.path.join(HERE, 'convergence_twin.json')


def oracle_hp():
    m = TC.MODEL
    return O.HP(encoder=O.EncoderHP(num_layers=m['L'], num_units=m['H']), num_channels=m['F'],
                decoder=O.DecoderHP(num_layers=1, num_units=m['Hd'], target_vocab_size=m['V'], attention_type=m['att'],
                                    bottom_only=True, pass_hidden_state=True),
                learning_rate=m['lr'], l2_reg_scale=m['l2'])


def to_torch(b):
    return {k: torch.tensor(v.astype(np.float64) if k == 'encoder_inputs' else v.astype(np.int64)) for k, v in b.items()}


def run(mxu, seed):
    hp = oracle_hp()
    p = O.init_params(hp, seed=seed)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(x) for k, x in p.items()}
    batches = [to_torch(b) for b in TC.train_batches()]
    curve, losses, pers = {}, [], {}
    tb, refs = TC.test_batches()
    tb = [to_torch(b) for b in tb]

    def held_out_per():
        hyps = []
        with torch.no_grad():
            for b in tb:
                (mem, ml), st = O.listener(b['encoder_inputs'], b['source_sequence_length'], p, hp.encoder, mxu)
                _, ids, _, _ = O.speller_greedy(hp, p, mem, ml, st, mxu)
                hyps += [row.tolist() for row in ids]
        return TC.per(hyps, refs)
    t0 = time.time()
    for t in range(TC.STEPS):
        out = O.train_step(hp, p, None, None, t + 1, batches[t % len(batches)], mxu=mxu)
        losses.append(float(out['loss']))
        if t + 1 in TC.CHECKPOINTS:
            curve[t + 1] = float(np.mean(losses[-TC.WINDOW:]))
            print(mxu, t + 1, curve[t + 1], '%.0f s' % (time.time() - t0), flush=True)
        p, m, v = O.adam_apply(p, m, v, out['clipped'], t + 1, hp.learning_rate)
        if t + 1 in TC.PER_STEPS:
            pers[t + 1] = held_out_per()
            print(mxu, 'PER at', t + 1, pers[t + 1], flush=True)
    return dict(curve={str(k): v for k, v in curve.items()}, per_at={str(k): v for k, v in pers.items()},
                per=float(np.median(list(pers.values()))), seconds=round(time.time() - t0))


if __name__ == '__main__':
    # usage: make_convergence_twin.py <f64|bf16> <seed> [<seed> ...]   (one json per invocation when TWIN_OUT is set; merge below)
    if sys.argv[1] == 'merge':
        res = {'corpus': dict(n_train=TC.N_TRAIN, n_test=TC.N_TEST, batch=TC.BATCH, steps=TC.STEPS, noise=TC.NOISE, window=TC.WINDOW,
                              per_steps=TC.PER_STEPS, model=TC.MODEL), 'f64': {}, 'bf16': {}}
        for path in sys.argv[2:]:
            part = json.load(open(path))
            for mxu in ('f64', 'bf16'):
                res[mxu].update(part.get(mxu, {}))
        json.dump(res, open(OUT, 'w'), indent=1)
        print({m: {s: round(v['per'], 2) for s, v in res[m].items()} for m in ('f64', 'bf16')})
        sys.exit(0)
    mxu, seeds = sys.argv[1], [int(a) for a in sys.argv[2:]]
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    res.setdefault(mxu, {})
    for seed in seeds:
        res[mxu][str(seed)] = run(mxu, seed)
        json.dump(res, open(OUT, 'w'), indent=1)
