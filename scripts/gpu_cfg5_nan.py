"""Where does a non-finite value first appear in a cfg5-class train run (binf_projection + bahdanau_monotonic)?
Runs STEPS eager train steps of `bench.py --config CFG` (default cfg5) at batch B (default: the config's) and reports
per step: loss, the decoder's saved tensors that hold non-finite values, the gradients that do, and the largest
magnitudes of the raw [lp1 | lp0] outputs.  Env: CFG, STEPS, B, T, U, LR, SEED."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from phones_las_amd import model_helper as mh  # noqa: E402


def bad(t):
    return int((~torch.isfinite(t.float())).sum().item())


def main():
    c = dict(bench.CONFIGS[os.environ.get('CFG', 'cfg5')])
    for k in ('B', 'T', 'U'):
        if os.environ.get(k):
            c[k] = int(os.environ[k])
    steps = int(os.environ.get('STEPS', '40'))
    lr = float(os.environ.get('LR', '1e-3'))
    dev = torch.device('cuda', 0)
    model = mh.LasModel(bench.build_params(c, lr=lr), binf2phone=bench.binf_matrix(c['binf']) if c.get('binf') else None)
    feats, labels = bench.synthetic_batch(c, int(os.environ.get('SEED', '1234')), dev)
    feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])
    trace = []

    def hook(t, d):
        torch.cuda.synchronize()
        row = {k: (bad(v), float(v.float().abs().nan_to_num(0, 0, 0).max())) for k, v in d.items() if v is not None}
        trace.append((t, row))

    if os.environ.get('HOOK', '1') != '0' and hasattr(model.speller, 'debug_hook'):
        model.speller.debug_hook = hook
    for step in range(steps):
        del trace[:]
        model.vars.grad.zero_()
        audio, logits, dlogits = model.forward_train(feats, labels, num_steps=c['U'])
        torch.cuda.synchronize()
        sv = getattr(model.speller, 'saved', None)
        rep = []
        if isinstance(sv, dict):
            for k, v in sv.items():
                vs = v if isinstance(v, (list, tuple)) else [v]
                for i, t in enumerate(vs):
                    if torch.is_tensor(t) and t.is_floating_point() and bad(t):
                        rep.append('%s[%d]:%d' % (k, i, bad(t)))
            att = sv.get('att')
            amax = float(att.float().abs().max()) if torch.is_tensor(att) else float('nan')
            al = sv.get('align')
            almax = float(al.abs().max()) if torch.is_tensor(al) else float('nan')
            alsum = al.sum(-1) if torch.is_tensor(al) else None
        else:
            amax = almax = float('nan')
            alsum = None
        model.backward(dlogits)
        torch.cuda.synchronize()
        first = next(((t, r) for t, r in trace if any(b for b, _ in r.values())), None)
        if trace:
            print('  bwd max-abs at t=U-1:', {k: '%.3g' % m for k, (b, m) in trace[0][1].items()})
            print('  bwd max-abs at t=0  :', {k: '%.3g' % m for k, (b, m) in trace[-1][1].items()})
            big = max(trace, key=lambda x: max(m for _, m in x[1].values()))
            print('  largest step t=%d:' % big[0], {k: '%.3g' % m for k, (b, m) in big[1].items()})
        if first is not None:
            print('  FIRST non-finite at decoder step t=%d:' % first[0], {k: b for k, (b, m) in first[1].items() if b},
                  ' max-abs:', {k: '%.3g' % m for k, (b, m) in first[1].items()})
            prev = [r for t, r in trace if t == first[0] + 1]
            if prev:
                print('  step before (t=%d):' % (first[0] + 1), {k: '%.3g' % m for k, (b, m) in prev[0].items()})
        gbad = ['%s:%d' % (n, bad(g)) for n, g in model.vars.grads.items() if bad(g)]
        gmax = max(float(g.abs().max()) for g in model.vars.grads.values())
        model.apply_gradients()
        loss = model.total_loss(audio)
        torch.cuda.synchronize()
        pbad = ['%s:%d' % (n, bad(p)) for n, p in model.vars.params.items() if bad(p)]
        print('step %3d loss %.5f audio %.5f |raw|max %.3f align max %.3g sum[min %.3g max %.3g] logits bad %d  saved bad %s  grad max %.3g bad %s  params bad %s'
              % (step, float(loss), float(audio), amax, almax,
                 float(alsum.min()) if alsum is not None else float('nan'), float(alsum.max()) if alsum is not None else float('nan'),
                 bad(logits), rep or '-', gmax, gbad or '-', pbad or '-'), flush=True)
        model.global_step += 1
        if pbad:
            break
    try:
        model.check_device_status()
        print('status clean')
    except Exception as e:      # noqa: BLE001
        print('status:', e)


if __name__ == '__main__':
    main()
