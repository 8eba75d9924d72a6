# determinism probe: the same forward + backward (no update) repeated on one model / batch; reports the largest
# deviation of the logits and of the flat gradient from the first repetition, relative to the largest magnitude.
# Split-K atomics reorder fp32 sums (1e-6-level noise is expected); anything larger points at a race.
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_batch, to_device
from phones_las_amd import model_helper as mh
from phones_las_amd.utils import params_utils as pu

CONFIGS = {
    'cli': dict(num_channels=13, encoder_layers=2, encoder_units=64, use_pyramidal=True, decoder_layers=1, decoder_units=64,
                target_vocab_size=9, bottom_only=True, pass_hidden_state=True, dropout=0.0, sampling_probability=0.0),
    'h256': dict(num_channels=13, encoder_layers=2, encoder_units=256, use_pyramidal=True, decoder_layers=1, decoder_units=256,
                 target_vocab_size=11, bottom_only=True, pass_hidden_state=True, dropout=0.0, sampling_probability=0.0),
    'stacked': dict(num_channels=13, encoder_layers=3, encoder_units=128, use_pyramidal=False, decoder_layers=2,
                    decoder_units=128, target_vocab_size=11, dropout=0.0, sampling_probability=0.0),
}
for name in os.environ.get('CONFIGS', 'cli,h256,stacked').split(','):
    hp = pu.get_default_hparams()
    for k, v in CONFIGS[name].items():
        hp.set_hparam(k, v)
    model = mh.LasModel(pu.get_encoder_decoder_hparams(hp))
    V = CONFIGS[name]['target_vocab_size']
    src_len = [32, 24, 16, 40, 24, 32, 16, 40]
    tgt_len = [5, 4, 3, 6, 4, 5, 3, 6]
    feats, labels = to_device(make_batch(B=8, T=40, U=6, V=V, src_len=src_len, tgt_len=tgt_len))
    ref = None
    worst_l = worst_g = 0.0
    for it in range(int(os.environ.get('REPS', 12))):
        model.vars.grad.zero_()
        loss, _, dlogits = model.forward_train(feats, labels, None)
        logits = model._last_logits.float().clone() if hasattr(model, '_last_logits') else dlogits.float().clone()
        model.backward(dlogits)
        torch.cuda.synchronize()
        g = model.vars.grad.clone()
        if ref is None:
            ref = (logits, g)
            continue
        dl = float((logits - ref[0]).abs().max() / ref[0].abs().max())
        dg = float((g - ref[1]).abs().max() / ref[1].abs().max())
        if max(dl, dg) > float(os.environ.get('REPORT', '1e-4')):          # name the tensors that moved
            names = []
            for (nm, shape, _), o in zip(model.vars.table, model.vars.offsets):
                n = 1
                for d_ in shape: n *= d_
                a, b_ = g[o:o + n], ref[1][o:o + n]
                dev = float((a - b_).abs().max() / (b_.abs().max() + 1e-30))
                if dev > 1e-4: names.append('%s %.1e' % (nm.split('/', 1)[-1], dev))
            print('  rep %d: logits dev %.1e grad dev %.1e: %s' % (it, dl, dg, '; '.join(names[:12])), flush=True)
        worst_l = max(worst_l, dl)
        worst_g = max(worst_g, dg)
        # scribble over freed blocks so that reads of uninitialised memory show up
        junk = torch.full((1 << 22,), float('nan'), device='cuda'); del junk
    model.check_device_status()
    print('%s: loss %.5f  logits/dlogits dev %.2e  grad dev %.2e' % (name, float(loss), worst_l, worst_g), flush=True)
