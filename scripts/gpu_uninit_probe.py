# every torch.empty() is filled with NaN (floats) / max (ints): a kernel that reads memory nobody wrote shows up as a
# NaN loss or gradient.  Runs a few train steps of several model configurations.
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
from helpers import make_batch, to_device
from phones_las_amd import model_helper as mh
from phones_las_amd.utils import params_utils as pu

CONFIGS = {
    'cli': dict(num_channels=13, encoder_layers=2, encoder_units=64, use_pyramidal=True, decoder_layers=1, decoder_units=64,
                target_vocab_size=9, bottom_only=True, pass_hidden_state=True, dropout=0.0, sampling_probability=0.0),
    'h256': dict(num_channels=13, encoder_layers=2, encoder_units=256, use_pyramidal=True, decoder_layers=1, decoder_units=256,
                 target_vocab_size=11, bottom_only=True, pass_hidden_state=True, dropout=0.1, sampling_probability=0.1),
    'stacked': dict(num_channels=13, encoder_layers=3, encoder_units=128, use_pyramidal=False, decoder_layers=2,
                    decoder_units=128, target_vocab_size=11, dropout=0.2, sampling_probability=0.1),
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
    for it in range(3):
        loss = model.train_step(feats, labels)
        torch.cuda.synchronize()
        bad = [n for n, t in model.vars.params.items() if not torch.isfinite(t).all()]
        print('%s step %d: loss %s  non-finite params: %s' % (name, it, float(loss), bad[:4]), flush=True)
    pred = model.predict(feats)
    print('%s predict: logits finite %s' % (name, bool(torch.isfinite(pred['logits']).all())), flush=True)
