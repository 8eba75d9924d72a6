# reproduces the order of the full GPU suite: the golden-logit forward passes (tests/test_golden.py), then the CLI
# train -> infer cycle; prints the loss trace and the PER of the cycle
import os, sys, io, contextlib, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from oracle import las_oracle as O
from phones_las_amd import model_helper as mh
from helpers import make_hparams, make_batch, to_device
import test_gpu_cli as t
import train, infer
if os.environ.get('GOLDEN', '1') != '0':
    for att in ('luong', 'bahdanau'):
        ohp, params = make_hparams(att=att)
        model = mh.LasModel(params)
        model.load_variables(O.init_params(ohp, bias_scale=0.1))
        feats, labels = to_device(make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5]))
        loss, logits, _ = model.forward_train(feats, labels)
        torch.cuda.synchronize()
        if os.environ.get('DELMODEL'):
            del model
d = tempfile.mkdtemp()
t._corpus(d)
common = ['--train', os.path.join(d, 'train.tfr'), '--model_dir', os.path.join(d, 'model'), '--encoder_layers', '2',
          '--encoder_units', '64', '--decoder_layers', '1', '--decoder_units', '64', '--use_pyramidal',
          '--bottom_only', '--pass_hidden_state', '--dropout', '0', '--sampling_probability', '0',
          '--batch_size', os.environ.get('BS', '8'), '--num_channels', '13', '--learning_rate', os.environ.get('LR', '0.0005')]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    train.main(train.parse_args(common + ['--num_epochs', os.environ.get('EPOCHS', '600')]))
    if os.environ.get('RESUME'):
        train.main(train.parse_args(common + ['--num_epochs', '5', '--encoder_units', '128']))
    per = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                       '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                       '--num_channels', '13', '--batch_size', '8']))
out = buf.getvalue()
tr = [ln.split('loss = ')[1].split()[0] for ln in out.split('\n') if ': loss = ' in ln]
print('PER %.2f  loss@10,20,30,40: %s  @1170..1200: %s' % (per, tr[:4], tr[-4:]), flush=True)
if os.environ.get('FULLTRACE'):
    print(' '.join('%d:%s' % ((i + 1) * 10, v) for i, v in enumerate(tr) if i % 4 == 3 or float(v) > 0.02 and i > 40), flush=True)
if per > 10:
    print(open(os.path.join(d, 'model', 'infer.txt')).read()[:600])
