# How robust is the CLI test's train -> infer outcome to rounding-level perturbations?  LAS_TN_ATOMIC=1 makes the speller's
# weight-gradient K slices meet in fp32 atomics again, so every run takes a slightly different trajectory (the distribution a
# kernel change samples from).  Prints sentences right / PER per run for the given EPOCHS (full batch of 16, lr 1e-3).
import os, sys, tempfile, io, contextlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_cli as t
import train, infer
for ep in os.environ.get('EPOCHS', '800,1200').split(','):
    res = []
    for run in range(int(os.environ.get('RUNS', 5))):
        d = tempfile.mkdtemp()
        t._corpus(d)
        common = ['--train', os.path.join(d, 'train.tfr'), '--model_dir', os.path.join(d, 'model'), '--encoder_layers', '2',
                  '--encoder_units', '64', '--decoder_layers', '1', '--decoder_units', '64', '--use_pyramidal',
                  '--bottom_only', '--pass_hidden_state', '--dropout', '0', '--sampling_probability', '0',
                  '--batch_size', '16', '--num_channels', '13', '--learning_rate', os.environ.get('LR', '0.001')]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            train.main(train.parse_args(common + ['--num_epochs', ep]))
            if os.environ.get('RESUME'):          # the test's second call: restore, five more steps
                train.main(train.parse_args(common + ['--num_epochs', '5', '--encoder_units', '128']))
            per = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                               '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                               '--num_channels', '13', '--batch_size', '8']))
        hyp = open(os.path.join(d, 'model', 'infer.txt')).read().split('\n')
        ref = open(os.path.join(d, 'model', 'infer_targets.txt')).read().split('\n')
        right = sum(a.strip() == b.strip() for a, b in zip(hyp, ref))
        last = buf.getvalue().split('step %s: loss = ' % ep)[1].split()[0]
        res.append((right, round(per, 1), last))
    print('epochs %s: %s' % (ep, res), flush=True)
