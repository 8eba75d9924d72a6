# repeats tests/test_gpu_cli.py::test_train_resume_infer (same flags) and, when the PER is high, shows hypotheses against targets
import os, sys, tempfile, io, contextlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_cli as t
import train, infer
for run in range(int(os.environ.get('RUNS', 5))):
    d = tempfile.mkdtemp()
    t._corpus(d)
    common = ['--train', os.path.join(d, 'train.tfr'), '--model_dir', os.path.join(d, 'model'), '--encoder_layers', '2',
              '--encoder_units', '64', '--decoder_layers', '1', '--decoder_units', '64', '--use_pyramidal',
              '--bottom_only', '--pass_hidden_state', '--dropout', '0', '--sampling_probability', '0',
              '--batch_size', '16', '--num_channels', '13', '--learning_rate', '0.001']
    iargs = ['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'), '--norm', os.path.join(d, 'norm.dmp'),
             '--model_dir', os.path.join(d, 'model'), '--num_channels', '13', '--batch_size', '8']
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        train.main(train.parse_args(common + ['--num_epochs', '800']))
        train.main(train.parse_args(common + ['--num_epochs', '5', '--encoder_units', '128']))
        per = infer.main(infer.parse_args(iargs))
    out = buf.getvalue()
    tr = [float(ln.split('loss = ')[1].split()[0]) for ln in out.split('\n') if ': loss = ' in ln]
    print('run %d: loss at 10/400/790/800 %.4f %.4f %.4f %.4f  max of the last 20 logged %.4f  PER %.2f' % (
        run, tr[0], tr[39], tr[78], tr[79], max(tr[60:80]), per), flush=True)
    if per >= 10:
        h = open(os.path.join(d, 'model', 'infer.txt')).read().split('\n')
        r = open(os.path.join(d, 'model', 'infer_targets.txt')).read().split('\n')
        for a, b in zip(h, r):
            print('   %-30s | %s %s' % (a, b, '' if a == b else '  <--'))
        for extra in range(3):
            with contextlib.redirect_stdout(io.StringIO()) as b2:
                train.main(train.parse_args(common + ['--num_epochs', '50']))
                per = infer.main(infer.parse_args(iargs))
            print('   after 50 more steps: PER %.2f  (last loss %s)' % (per, [ln.split('loss = ')[1].split()[0] for ln in b2.getvalue().split('\n') if ': loss = ' in ln][-1]))
