# repeats the CLI train -> infer cycle of tests/test_gpu_cli.py and prints final loss / PER per run (flakiness probe)
import os, sys, tempfile, io, contextlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_cli as t
import train, infer
for run in range(int(os.environ.get('RUNS', 6))):
    d = tempfile.mkdtemp()
    t._corpus(d)
    common = ['--train', os.path.join(d, 'train.tfr'), '--model_dir', os.path.join(d, 'model'), '--encoder_layers', '2',
              '--encoder_units', '64', '--decoder_layers', '1', '--decoder_units', '64', '--use_pyramidal',
              '--bottom_only', '--pass_hidden_state', '--dropout', '0', '--sampling_probability', '0',
              '--batch_size', '8', '--num_channels', '13', '--learning_rate', os.environ.get('LR', '0.003')]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        train.main(train.parse_args(common + ['--num_epochs', os.environ.get('EPOCHS', '600')]))
        per = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                           '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                           '--num_channels', '13', '--batch_size', '8']))
        per2 = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                            '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                            '--num_channels', '13', '--batch_size', '8']))
    out = buf.getvalue()
    last = out.split('step %d: loss = ' % (2 * int(os.environ.get('EPOCHS', '600'))))[1].split()[0]
    print('run %d: loss@1200 %s  PER %.2f  PER(again) %.2f' % (run, last, per, per2), flush=True)
