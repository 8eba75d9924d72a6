# repeats the CLI train -> infer cycle of tests/test_gpu_cli.py and prints final loss / PER per run (flakiness probe)
import os, sys, tempfile, io, contextlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
if os.environ.get('NANFILL'):      # torch.empty() returns NaN / max-int filled memory: uninitialised reads become visible
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
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
        if os.environ.get('RESUME'):
            train.main(train.parse_args(common + ['--num_epochs', '5', '--encoder_units', '128']))
        per = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                           '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                           '--num_channels', '13', '--batch_size', '8']))
        per2 = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                            '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                            '--num_channels', '13', '--batch_size', '8']))
    out = buf.getvalue()
    last = out.split('step %d: loss = ' % (2 * int(os.environ.get('EPOCHS', '600'))))[1].split()[0]
    print('run %d: loss@1200 %s  PER %.2f  PER(again) %.2f' % (run, last, per, per2), flush=True)
    if os.environ.get('TRACE'):      # per-run loss trace: the first logged step where a run leaves the common trajectory
        tr = [ln.split('loss = ')[1].split()[0] for ln in out.split('\n') if ': loss = ' in ln]
        if run == 0:
            ref_trace = tr
        else:
            diff = [i for i, (a, b) in enumerate(zip(tr, ref_trace)) if a != b]
            if diff:
                i = diff[0]
                print('   first difference at logged point %d (step %d): %s vs %s; then %s vs %s' % (
                    i, (i + 1) * 10, tr[i], ref_trace[i], tr[i + 1:i + 4], ref_trace[i + 1:i + 4]), flush=True)
