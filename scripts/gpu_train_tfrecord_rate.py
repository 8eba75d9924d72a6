# End-to-end rate of train.py on a synthetic T=800 TFRecord corpus at the metric-M model (SURVEY 8(d) flags): the CLI reads
# the records through the C parser + prefetch thread and prints "throughput after step 10".  Compare with bench.py's value.
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, '.')
import train
from phones_las_amd.utils import tfrecord as tfr
rng = np.random.default_rng(0)
T, F, U, N, B = 800, 40, 80, 512, 64
d = tempfile.mkdtemp()
toks = ['p%d' % i for i in range(61)]
open(os.path.join(d, 'vocab.txt'), 'w').write('\n'.join(toks) + '\n')
t0 = time.time()
x = rng.standard_normal((T + N, F)).astype(np.float32)
with tfr.TFRecordWriter(os.path.join(d, 'train.tfr')) as w:
    for i in range(N):
        w.write(tfr.make_example(x[i:i + T], [toks[int(k)] for k in rng.integers(0, 61, U - 1)]))
print('corpus: %d utterances, %.0f MB, written in %.1f s' % (N, os.path.getsize(os.path.join(d, 'train.tfr')) / 1e6, time.time() - t0))
argv = ['--train', os.path.join(d, 'train.tfr'), '--model_dir', os.path.join(d, 'model'), '--num_channels', str(F), '--batch_size', str(B),
        '--num_epochs', '12', '--encoder_layers', '3', '--encoder_units', '256', '--use_pyramidal', '--decoder_layers', '1',
        '--decoder_units', '256', '--attention_type', 'luong', '--bottom_only', '--pass_hidden_state', '--dropout', '0',
        '--sampling_probability', '0', '--l2_reg_scale', '1e-6', '--learning_rate', '1e-3'] + sys.argv[1:]
train.main(train.parse_args(argv))
