import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_gpu_cli import _corpus
import train, infer
d = tempfile.mkdtemp()
_corpus(d)
common = ['--train', os.path.join(d, 'train.tfr'), '--model_dir', os.path.join(d, 'model'), '--encoder_layers', '2',
          '--encoder_units', '64', '--decoder_layers', '1', '--decoder_units', '64', '--use_pyramidal',
          '--bottom_only', '--pass_hidden_state', '--dropout', '0', '--sampling_probability', '0',
          '--batch_size', '8', '--num_channels', '13', '--learning_rate', '0.01']
train.main(train.parse_args(common + ['--num_epochs', '150']))
per = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                   '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                   '--num_channels', '13', '--batch_size', '8']))
print(open(os.path.join(d, 'model', 'infer.txt')).read())
print('----')
print(open(os.path.join(d, 'model', 'infer_targets.txt')).read())
