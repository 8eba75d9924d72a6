"""End-to-end CLI on the GPU: write a tiny synthetic TFRecord corpus, train.py overfits it, resume works,
infer.py reads hparams.json + checkpoint and reports a low PER (integration tier of SURVEY.md §4)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _corpus(d, n=16, F=13, seed=0):
    from phones_las_amd.utils import tfrecord
    from phones_las_amd.utils.features_utils import save_normalization
    rng = np.random.default_rng(seed)
    phones = ['p%d' % i for i in range(6)]
    open(os.path.join(d, 'vocab.txt'), 'w').write('\n'.join(phones) + '\n')
    protos = rng.standard_normal((len(phones), F)).astype(np.float32) * 2
    truth = []
    with tfrecord.TFRecordWriter(os.path.join(d, 'train.tfr')) as w:
        for _ in range(n):
            ys = [int(v) for v in rng.integers(0, len(phones), size=rng.integers(2, 5))]
            x = np.concatenate([np.repeat(protos[y][None], 8, 0) for y in ys]) + 0.05 * rng.standard_normal((8 * len(ys), F))
            w.write(tfrecord.make_example(x.astype(np.float32), [phones[y] for y in ys]))
            truth.append(' '.join(phones[y] for y in ys))
    save_normalization(os.path.join(d, 'norm.dmp'), np.zeros(F, np.float32), np.ones(F, np.float32))
    return truth


def test_train_resume_infer(tmp_path, capsys):
    import train, infer
    d = str(tmp_path)
    truth = _corpus(d)
    common = ['--train', os.path.join(d, 'train.tfr'), '--model_dir', os.path.join(d, 'model'), '--encoder_layers', '2',
              '--encoder_units', '64', '--decoder_layers', '1', '--decoder_units', '64', '--use_pyramidal',
              '--bottom_only', '--pass_hidden_state', '--dropout', '0', '--sampling_probability', '0',
              '--batch_size', '16', '--num_channels', '13', '--learning_rate', '0.001']
    # (Full-batch steps: with two alternating batches of 8 the loss of this toy problem spikes now and then once it is
    #  near zero -- Adam with vanishing gradients -- and fp32 sums through atomics make two runs agree to rounding only,
    #  so about 3 % of the runs were evaluated right after a spike and ended at 24 % PER.  With all 16 utterances in
    #  every step the loss falls monotonically; scripts/gpu_cli_flaky.py and gpu_golden_then_cli.py are the probes.)
    train.main(train.parse_args(common + ['--num_epochs', '1200']))     # 16 utts / 16 = 1 step per epoch -> 1200 steps
    out = capsys.readouterr().out
    assert 'finished at global_step 1200' in out
    first = float(out.split('step 10: loss = ')[1].split()[0].rstrip(','))
    last = float(out.split('step 1200: loss = ')[1].split()[0].rstrip(','))
    assert last < 0.2 * first
    # the reference's TRAIN-mode log line carries the last batch's mean edit distance beside the loss (model_helper.py:435-439)
    ed_first = float(out.split('step 10: loss = ')[1].split('edit_distance = ')[1].split()[0])
    ed_last = float(out.split('step 1200: loss = ')[1].split('edit_distance = ')[1].split()[0])
    assert ed_first > 0.3 and ed_last < 0.05, (ed_first, ed_last)
    assert os.path.exists(os.path.join(d, 'model', 'hparams.json'))
    # resume: the checkpoint restores the step counter; hparams.json wins over the (different) CLI value
    train.main(train.parse_args(common + ['--num_epochs', '5', '--encoder_units', '128']))
    out = capsys.readouterr().out
    assert 'restored' in out and 'at global_step 1200' in out
    per = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                       '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                       '--num_channels', '13', '--batch_size', '8']))
    def sentences_right():
        hyp = open(os.path.join(d, 'model', 'infer.txt')).read().split('\n')
        ref = open(os.path.join(d, 'model', 'infer_targets.txt')).read().split('\n')
        return sum(a.strip() == b.strip() for a, b in zip(hyp, ref))

    # Training is bit-reproducible since round 3 (the weight-gradient K slices, bias sums and norms are added in a fixed order:
    # test_training_is_bit_reproducible), so this run has ONE trajectory, not a distribution: round 2's retries and its
    # `per < 55` bound (for the one run in seven that ended on a plateau) are gone.
    # (ONE trajectory per build, and Adam on a loss near zero spikes now and then: this build's trajectory has the loss at 0.022
    #  at step 800 -- thirty times its neighbours -- where one greedy hypothesis never emits </s> (15 of 16 right, PER 24 %);
    #  at step 1200 it is at 0.0005 like every perturbed run.  scripts/gpu_cli_robust.py samples the trajectories a kernel change
    #  draws from (speller weight-gradient slices through atomics): 12 of 12 runs end with 16 sentences right and PER 0 at
    #  800 and at 1200 steps, with and without the resume.  The criterion stays the strict one.)
    assert sentences_right() == 16 and per < 10.0, (sentences_right(), per)
    assert len(open(os.path.join(d, 'model', 'infer.txt')).read().split('\n')) == 16
    # the three output files, as the reference writes them (infer.py:269-271,345-359): infer.txt = to_text of the ids (cut at
    # the first </s> symbol), infer.dmp = joblib list of {'transcription': line}, infer_targets.txt = the targets
    from joblib import load
    lines = open(os.path.join(d, 'model', 'infer.txt')).read().split('\n')
    assert load(os.path.join(d, 'model', 'infer.dmp')) == [{'transcription': l} for l in lines]
    assert open(os.path.join(d, 'model', 'infer_targets.txt')).read().split('\n') == truth
    assert all(set(l.split(' ')) <= set('p%d' % i for i in range(6)) for l in lines)           # no <s> / </s> symbols
    # --plain_targets (infer.py:246-271): a `sound,lang,phrase` CSV replaces the TFRecord's labels as targets; the phrase is
    # lower-cased and split; here the targets are the model's own hypotheses with one symbol appended to every second one, so
    # PER = (appended symbols) / (all target symbols) whatever the model got right
    csv = os.path.join(d, 'plain.csv')
    plain = []
    with open(csv, 'w') as f:
        for i, t in enumerate(lines):
            toks = t.split(' ')
            if i % 2:
                toks = toks + ['p0']
            plain.append(toks)
            f.write('utt%d.wav,en,%s\n' % (i, ' '.join(x.upper() for x in toks)))
    per_plain = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                             '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                             '--num_channels', '13', '--batch_size', '8', '--plain_targets', csv]))
    assert open(os.path.join(d, 'model', 'infer_targets.txt')).read().split('\n') == [' '.join(t) for t in plain]
    assert open(os.path.join(d, 'model', 'infer.txt')).read().split('\n') == lines
    assert abs(per_plain - 100.0 * 8 / sum(len(t) for t in plain)) < 1e-6 and 'Optimistic PER' in capsys.readouterr().out
    # beam search over the same checkpoint (infer.py --beam_width) and the stand-alone evaluation (eval.py)
    per_beam = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                            '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                            '--num_channels', '13', '--batch_size', '8', '--beam_width', '3']))
    assert per_beam < 10.0 and sentences_right() == 16 and 'Optimistic PER' in capsys.readouterr().out
    import eval as eval_cli
    loss, ed = eval_cli.main(eval_cli.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                                  '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                                  '--num_channels', '13', '--batch_size', '8']))
    assert np.isfinite(loss) and 0.0 <= ed < 0.1


def test_train_with_the_reference_default_architecture_flags(tmp_path, capsys):
    # train.py defaults (train.py:34-75): stacked (non-pyramidal) 3x128 BiLSTM listener, 2x128 decoder wrapped by Luong
    # attention, dropout 0.2, sampling_probability 0.1 -- only the data-dependent flags are given
    import train
    d = str(tmp_path)
    _corpus(d, n=8)
    train.main(train.parse_args(['--train', os.path.join(d, 'train.tfr'), '--model_dir', os.path.join(d, 'model'),
                                 '--num_channels', '13', '--batch_size', '8', '--num_epochs', '30']))
    out = capsys.readouterr().out
    assert 'finished at global_step 30' in out
    first = float(out.split('step 10: loss = ')[1].split()[0].rstrip(','))
    last = float(out.split('step 30: loss = ')[1].split()[0].rstrip(','))
    assert np.isfinite(last) and last < first
    # the same (stacked, per-layer encoder states: no 'embedding' in the predictions, model_helper.py:259-268) through infer.py
    import infer
    per = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                       '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                       '--num_channels', '13', '--batch_size', '8']))
    assert np.isfinite(per)


def test_train_resume_infer_with_unit_counts_the_kernels_are_not_built_for(tmp_path, capsys):
    # las/ops.py:10-12 takes any --encoder_units / --decoder_units; 96 / 72 run zero-padded at 128 / 128 (model_helper.
    # physical_params).  hparams.json keeps the numbers the user gave, the checkpoint restores, infer.py decodes.
    import json
    import train, infer
    d = str(tmp_path)
    _corpus(d, n=8)
    common = ['--train', os.path.join(d, 'train.tfr'), '--model_dir', os.path.join(d, 'model'), '--encoder_layers', '2',
              '--encoder_units', '96', '--decoder_layers', '1', '--decoder_units', '72', '--use_pyramidal', '--bottom_only',
              '--dropout', '0', '--sampling_probability', '0', '--batch_size', '8', '--num_channels', '13']
    train.main(train.parse_args(common + ['--num_epochs', '200']))
    out = capsys.readouterr().out
    assert 'finished at global_step 200' in out
    first = float(out.split('step 10: loss = ')[1].split()[0].rstrip(','))
    last = float(out.split('step 200: loss = ')[1].split()[0].rstrip(','))
    assert last < 0.5 * first
    hp = json.load(open(os.path.join(d, 'model', 'hparams.json')))
    hp = json.loads(hp) if isinstance(hp, str) else hp
    assert hp['encoder_units'] == 96 and hp['decoder_units'] == 72
    train.main(train.parse_args(common + ['--num_epochs', '5']))
    out = capsys.readouterr().out
    assert 'restored' in out and 'at global_step 200' in out
    per = infer.main(infer.parse_args(['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'),
                                       '--norm', os.path.join(d, 'norm.dmp'), '--model_dir', os.path.join(d, 'model'),
                                       '--num_channels', '13', '--batch_size', '8']))
    assert np.isfinite(per)          # (200 steps of this toy problem do not bound the PER: 2 % to 96 % from run to run; the loss is the criterion)


def _binf_csv(path, phones, nf=6, seed=3):
    """A binary-feature map in the reference's CSV layout (misc/binf_map*.csv: a header row of phones, one row per feature)."""
    rng = np.random.default_rng(seed)
    with open(path, 'w') as f:
        f.write(',' + ','.join(phones) + '\n')
        codes = set()
        cols = []
        while len(cols) < len(phones):                     # distinct feature vectors per phone
            c = tuple(int(v) for v in rng.integers(0, 2, nf))
            if c not in codes and any(c):
                codes.add(c)
                cols.append(c)
        for k in range(nf):
            f.write('f%d,' % k + ','.join(str(c[k]) for c in cols) + '\n')


@pytest.mark.parametrize('mode', ['binf_projection', 'sigmoid', 'multitask', 'binf_trainable'])
def test_binary_feature_cli_modes_train_and_infer(tmp_path, capsys, mode):
    """The reference's --binary_outputs flag family through train.py / infer.py (train.py:117-127, infer.py:203-223):
    --binf_projection (DenseBinfDecoder), --binary_outputs alone (sigmoid-output decoder: feature logits, InferenceHelper
    decode), --multitask (phone + binary decoders); infer.py --use_phones_from_binf reads sample_ids_phones_binf."""
    import train, infer
    d = str(tmp_path)
    _corpus(d)
    phones = ['p%d' % i for i in range(6)]
    binf = os.path.join(d, 'binf_map.csv')
    _binf_csv(binf, phones)
    flags = ['--binary_outputs', '--output_ipa', '--binf_map', binf]
    if mode in ('binf_projection', 'multitask', 'binf_trainable'):
        flags += ['--binf_projection']
    if mode == 'binf_trainable':                           # the feature map as a variable (model_helper.py:181-186), checkpointed
        flags += ['--binf_trainable']
    if mode == 'multitask':
        flags += ['--multitask']
    train.main(train.parse_args(['--train', os.path.join(d, 'train.tfr'), '--model_dir', os.path.join(d, 'model'), '--encoder_layers', '2',
                                 '--encoder_units', '64', '--decoder_layers', '1', '--decoder_units', '64', '--use_pyramidal',
                                 '--bottom_only', '--pass_hidden_state', '--dropout', '0', '--sampling_probability', '0',
                                 '--batch_size', '16', '--num_channels', '13', '--learning_rate', '0.002', '--num_epochs', '300'] + flags))
    out = capsys.readouterr().out
    assert 'finished at global_step 300' in out
    first = float(out.split('step 10: loss = ')[1].split()[0].rstrip(','))
    last = float(out.split('step 300: loss = ')[1].split()[0].rstrip(','))
    assert np.isfinite(last) and last < 0.6 * first, (first, last)
    base = ['--data', os.path.join(d, 'train.tfr'), '--vocab', os.path.join(d, 'vocab.txt'), '--norm', os.path.join(d, 'norm.dmp'),
            '--model_dir', os.path.join(d, 'model'), '--num_channels', '13', '--batch_size', '8', '--binf_map', binf]
    per = infer.main(infer.parse_args(base))
    assert np.isfinite(per)
    if mode != 'sigmoid':                                  # phone ids decoded from the binary-feature decoder
        per_b = infer.main(infer.parse_args(base + ['--use_phones_from_binf']))
        assert np.isfinite(per_b) and per_b < 60.0, per_b
