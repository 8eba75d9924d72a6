#!/usr/bin/env python3
"""Train Listen-Attend-Spell on MI355X with the flag surface of the reference's train.py (train.py:12-109):
same flag names and defaults, vocab.txt / norm.dmp looked up next to --train (train.py:112-114), hparams.json in
--model_dir (existing file wins unless --reset), TFRecord input.  The Estimator is replaced by LasModel
(phones-las_amd/model_helper.py) driving liblas_hip.so; data parallelism = one process per GPU:

    python train.py --train data/train.tfr --model_dir out --use_pyramidal --bottom_only --pass_hidden_state ...
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...

TPU / T2T flags are accepted for command-line compatibility and rejected when used."""
import argparse
import multiprocessing
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Listen, Attend and Spell (LAS) on MI355X (HIP), drop-in for the '
                                            'TensorFlow reference CLI.')
    p.add_argument('--train', type=str, required=True, help='training data in TFRecord format')
    p.add_argument('--valid', type=str, help='validation data in TFRecord format')
    p.add_argument('--t2t_format', action='store_true')
    p.add_argument('--t2t_problem_name', type=str)
    p.add_argument('--mapping', type=str, help='additional mapping when evaluation')
    p.add_argument('--model_dir', type=str, required=True, help='path of saving model')
    p.add_argument('--eval_secs', type=int, default=300)
    p.add_argument('--encoder_units', type=int, default=128)
    p.add_argument('--encoder_layers', type=int, default=3)
    p.add_argument('--use_pyramidal', action='store_true')
    p.add_argument('--unidirectional', action='store_true')
    p.add_argument('--decoder_units', type=int, default=128)
    p.add_argument('--decoder_layers', type=int, default=2)
    p.add_argument('--embedding_size', type=int, default=0)
    p.add_argument('--sampling_probability', type=float, default=0.1)
    p.add_argument('--attention_type', type=str, default='luong',
                   choices=['luong', 'bahdanau', 'custom', 'luong_monotonic', 'bahdanau_monotonic'])
    p.add_argument('--attention_layer_size', type=int)
    p.add_argument('--bottom_only', action='store_true')
    p.add_argument('--pass_hidden_state', action='store_true')
    p.add_argument('--batch_size', type=int, default=8)
    p.add_argument('--num_parallel_calls', type=int, default=multiprocessing.cpu_count())
    p.add_argument('--num_channels', type=int)
    p.add_argument('--num_epochs', type=int, default=150)
    p.add_argument('--learning_rate', type=float, default=1e-3)
    p.add_argument('--dropout', type=float, default=0.2)
    p.add_argument('--l2_reg_scale', type=float, default=1e-6)
    p.add_argument('--add_noise', type=int, default=0)
    p.add_argument('--noise_std', type=float, default=0.1)
    p.add_argument('--slow_input', action='store_true',
                   help='read the training TFRecords with the pure-Python parser on the step loop\'s thread instead of the '
                        'C parser + prefetch thread (same batches for the same seed)')
    p.add_argument('--dp_overlap', action='store_true',
                   help='multi-GPU (torch.distributed.run): exchange the gradients in two buckets beside the backward pass')
    p.add_argument('--binary_outputs', action='store_true')
    p.add_argument('--output_ipa', action='store_true')
    p.add_argument('--binf_map', type=str, default='misc/binf_map.csv')
    p.add_argument('--ctc_weight', type=float, default=-1.)
    p.add_argument('--reset', help='Reset HParams.', action='store_true')
    p.add_argument('--binf_sampling', action='store_true')
    p.add_argument('--binf_projection', action='store_true')
    p.add_argument('--binf_projection_reg_weight', type=float, default=1.0)
    p.add_argument('--binf_trainable', action='store_true')
    p.add_argument('--multitask', action='store_true')
    p.add_argument('--tpu_name', type=str, default='')
    p.add_argument('--max_frames', type=int, default=-1)
    p.add_argument('--max_symbols', type=int, default=-1)
    p.add_argument('--tpu_checkpoints_interval', type=int, default=600)
    p.add_argument('--t2t_features_hparams_override', type=str, default='')
    return p.parse_args(argv)


def to_device(features, labels, dev):
    f = {k: torch.from_numpy(v).to(dev) for k, v in features.items()}
    l = {k: torch.from_numpy(v).to(dev) for k, v in labels.items()} if labels is not None else None
    return f, l


def save_checkpoint(model, path):
    v = model.vars
    # global_step = steps the host has issued; adam_step = updates the device has APPLIED + 1 (its Adam t).  They differ only
    # when a step's update was withheld after a persistent-kernel timeout (las_counter_add_unless) -- and this function is
    # only reached behind a check_device_status() that raises in that case -- but the checkpoint keeps both, so a resumed run
    # continues with the device's own bias-correction count whatever happened.
    torch.save({'flat': v.flat.cpu(), 'm': v.m.cpu(), 'v': v.v.cpu(), 'global_step': model.global_step,
                'adam_step': int(model.step_dev.item()),
                'names': [n for n, _, _ in v.table], 'offsets': v.offsets}, path + '.tmp')
    os.replace(path + '.tmp', path)


def load_checkpoint(model, path):
    ck = torch.load(path, map_location='cpu')
    v = model.vars
    if ck['names'] != [n for n, _, _ in v.table] or ck['offsets'] != v.offsets:
        raise ValueError('checkpoint %s does not match the model built from hparams.json' % path)
    v.flat.copy_(ck['flat']); v.m.copy_(ck['m']); v.v.copy_(ck['v'])
    model.global_step = int(ck['global_step'])
    model.step_dev.fill_(int(ck.get('adam_step', model.global_step + 1)))
    model.refresh_images()


def main(args):
    if args.t2t_format or args.tpu_name:
        raise SystemExit('--t2t_format / --tpu_name are TensorFlow-only input/back-end options and are not supported')
    if args.binary_outputs and not args.output_ipa:
        raise SystemExit('--binary_outputs needs --output_ipa: without it the reference hands no binf2phone map to the model '
                         '(train.py:125-127) and its TRAIN graph has no decoder inputs (model_helper.py:199-200)')
    from phones_las_amd import dp, utils
    from phones_las_amd import model_helper as mh

    rank, world, local = dp.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    train_dir = os.path.dirname(args.train)
    vocab_name = os.path.join(train_dir, 'vocab.txt')
    norm_name = os.path.join(train_dir, 'norm.dmp')
    vocab_list = utils.load_vocab(vocab_name)
    binf2phone_np, mapping, binf_count = None, None, None
    if args.binary_outputs:               # train.py:117-126 of the reference
        if args.mapping is not None:
            vocab_list, mapping = utils.get_mapping(args.mapping, vocab_name)
            args.mapping = None
        binf2phone = utils.load_binf2phone(args.binf_map, vocab_list)
        binf_count = len(binf2phone.index)
        if args.output_ipa:
            binf2phone_np = binf2phone.values
    if not args.num_channels:
        first = next(iter(utils.read_dataset(args.train, None)()))
        args.num_channels = int(first[0].shape[1])
    # one writer per model_dir: rank 0 creates / merges hparams.json (atomic replace), the others read it afterwards
    if rank == 0:
        hparams = utils.create_hparams(args, len(vocab_list), binf_count, utils.SOS_ID, utils.EOS_ID)
    dp.barrier()
    if rank != 0:
        hparams = utils.create_hparams(args, len(vocab_list), binf_count, utils.SOS_ID, utils.EOS_ID, write=False)
    if mapping is not None:
        hparams.del_hparam('mapping')
        hparams.add_hparam('mapping', mapping)
    model = mh.LasModel(hparams, world_size=world, binf2phone=binf2phone_np, rank=rank)
    if world > 1 and getattr(args, 'dp_overlap', False):
        model.enable_exchange_overlap()
    ckpt = os.path.join(args.model_dir, 'checkpoint.pt')
    if os.path.exists(ckpt):
        load_checkpoint(model, ckpt)
        print('restored %s at global_step %d' % (ckpt, model.global_step))
    if rank == 0:
        print('Trainable parameters: %d' % model.vars.num_parameters())

    global_batch = args.batch_size * world

    def make_input(path, epochs, infer=False):
        return utils.input_fn(path, vocab_name, norm_name, num_channels=hparams.num_channels, batch_size=global_batch,
                              num_epochs=epochs, num_parallel_calls=args.num_parallel_calls,
                              max_frames=args.max_frames, max_symbols=args.max_symbols, is_infer=infer, seed=1234)

    def make_train_input():
        """The training stream: same pipeline semantics, parsed in C by a prefetch thread, normalised / cast / padded on the
        device (phones-las_amd/utils/fast_input.py); --slow_input keeps the pure-Python reader."""
        if args.slow_input:
            return make_input(args.train, args.num_epochs)
        from phones_las_amd.utils.fast_input import fast_input_fn
        return fast_input_fn(args.train, vocab_name, norm_name, num_channels=hparams.num_channels, batch_size=global_batch,
                             num_epochs=args.num_epochs, max_frames=args.max_frames, max_symbols=args.max_symbols, seed=1234,
                             time_multiple=model.listener.time_multiple, shard=(rank, world))

    max_steps = args.num_epochs * 1000 * args.batch_size          # train.py:189,202 (quirk B2)
    t_last, last_eval, t0 = time.time(), time.time(), time.time()
    n_utt, t_log = 0, time.time()
    for features, labels in make_train_input():
        if model.global_step >= max_steps:
            break
        # decoder steps of this batch = the longest target: a host number on both input paths (no device round trip)
        num_steps = labels.pop('max_target_length', None)
        if world > 1 and args.slow_input:
            # (the C input path hands every rank ITS shard: it parses and copies 1/world of the global batch; the pure-Python
            # reader yields the whole global batch on the host and is cut here)
            features, labels = dp.shard_batch(features, rank, world), dp.shard_batch(labels, rank, world)
        if num_steps is None:
            num_steps = int(labels['target_sequence_length'].max())
        elif world > 1:
            num_steps = min(num_steps, labels['targets_inputs'].shape[1])
        if torch.is_tensor(features['encoder_inputs']):
            f, l = features, labels
        else:
            f, l = to_device(features, labels, dev)
        loss = model.train_step(f, l, num_steps=num_steps)
        n_utt += global_batch
        if model.global_step % 10 == 0:                            # LoggingTensorHook(every_n_iter=10)
            lv = dp.mean_scalar(float(loss))
            model.check_device_status()          # bounded waits of the persistent kernels: fail loudly, never silently
            dt = time.time() - t_last            # (before the metric below: its argmax + copy are not the step loop's time)
            # the reference's train_log_data (model_helper.py:435-439): loss and the last batch's mean edit distance
            ed = dp.mean_scalar(model.train_edit_distance(l))
            if rank == 0:
                print('step %d: loss = %.5f, edit_distance = %.5f (%.2f utt/s)' % (model.global_step, lv, ed, 10 * global_batch / max(dt, 1e-9)))
            t_last = time.time()
            if model.global_step == 10:          # throughput of the run without its first steps (allocations, indexing)
                n_utt, t_log = 0, time.time()
            # rank 0's clock decides when to evaluate (every replica must enter the evaluation at the same step)
            if args.valid and dp.broadcast_int(time.time() - last_eval > args.eval_secs):
                evaluate(model, make_input(args.valid, 1, True), dev, rank, world)
                last_eval = time.time()
        if model.global_step % 500 == 0:
            model.check_device_status()          # never checkpoint parameters of a step whose kernels timed out
            if rank == 0:
                save_checkpoint(model, ckpt)
    model.check_device_status()
    torch.cuda.synchronize()
    t_end = time.time()                          # (the steps' clock stops before the final checkpoint is written)
    if rank == 0:
        save_checkpoint(model, ckpt)
        print('finished at global_step %d in %.1f s' % (model.global_step, time.time() - t0))
        if model.global_step > 10 and n_utt:
            print('throughput after step 10: %.1f utterances/s' % (n_utt / max(t_end - t_log, 1e-9)))
    if args.valid:
        evaluate(model, make_input(args.valid, 1, True), dev, rank, world)


def evaluate(model, batches, dev, rank=0, world=1):
    """tf.estimator evaluate: streaming mean of loss and normalised edit distance (model_helper.py:299-309).  With several
    replicas every batch is dealt round-robin to them (every replica walks the same batches) and the sums are
    all-reduced at the end, so all replicas leave the evaluation together."""
    from phones_las_amd import dp
    loss_sum, n_batches, ed_sum, n_utt = 0.0, 0, 0.0, 0
    for i, (features, labels) in enumerate(batches):
        if world > 1 and i % world != rank:
            continue
        f, l = to_device(features, labels, dev)
        loss, ed, _ = model.evaluate(f, l)
        loss_sum += float(loss)
        n_batches += 1
        ed_sum += float(np.sum(ed))
        n_utt += len(ed)
    model.check_device_status()
    if world > 1:
        loss_sum, n_batches, ed_sum, n_utt = dp.sum_floats([loss_sum, n_batches, ed_sum, n_utt])
    if not n_batches:
        return None, None
    res = (loss_sum / n_batches, ed_sum / max(n_utt, 1))
    if rank == 0:
        print('eval: loss = %.5f, edit_distance = %.5f' % res)
    return res


if __name__ == '__main__':
    main(parse_args())
