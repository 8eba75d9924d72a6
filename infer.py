#!/usr/bin/env python3
"""Greedy decoding + PER with the flag surface of the reference's infer.py (infer.py:21-63,192-359): reads
hparams.json and the checkpoint from --model_dir, decodes --data (TFRecord), writes model_dir/infer.txt and
infer_targets.txt and prints PER = 100 * sum(edit distance) / sum(len(reference)) (infer.py:270-303,338), with the
optional 61->39 style --mapping.  Beam search, frame-level binary-feature accuracy and IPA conversion are not on
the HIP path."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='LAS inference on MI355X (HIP), drop-in for the reference CLI.')
    p.add_argument('--data', type=str, required=True, help='inference data in TFRecord format')
    p.add_argument('--plain_targets', type=str, help='Path to CSV file with targets.')
    p.add_argument('--vocab', type=str, required=True, help='vocabulary table, listing vocabulary line by line')
    p.add_argument('--norm', type=str, default=None, help='normalization params')
    p.add_argument('--mapping', type=str, help='additional mapping when evaluation')
    p.add_argument('--model_dir', type=str, required=True, help='path of imported model')
    p.add_argument('--beam_width', type=int, default=0)
    p.add_argument('--batch_size', type=int, default=8)
    p.add_argument('--num_channels', type=int, default=39)
    p.add_argument('--delimiter', help='Symbols delimiter. Default: " "', type=str, default=' ')
    p.add_argument('--take', help='Use this number of elements (0 for all).', type=int, default=0)
    p.add_argument('--binf_map', type=str, default='misc/binf_map.csv')
    p.add_argument('--use_phones_from_binf', action='store_true')
    p.add_argument('--convert_targets_to_ipa', action='store_true')
    p.add_argument('--calc_frame_binf_accuracy', action='store_true')
    p.add_argument('--mapping_for_frame_accuracy', type=str)
    p.add_argument('--encoder_frame_step', type=int, default=40)
    p.add_argument('--use_markup_segments', action='store_true')
    return p.parse_args(argv)


def to_text(vocab_list, sample_ids, delimiter=' '):
    """infer.py:65-67."""
    return delimiter.join(vocab_list[i] for i in sample_ids)


def main(args):
    if args.calc_frame_binf_accuracy or args.convert_targets_to_ipa:
        raise SystemExit('--calc_frame_binf_accuracy / --convert_targets_to_ipa (TIMIT markup and IPA analysis) are not on the HIP path')
    from phones_las_amd import utils
    from phones_las_amd import model_helper as mh
    from phones_las_amd.utils.metrics_utils import _levenshtein
    from train import load_checkpoint, to_device

    vocab_list = utils.load_vocab(args.vocab)
    args_h = argparse.Namespace(model_dir=args.model_dir, mapping=args.mapping)
    hparams = utils.create_hparams(args_h)              # requires an existing hparams.json (params_utils.py:96-97)
    hparams.decoder.set_hparam('beam_width', args.beam_width)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    binf2phone_np = None
    if hparams.decoder.binary_outputs:       # infer.py:203-210 of the reference
        if args.mapping is not None:
            vocab_list, mapping_list = utils.get_mapping(args.mapping, args.vocab)
            hparams.del_hparam('mapping')
            hparams.add_hparam('mapping', mapping_list)
        binf2phone_np = utils.load_binf2phone(args.binf_map, vocab_list).values
    model = mh.LasModel(hparams, binf2phone=binf2phone_np)
    load_checkpoint(model, os.path.join(args.model_dir, 'checkpoint.pt'))

    mapping = hparams.mapping
    batches = utils.input_fn(args.data, args.vocab, args.norm, num_channels=args.num_channels,
                             batch_size=args.batch_size, num_epochs=1, take=args.take, is_infer=True)
    hyps, refs = [], []
    optimistic_err = 0
    for features, labels in batches:
        f, _ = to_device(features, None, dev)
        pred = model.predict(f, transparent_projection=bool(args.use_phones_from_binf))   # transcribe_audio_file.py:90 couples the two
        # infer.py:223: the phones decoded from the binary-feature decoder, or the phone decoder's
        ids = pred['sample_ids_phones_binf' if args.use_phones_from_binf else 'sample_ids'].cpu().numpy()   # [B,T] / [B,T,K]
        for b in range(ids.shape[0]):
            t = labels['targets_outputs'][b][:labels['target_sequence_length'][b] - 1].tolist()
            if mapping is not None:
                t = [mapping[x] for x in t]
                t = [x for x in t if x >= 0]
            beams = ids[b].T if ids.ndim == 3 else ids[b][None]
            cut = []
            for beam in beams:
                i = beam.tolist() + [utils.EOS_ID]
                cut.append(i[:i.index(utils.EOS_ID)])    # cut at the first EOS (infer.py:296-298)
            optimistic_err += min(_levenshtein(i, t) for i in cut)
            hyps.append(cut[0])                          # the best-scoring beam is the hypothesis
            refs.append(t)
    model.check_device_status()          # a timed-out persistent kernel would have produced invalid hypotheses
    err = sum(_levenshtein(h, r) for h, r in zip(hyps, refs))
    tot = sum(len(r) for r in refs)
    with open(os.path.join(args.model_dir, 'infer.txt'), 'w') as f:
        f.write('\n'.join(to_text(vocab_list, h, args.delimiter) for h in hyps))
    with open(os.path.join(args.model_dir, 'infer_targets.txt'), 'w') as f:
        f.write('\n'.join(to_text(vocab_list, r, args.delimiter) for r in refs))
    per = 100.0 * err / max(tot, 1)
    print('PER: %2.2f%%' % per)
    if args.beam_width > 0:                              # infer.py:345-346
        print('Optimistic PER: %2.2f%%' % (100.0 * optimistic_err / max(tot, 1)))
    return per


if __name__ == '__main__':
    main(parse_args())
