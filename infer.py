#!/usr/bin/env python3
"""Decoding with the flag surface and the OUTPUT FILES of the reference's infer.py (infer.py:21-63,192-359): reads
hparams.json and the checkpoint from --model_dir, decodes --data (TFRecord; no shuffle, the tail batch kept -- SURVEY B1)
greedily or with --beam_width, and writes, as the reference does (infer.py:345-359):

  model_dir/infer.txt          one line per utterance: to_text(ids) = the symbols up to the first </s>, joined with
                               --delimiter (infer.py:65-67 cuts at the first EOS SYMBOL; the best beam under --beam_width)
  model_dir/infer.dmp          joblib.dump([{'transcription': <that line>}, ...])
  model_dir/infer_targets.txt  the targets, joined with --delimiter (infer.py:269-271)

--plain_targets CSV (infer.py:246-271): rows `sound,lang,phrase` (tab-separated when the line holds a tab); the phrase is
split on blanks, stripped and lower-cased, zipped with the predictions in file order, and PER / Optimistic PER are the
text-level edit distances against it (infer.py:277-303,338-339; with --mapping on a binary-outputs model the target symbols
go through the 61->39 style map, infer.py:300-303).  WITHOUT --plain_targets the reference prints no PER at all; here the
labels stored in the TFRecord serve as targets (same files, same formulas) -- an addition, not a difference in the files.
Frame-level binary-feature accuracy (TIMIT markup) and text -> IPA conversion are out of scope.

Deliberate deviations in the PRINTED numbers (the files are the reference's):
  * Optimistic PER, greedy decode: the reference only lowers its per-utterance `min_err = 100000` inside the beam loop
    (infer.py:278-289), so without --beam_width it adds 100000 per utterance; here the minimum runs over the hypotheses
    that exist (one, in greedy mode: Optimistic PER = PER).  In its beam loop the reference scores the beams against `t`
    BEFORE this utterance's `t` is assigned (infer.py:286 vs :296: the previous utterance's target); here against this one's.
  * blank lines of --plain_targets are skipped; the reference's split(',') would raise on them."""
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
    """infer.py:65-67: the symbols of the ids up to (not including) the first </s> SYMBOL, joined with the delimiter."""
    from phones_las_amd.utils import EOS
    sym_list = [vocab_list[x] for x in sample_ids] + [EOS]
    return delimiter.join(sym_list[:sym_list.index(EOS)])


def main(args):
    if args.calc_frame_binf_accuracy or args.convert_targets_to_ipa:
        raise SystemExit('--calc_frame_binf_accuracy / --convert_targets_to_ipa (TIMIT markup and IPA analysis) are not on the HIP path')
    from phones_las_amd import utils
    from phones_las_amd import model_helper as mh
    from phones_las_amd.utils.metrics_utils import _levenshtein
    from train import load_checkpoint, to_device

    vocab_list = utils.load_vocab(args.vocab)
    vocab_list_orig = vocab_list
    args_h = argparse.Namespace(model_dir=args.model_dir, mapping=args.mapping)
    hparams = utils.create_hparams(args_h)              # requires an existing hparams.json (params_utils.py:96-97)
    hparams.decoder.set_hparam('beam_width', args.beam_width)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    binf2phone_np = None
    text_mapping = None                      # the reference's local `mapping` (infer.py:202-207): binary-outputs models only
    if hparams.decoder.binary_outputs:       # infer.py:203-210 of the reference
        if args.mapping is not None:
            vocab_list, mapping_list = utils.get_mapping(args.mapping, args.vocab)
            text_mapping = mapping_list
            hparams.del_hparam('mapping')
            hparams.add_hparam('mapping', mapping_list)
        binf2phone_np = utils.load_binf2phone(args.binf_map, vocab_list).values
    model = mh.LasModel(hparams, binf2phone=binf2phone_np)
    load_checkpoint(model, os.path.join(args.model_dir, 'checkpoint.pt'))

    mapping = hparams.mapping
    batches = utils.input_fn(args.data, args.vocab, args.norm, num_channels=args.num_channels,
                             batch_size=args.batch_size, num_epochs=1, take=args.take, is_infer=True)
    # per utterance: the beams' id lists (greedy: one), uncut; and the TFRecord's own labels
    beams_all, record_targets = [], []
    for features, labels in batches:
        f, _ = to_device(features, None, dev)
        pred = model.predict(f, transparent_projection=bool(args.use_phones_from_binf))   # transcribe_audio_file.py:90 couples the two
        # infer.py:223: the phones decoded from the binary-feature decoder, or the phone decoder's
        ids = pred['sample_ids_phones_binf' if args.use_phones_from_binf else 'sample_ids'].cpu().numpy()   # [B,T] / [B,T,K]
        for b in range(ids.shape[0]):
            beams_all.append([beam.tolist() for beam in (ids[b].T if ids.ndim == 3 else ids[b][None])])
            t = labels['targets_outputs'][b][:labels['target_sequence_length'][b] - 1].tolist()
            if mapping is not None:
                t = [mapping[x] for x in t]
                t = [x for x in t if x >= 0]
            record_targets.append([vocab_list[x] for x in t])
    model.check_device_status()          # a timed-out persistent kernel would have produced invalid hypotheses

    def text_of(ids):                    # infer.py:65-67 (as a symbol list)
        syms = [vocab_list[x] for x in ids] + [utils.EOS]
        return syms[:syms.index(utils.EOS)]

    if args.plain_targets:               # infer.py:246-268
        targets = []
        for line in open(args.plain_targets, 'r'):
            if not line.strip():
                continue
            cells = line.split('\t' if '\t' in line else ',')
            targets.append([x.strip().lower() for x in cells[2].split()])
    else:
        targets = record_targets
    err = tot = optimistic_err = 0
    for beams, t in zip(beams_all, targets):            # zip: the shorter of the two, as the reference's loop (infer.py:276)
        texts = [text_of(i) for i in beams]
        if args.plain_targets and text_mapping is not None:
            # infer.py:300-303: symbols of the original vocabulary through the id map (infer_targets.txt keeps the raw phrase)
            t = [vocab_list[text_mapping[vocab_list_orig.index(x)]] for x in t]
        err += _levenshtein(texts[0], t)                # the best-scoring beam is the hypothesis (infer.py:286-288)
        optimistic_err += min(_levenshtein(x, t) for x in texts)
        tot += len(t)
    with open(os.path.join(args.model_dir, 'infer_targets.txt'), 'w') as f:
        f.write('\n'.join(args.delimiter.join(t) for t in targets))
    predictions = [{'transcription': args.delimiter.join(text_of(beams[0]))} for beams in beams_all]      # infer.py:345-353
    with open(os.path.join(args.model_dir, 'infer.txt'), 'w') as f:
        f.write('\n'.join(p['transcription'] for p in predictions))
    from joblib import dump
    dump(predictions, os.path.join(args.model_dir, 'infer.dmp'))                                           # infer.py:358-359
    per = 100.0 * err / max(tot, 1)
    print('PER: %2.2f%%' % per)
    if args.beam_width > 0 or args.plain_targets:        # infer.py:338-339 prints both whenever it prints
        print('Optimistic PER: %2.2f%%' % (100.0 * optimistic_err / max(tot, 1)))
    return per


if __name__ == '__main__':
    main(parse_args())
