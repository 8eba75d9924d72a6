"""Stand-alone evaluation of a trained model on one TFRecord set — the drop-in of the reference's eval.py:10-73
(`Estimator.evaluate`: streaming mean of the EVAL loss and of the normalised edit distance) on the HIP path.

    python eval.py --data test.tfr --vocab vocab.txt --norm norm.dmp --model_dir model/ [--mapping ...]
"""
import argparse
import os

import torch


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Run model evaluation.')
    p.add_argument('--data', type=str, help='data in TFRecord format')
    p.add_argument('--vocab', type=str, help='vocabulary table, listing vocabulary line by line')
    p.add_argument('--norm', type=str, default=None, help='normalization params')
    p.add_argument('--t2t_format', action='store_true')
    p.add_argument('--t2t_problem_name', type=str)
    p.add_argument('--mapping', type=str, help='additional mapping when evaluation')
    p.add_argument('--model_dir', type=str, required=True, help='path of saving model')
    p.add_argument('--batch_size', type=int, default=8)
    p.add_argument('--num_channels', type=int, default=39)
    p.add_argument('--binf_map', type=str, default='misc/binf_map.csv')
    p.add_argument('--t2t_features_hparams_override', type=str, default='')
    return p.parse_args(argv)


def main(args):
    if args.t2t_format:
        raise SystemExit('--t2t_format is a TensorFlow-only input option and is not supported')
    from phones_las_amd import utils
    from phones_las_amd import model_helper as mh
    from train import load_checkpoint, evaluate

    hparams = utils.create_hparams(args)              # requires an existing hparams.json (params_utils.py:96-97)
    vocab_list = utils.load_vocab(args.vocab)
    binf2phone_np = None
    if hparams.decoder.binary_outputs:                # eval.py:46-49
        binf2phone_np = utils.load_binf2phone(args.binf_map, vocab_list).values
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    model = mh.LasModel(hparams, binf2phone=binf2phone_np)
    load_checkpoint(model, os.path.join(args.model_dir, 'checkpoint.pt'))
    eval_name = str(os.path.basename(args.data).split('.')[0])
    print('Evaluating on {}'.format(eval_name))
    # is_infer=True: file order, last partial batch kept (same metrics as train.py's in-loop evaluation and infer.py)
    batches = utils.input_fn(args.data, args.vocab, args.norm, num_channels=args.num_channels, batch_size=args.batch_size,
                             is_infer=True)
    return evaluate(model, batches, dev)


if __name__ == '__main__':
    main(parse_args())
