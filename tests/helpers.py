"""Shared helpers for the parity tests: build matching oracle / product hyper-parameters and batches."""
import numpy as np
import torch

from oracle import las_oracle as O


def make_hparams(F=13, L=2, H=64, Hd=None, V=11, att='luong', dec_layers=1, bottom_only=True, pass_hidden=True,
                 unidirectional=False, lr=1e-3, l2=1e-6, pyramidal=True, ctc=-1.0, als=None, emb=0, binf=None,
                 binf_reg=1.0, sigmoid=False, multitask=False, sampling=0.0, binf_trainable=False):
    """Returns (oracle HP, product params) describing the same model.  binf: the [nf, V] feature map of a binary decoder:
    --binf_projection unless sigmoid=True (--binary_outputs alone: feature-logit outputs); multitask adds the phone
    decoder in front of it."""
    Hd = Hd or H
    ohp = O.HP(encoder=O.EncoderHP(num_layers=L, num_units=H, unidirectional=unidirectional, use_pyramidal=pyramidal),
               num_channels=F,
               decoder=O.DecoderHP(num_layers=dec_layers, num_units=Hd, target_vocab_size=V, attention_type=att,
                                   bottom_only=bottom_only, pass_hidden_state=pass_hidden, attention_layer_size=als,
                                   embedding_size=emb, binf_projection=binf is not None and not sigmoid,
                                   binary_outputs=binf is not None, multitask=multitask,
                                   binf_count=0 if binf is None else int(binf.shape[0]), binf_map=binf,
                                   binf_projection_reg_weight=binf_reg, sampling_probability=sampling,
                                   binf_trainable=binf_trainable),
               learning_rate=lr, l2_reg_scale=l2, ctc_weight=ctc)
    from phones_las_amd.utils import params_utils as pu
    hp = pu.get_default_hparams()
    for k, v in dict(num_channels=F, encoder_layers=L, encoder_units=H, use_pyramidal=pyramidal,
                     unidirectional=unidirectional, decoder_layers=dec_layers, decoder_units=Hd,
                     target_vocab_size=V, attention_type=att, bottom_only=bottom_only, pass_hidden_state=pass_hidden,
                     dropout=0.0, sampling_probability=sampling, learning_rate=lr, l2_reg_scale=l2, ctc_weight=ctc,
                     attention_layer_size=als, embedding_size=emb).items():
        hp.set_hparam(k, v)
    if binf is not None:             # --binary_outputs --binf_projection --binf_map (cfg5)
        for k, v in dict(binary_outputs=True, binf_projection=not sigmoid, binf_count=int(binf.shape[0]),
                         binf_projection_reg_weight=binf_reg, multitask=multitask, binf_trainable=binf_trainable).items():
            hp.set_hparam(k, v)
    return ohp, pu.get_encoder_decoder_hparams(hp)


def make_batch(B=3, T=12, F=13, V=11, U=6, src_len=None, tgt_len=None, seed=0):
    b = O.synthetic_batch(B, T, F, V, U, ragged=False, seed=seed)
    if src_len is not None:
        b['source_sequence_length'] = torch.tensor(src_len)
        for i, n in enumerate(src_len):
            b['encoder_inputs'][i, n:] = 0
    if tgt_len is not None:
        b['target_sequence_length'] = torch.tensor(tgt_len)
        for i, n in enumerate(tgt_len):          # keep <s> y.. / y.. </s> consistent with the new length
            b['targets_outputs'][i, n - 1:] = O.EOS_ID
            b['targets_inputs'][i, n:] = O.EOS_ID
    return b


def to_device(batch):
    feats = {'encoder_inputs': batch['encoder_inputs'].float().cuda(),
             'source_sequence_length': batch['source_sequence_length'].to(torch.int32).cuda()}
    labels = {'targets_inputs': batch['targets_inputs'].to(torch.int32).cuda(),
              'targets_outputs': batch['targets_outputs'].to(torch.int32).cuda(),
              'target_sequence_length': batch['target_sequence_length'].to(torch.int32).cuda()}
    return feats, labels


def relerr(got, ref):
    ref = ref.detach().double()
    return float((got.detach().double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-12))
