# do rows 8..15 of the one-launch decoders' MFMA tiles leak into rows 0..7?  Those lanes read 16 bytes of the
# workspace header (words 4..7, "never written"): fill exactly those bytes with NaN / Inf / huge bf16 patterns and compare a
# train step's logits and gradients bit for bit with the clean run (training is deterministic).  POISON=fwd|bwd|both.
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_hparams, make_batch, to_device
from oracle import las_oracle as O
from phones_las_amd import model_helper as mh

def run(att, H, B, pattern, which):
    ohp, params = make_hparams(att=att, H=H, F=13, L=2)
    op = O.init_params(ohp, bias_scale=0.1)
    model = mh.LasModel(params)
    model.load_variables(op)
    src = [24 - (i * 5) % 17 for i in range(B)]
    tgt = [6 - i % 4 for i in range(B)]
    feats, labels = to_device(make_batch(B=B, T=24, src_len=src, tgt_len=tgt))
    outs = []
    for poison in (False, True):
        model.vars.grad.zero_()
        if poison:
            for k, ws in model.speller._persist_cache.items():
                if which == 'both' or k == which:
                    ws[16:32].view(torch.int16).fill_(pattern)
            torch.cuda.synchronize()
        loss, logits, dlogits = model.forward_train(feats, labels)
        model.backward(dlogits)
        torch.cuda.synchronize()
        outs.append((logits.clone(), model.vars.grad.clone()))
        if not poison:           # first (clean) run created the workspaces
            pass
    for k, ws in model.speller._persist_cache.items():
        ws[16:32].zero_()
    same_l = torch.equal(outs[0][0], outs[1][0])
    g0, g1 = outs[0][1], outs[1][1]
    same_g = torch.equal(g0, g1)
    nf = int((~torch.isfinite(g1)).sum())
    md = float((g0 - g1).abs().nan_to_num(1e30).max())
    return same_l, same_g, nf, md

pats = {'nan': 0x7fc0, 'inf': 0x7f80, 'big': 0x7f7f, 'one': 0x3f80}
for att in ('luong', 'bahdanau'):
    for H in (128, 256):
        for B in (5, 16):
            for which in os.environ.get('POISON', 'fwd,bwd').split(','):
                for name, pat in pats.items():
                    r = run(att, H, B, pat if pat < 0x8000 else pat - 0x10000, which)
                    print('%-9s H=%d B=%-2d poison %-4s %-4s: logits identical %s, grads identical %s, non-finite grads %d, max diff %.3g'
                          % (att, H, B, which, name, *r), flush=True)
