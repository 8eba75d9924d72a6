# gradient parity of a 512-unit listener + speller against the oracle (bf16 storage model and exact f64 model) with the recurrent
# backward's partial sums as fp32 granules or as bf16 pairs: LAS_LSTM_ROWS=8 [LAS_LSTM_BWD_PACK=1] python scripts/gpu_pack_parity.py
import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import make_hparams, make_batch, to_device, relerr
from oracle import las_oracle as O
from phones_las_amd import model_helper as mh
T = int(os.environ.get('T', 96)); B = int(os.environ.get('B', 16))
for kw in (dict(H=512, Hd=512, L=2, pass_hidden=True, att='bahdanau'), dict(H=512, Hd=256, L=3, pass_hidden=False, att='luong')):
    ohp, params = make_hparams(**kw)
    op = O.init_params(ohp, bias_scale=0.1)
    model = mh.LasModel(params)
    model.load_variables({k: v for k, v in op.items()})
    torch.manual_seed(3)
    src = [T - (7 * i) % (T // 2) for i in range(B)]; src[0] = T
    tgt = [6 + i % 5 for i in range(B)]
    batch = make_batch(B=B, T=T, src_len=src, tgt_len=tgt, U=max(tgt))
    feats, labels = to_device(batch)
    model.vars.grad.zero_()
    loss, logits, dlogits = model.forward_train(feats, labels)
    model.backward(dlogits)
    torch.cuda.synchronize()
    assert model.read_and_clear_status() == []
    for mxu in ('bf16', 'f64'):
        out = O.train_step(ohp, op, None, None, 1, batch, mxu=mxu)
        errs = sorted(((float(relerr(model.vars.grads[n], out['grads'][n] - ohp.l2_reg_scale * op[n])), n) for n, _, _ in model.vars.table), reverse=True)
        print(kw, 'PACK', os.environ.get('LAS_LSTM_BWD_PACK', '0'), 'oracle', mxu, 'loss', float(loss), float(out['aux']['ce']), 'worst grads:', [(round(e, 5), n.split('/')[1] + '/' + n.split('/')[-1]) for e, n in errs[:4]])
