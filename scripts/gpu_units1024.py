# 1024-unit smoke: listener + speller vs oracle (small B, T)
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import make_hparams, make_batch, to_device, relerr
from oracle import las_oracle as O
from phones_las_amd import model_helper as mh
for kw in (dict(H=1024, Hd=256, L=2, pass_hidden=False), dict(H=1024, Hd=1024, L=2, pass_hidden=True), dict(H=1024, Hd=1024, L=2, pass_hidden=True, att='bahdanau', dec_layers=2)):
    ohp, params = make_hparams(**kw)
    op = O.init_params(ohp, bias_scale=0.1)
    model = mh.LasModel(params)
    model.load_variables({k: v for k, v in op.items()})
    batch = make_batch(B=5, T=14, src_len=[14, 7, 10, 3, 12], tgt_len=[6, 4, 5, 2, 6])
    feats, labels = to_device(batch)
    out = O.train_step(ohp, op, None, None, 1, batch, mxu='bf16')
    model.vars.grad.zero_()
    loss, logits, dlogits = model.forward_train(feats, labels)
    model.backward(dlogits)
    torch.cuda.synchronize()
    print(kw, 'status', model.read_and_clear_status(), 'loss', float(loss), float(out['aux']['ce']))
    V = ohp.decoder.target_vocab_size
    print('  logits relerr', max(float(relerr(logits[b, :n, :V], out['aux']['logits'][b, :n])) for b, n in enumerate([6, 4, 5, 2, 6])))
    worst = max((float(relerr(model.vars.grads[n], out['grads'][n] - ohp.l2_reg_scale * op[n])), n) for n, _, _ in model.vars.table)
    print('  worst grad relerr', worst)
