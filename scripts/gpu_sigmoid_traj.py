# loss trajectory of the sigmoid-decoder training used by tests/test_gpu_binary_decoders.py (diagnostic)
import sys
sys.path.insert(0, '.')
import torch
from tests.helpers import make_batch, to_device
from tests.test_gpu_binary_decoders import _toy_binf, _models
binf = _toy_binf(8, 11)
O, ohp, op, model = _models(binf=binf, sigmoid=True, att='luong', lr=1e-2)
src_len, tgt_len = [24, 9, 17, 24, 12], [6, 4, 5, 6, 3]
batch = make_batch(B=5, T=24, src_len=src_len, tgt_len=tgt_len)
feats, labels = to_device(batch)
out = []
for i in range(300):
    out.append(float(model.train_step(feats, labels)))
print(' '.join('%.3f' % v for v in out[::10]))
# gradient parity at the weights reached after 100 more steps from a fresh model
from tests.helpers import relerr
O, ohp, op, model = _models(binf=binf, sigmoid=True, att='luong', lr=1e-2)
for i in range(100):
    model.train_step(feats, labels)
trained = {n: t.detach().double().cpu() for n, t in model.vars.params.items()}
model.vars.grad.zero_()
loss, logits, dl = model.forward_train(feats, labels)
model.backward(dl)
torch.cuda.synchronize()
ref = O.train_step(ohp, trained, None, None, 1, batch, mxu='bf16')
print('loss dev %.5f oracle %.5f' % (float(loss), float(ref['audio_loss'])))
worst = max((relerr(model.vars.grads[n], ref['grads'][n] - ohp.l2_reg_scale * trained[n]), n) for n, _, _ in model.vars.table)
print('worst gradient relerr', worst)
