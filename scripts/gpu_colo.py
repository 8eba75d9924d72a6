# diagnostics: how many cooperative recurrent workgroups saw their whole group on one XCD
import sys, torch
sys.path.insert(0, '.')
from phones_las_amd.las import ops
from tests.helpers import make_hparams, make_batch, to_device
from phones_las_amd import model_helper as mh
for H, B in ((256, 64), (512, 64), (256, 48), (256, 16)):
    ohp, params = make_hparams(F=40, L=3, H=H, Hd=256, V=64)
    model = mh.LasModel(params)
    feats, labels = to_device(make_batch(B=B, T=96, F=40, V=64, U=8))
    for it in range(3):
        model.train_step(feats, labels)
        torch.cuda.synchronize()
        ws = ops.lstm_workspace(B, H, 2)[:16].view(torch.int32).tolist()
        print(H, B, 'status/local/fabric', ws[:3])
