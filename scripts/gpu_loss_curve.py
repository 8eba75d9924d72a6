# Loss of `LasModel.train_step` on bench.py's fixed synthetic batch every EVERY steps:  python scripts/gpu_loss_curve.py [config] [steps] [every]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from phones_las_amd import model_helper as mh

cfg = sys.argv[1] if len(sys.argv) > 1 else 'metric-M-ragged'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 250
c = bench.CONFIGS[cfg]
dev = torch.device('cuda', 0)
model = mh.LasModel(bench.build_params(c), binf2phone=bench.binf_matrix(c['binf']) if c.get('binf') else None)
feats, labels = bench.synthetic_batch(c, 1234, dev)
feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])
out = []
for s in range(steps):
    l = model.train_step(feats, labels, num_steps=c['U'])
    if s % every == 0 or s == steps - 1:
        out.append((s, l))
model.check_device_status()
print(cfg, ' '.join('%d:%.3f' % (s, float(l)) for s, l in out))
