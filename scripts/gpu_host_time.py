# host time to ENQUEUE one eager train step (no synchronisation inside) against the GPU time of the step: eager launches
# only pay off while the host stays well ahead of the GPU
import sys, time, torch
sys.path.insert(0, '.')
import bench
from phones_las_amd import model_helper as mh
c = bench.CONFIGS['metric-M']
dev = torch.device('cuda', 0)
model = mh.LasModel(bench.build_params(c))
feats, labels = bench.synthetic_batch(c, 1234, dev)
feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])
for _ in range(3):
    model.train_step(feats, labels, num_steps=c['U'])
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    model.train_step(feats, labels, num_steps=c['U'])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue %.2f ms per step; wall %.2f ms per step (GPU-bound when the first is well below the second)' % (
    (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
