# How many chain members found their whole group on ONE XCD (workspace header words [1] local / [2] fabric flavour), in a real train
# step (eager launches) and in the stand-alone timing loop?  usage: python scripts/gpu_colo_count.py [config]
import sys, torch
sys.path.insert(0, '.')
import bench
from phones_las_amd import model_helper as mh
from phones_las_amd.las import ops
cfg = sys.argv[1] if len(sys.argv) > 1 else 'metric-M'
c = bench.CONFIGS[cfg]
model = mh.LasModel(bench.build_params(c))
feats, labels = bench.synthetic_batch(c, 1234, torch.device('cuda', 0))
feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])
for _ in range(3):
    model.train_step(feats, labels, num_steps=c['U'])
torch.cuda.synchronize()
ws = ops.lstm_workspace(c['B'], c['H'], 2)
def words():
    return [int(x) for x in ws[:32].view(torch.int32)[:3].tolist()]
w0 = words()
for i in range(5):
    model.train_step(feats, labels, num_steps=c['U'])
    torch.cuda.synchronize()
    w1 = words()
    print('%s step %d: status %d, members on one XCD with their group: %d, spread: %d' % (cfg, i, w1[0], w1[1] - w0[1], w1[2] - w0[2]))
    w0 = w1
