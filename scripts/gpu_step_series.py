# per-step wall time of consecutive train steps (HIP events around each step): does the step time drift after start-up?
import sys, torch
sys.path.insert(0, '.')
import bench
from phones_las_amd import model_helper as mh
cfg = sys.argv[1] if len(sys.argv) > 1 else 'metric-M'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = bench.CONFIGS[cfg]
model = mh.LasModel(bench.build_params(c))
feats, labels = bench.synthetic_batch(c, 1234, torch.device('cuda', 0))
feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
ev[0].record()
for i in range(n):
    model.train_step(feats, labels, num_steps=c['U'])
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print(cfg, 'per-step ms:', ' '.join('%.3f' % x for x in ms))
print('mean of steps 5..: %.3f   10..: %.3f   20..: %.3f' % (sum(ms[5:]) / len(ms[5:]), sum(ms[10:]) / len(ms[10:]), sum(ms[20:]) / len(ms[20:])))
