# Per-launch durations of the recurrent kernels inside real train steps (HIP events around each launch, eager launches), next to
# the stand-alone loop of scripts/gpu_lstm_time.py.  usage: [LAS_SERIAL=1 LAS_LSTM_STREAM=0] python scripts/gpu_chain_insitu.py [config] [steps]
import sys, torch
sys.path.insert(0, '.')
import bench
from phones_las_amd import hip, model_helper as mh
cfg = sys.argv[1] if len(sys.argv) > 1 else 'metric-M'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
c = bench.CONFIGS[cfg]
model = mh.LasModel(bench.build_params(c))
feats, labels = bench.synthetic_batch(c, 1234, torch.device('cuda', 0))
feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])
for _ in range(3):
    model.train_step(feats, labels, num_steps=c['U'])
torch.cuda.synchronize()
for s in range(steps):
    with hip.KernelTimer() as kt:
        model.train_step(feats, labels, num_steps=c['U'])
        torch.cuda.synchronize()
        line = ['%s %.3f' % (n, e0.elapsed_time(e1)) for n, _, e0, e1 in kt.records if n.startswith('lstm_')]
    print('%s step %d: %s' % (cfg, s, '  '.join(line)))
