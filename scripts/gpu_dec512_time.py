# forward / backward time of the fused decoder at the metric-L shape (Hd=512, M=2048, T'=100, U=80, B=64):
# LAS_DEC_PERSIST=1|0 (one launch / per-step launches), ATT=bahdanau|luong, CFG=metric-L|metric-M (256 units, T'=200), B=utterances
import os, sys, time, torch
sys.path.insert(0, '.')
import bench
from phones_las_amd import model_helper as mh
c = dict(bench.CONFIGS[os.environ.get('CFG', 'metric-L')], att=os.environ.get('ATT', 'bahdanau'))
if os.environ.get('B'):
    c['B'] = int(os.environ['B'])          # fewer utterances = fewer groups streaming their weight slices at once
model = mh.LasModel(bench.build_params(c))
feats, labels = bench.synthetic_batch(c, 1234, torch.device('cuda', 0))
feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])
(mem, ml), st = model.listener.forward(feats['encoder_inputs'], feats['source_sequence_length'], 'train')
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lg = model.speller.forward_train(mem, ml, st, labels['targets_inputs'], 80)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    dl = (torch.randn_like(lg) * 1e-3).to(torch.bfloat16)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    model.speller.backward(dl, model.vars.grads)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print('ATT=%s LAS_DEC_PERSIST=%s forward %.2f ms backward %.2f ms' % (c['att'], os.environ.get('LAS_DEC_PERSIST', '1'), (t1 - t0) * 1e3, (t3 - t2) * 1e3))
