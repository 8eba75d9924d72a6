# Phase times of the two-cell one-launch decoders (workgroup 0) from the diagnostics build (see gpu_dec_stamps.py): the stamps of a
# step in time order, forward and backward.
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from phones_las_amd import hip, model_helper as mh
c = bench.CONFIGS[os.environ.get('CFG', 'default-arch')]
model = mh.LasModel(bench.build_params(c))
feats, labels = bench.synthetic_batch(c, 1234, 'cuda')
feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])
for _ in range(3):
    model.train_step(feats, labels, num_steps=c['U'])
torch.cuda.synchronize()
lib = hip.lib()
lib.las_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(2 * 256 * 16, np.uint64)
hip.check(lib.las_debug_read_stamps(buf.ctypes.data, buf.size))
U = c['U']
FWD = {0: 'top', 11: 'G0 resident products', 12: 'G0 streamed', 1: 'G0 partials to LDS', 2: 'G0 granules', 13: 'cell-0 body done', 14: 'barrier', 15: 'G1 done',
       3: 'cell done (attention body)', 4: 'sync', 5: 'own scores', 6: 'gather scores', 7: 'softmax', 8: 'context partial', 9: 'reduce+stores', 10: 'end barrier'}
for which, off, names in (('forward', 0, FWD), ('backward', 256, None)):
    st = buf.reshape(512, 16)[off:off + U].astype(np.int64)
    print('%s: step time %.2f us' % (which, np.diff(st[:, 0]).mean() / 100.0))
    rel = (st[5:-1] - st[5:-1, :1]) / 100.0
    mean = rel.mean(0)
    order = [k for k in np.argsort(mean) if st[5:-1, k].any()]
    prev = 0.0
    for k in order:
        print('  stamp %2d  at %7.2f us  (+%6.2f)  %s' % (k, mean[k], mean[k] - prev, names.get(k, '') if names else ''))
        prev = mean[k]
