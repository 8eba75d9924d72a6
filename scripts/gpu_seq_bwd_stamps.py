# Phase times of the sequential one-launch backward decoder (dec_seq_bwd_kernel, workgroup 0) from the DIAGNOSTICS build:
#   LAS_CXXFLAGS=-DLAS_STAMPS LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python phones-las_amd/build.py --force
#   LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_seq_bwd_stamps.py      (CFG=cfg5)
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from phones_las_amd import hip, model_helper as mh
c = bench.CONFIGS[os.environ.get('CFG', 'cfg5')]
model = mh.LasModel(bench.build_params(c), binf2phone=bench.binf_matrix(c['binf']) if c.get('binf') else None)
feats, labels = bench.synthetic_batch(c, 1234, torch.device('cuda', 0))
feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])
for _ in range(2):
    model.vars.grad.zero_()
    _, _, dl = model.forward_train(feats, labels, num_steps=c['U'])
    model.backward(dl)
torch.cuda.synchronize()
lib = hip.lib()
lib.las_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(2 * 256 * 16, np.uint64)
hip.check(lib.las_debug_read_stamps(buf.ctypes.data, buf.size))
st = buf.reshape(512, 16)[256:256 + c['U']].astype(np.int64)
# stamp k -> k+1 intervals of one step (stamp 7 is unused in the fused body)
names = ['d(attention) + d[query|context] = datt W_al^T', 'body: d(context) total', 'body: d(alignments) (values W_c resident in LDS)',
         'body: normaliser backward', 'body: query path (keys, d(keys))', 'body: reduce + Wq^T', 'body: cell backward',
         'd(feed) = dz K^T (MFMA)']
order = [0, 1, 2, 3, 4, 5, 6, 8, 9]
rows = st[5:c['U'] - 5]
print('step time %.2f us' % (np.diff(st[5:c['U'] - 5, 0]).mean() / -100.0))
for k in range(len(order) - 1):
    d = (rows[:, order[k + 1]] - rows[:, order[k]]) / 100.0
    print('  %-52s %7.2f us' % (names[k], d.mean()))
