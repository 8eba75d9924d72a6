# Phase times of the one-launch decoder (workgroup 0: group 0, member 0 = utterance 0, part 0) from the diagnostics build:
#   LAS_CXXFLAGS=-DLAS_STAMPS LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python phones-las_amd/build.py --force
#   LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_dec_stamps.py
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from phones_las_amd import hip, model_helper as mh
c = bench.CONFIGS[os.environ.get('CFG', 'metric-M')]
model = mh.LasModel(bench.build_params(c), binf2phone=bench.binf_matrix(c['binf']) if c.get('binf') else None)
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
NAMES = {'forward': 'G:operands+MFMA,G:LDS reduce+z granules,S:wait for z+cell,S:sync,S:own scores,S:gather scores,S:softmax,'
                    'S:context partial,S:reduce+stores,barrier',
         'backward': 'top: saved-value requests + wait for d(feed),S1:d(context) row,S1:d(align) over own frames,S1:dot reduce + granule,'
                     'S2:wait for the dots,S2:ds,S2:dh over own keys,S3:partials + cell (part 0),barrier,G:operands+MFMA,G:granules'}
for which, off in (('forward', 0), ('backward', 256)):
    st = buf.reshape(512, 16)[off:off + U].astype(np.int64)
    names = NAMES[which].split(',')
    print('%s: step time (stamp 0 to stamp 0): %.2f us' % (which, np.diff(st[:, 0]).mean() / 100.0))
    # A stamp slot this launch form never writes stays zero, and a difference against it is garbage (round 5 kept a profile with a
    # phase of -9.6e11 us).  Phases are the differences of consecutive WRITTEN slots; a slot written in some steps only, or
    # stamps that are not in time order, end the script with an error instead of a number.
    cols = st[5:-1, :len(names) + 1]
    written = [k for k in range(cols.shape[1]) if (cols[:, k] != 0).all()]
    partly = [k for k in range(cols.shape[1]) if k not in written and (cols[:, k] != 0).any()]
    if partly or len(written) < 2:
        sys.exit('%s: stamp slots %s were written in some steps only (or fewer than two slots at all)' % (which, partly))
    d = np.diff(cols[:, written], axis=1) / 100.0            # 100 MHz ticks -> us
    if (d < 0).any() or (d > 1e4).any():
        sys.exit('%s: stamps out of time order or more than 10 ms apart' % which)
    for i in range(d.shape[1]):
        k = written[i]
        label = names[k] if written[i + 1] == k + 1 else '%s .. %s' % (names[k], names[written[i + 1] - 1])
        print('  %-44s %6.2f us  (min %5.2f max %5.2f)' % (label, d[:, i].mean(), d[:, i].min(), d[:, i].max()))
st = buf.reshape(512, 16)[0:U].astype(np.int64)
if st[5:, 11].any():
    print('forward G role: operands + resident products %.2f us, streamed chunks %.2f us, partial sums to LDS %.2f us' % (
        ((st[5:-1, 11] - st[5:-1, 0]) / 100.0).mean(), ((st[5:-1, 12] - st[5:-1, 11]) / 100.0).mean(), ((st[5:-1, 1] - st[5:-1, 12]) / 100.0).mean()))
loc = buf.reshape(512, 16)[255]
print('lean forward kernel, sampled workgroups whose group sits on one XCD: %d, spread over XCDs: %d (counts over all launches)' % (int(loc[15]), int(loc[14])))
