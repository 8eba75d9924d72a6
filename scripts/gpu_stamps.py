# diagnostics: per-wave timeline of the cooperative recurrent forward kernel (library built with -DLAS_STAMPS)
#   LAS_HIP_LIB=phones-las_amd/liblas_hip_stamps.so LAS_CXXFLAGS=-DLAS_STAMPS python phones-las_amd/build.py
#   LAS_HIP_LIB=phones-las_amd/liblas_hip_stamps.so python scripts/gpu_stamps.py
import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, '.')
from phones_las_amd import hip
from phones_las_amd.las import ops
H, B, T, nd = 256, 64, 200, 2
lib = hip.lib()
stamps = torch.zeros(64 * 5 * 32 * 8, dtype=torch.int64, device='cuda')
lib.las_lstm_debug_stamps.argtypes = [C.c_void_p]
assert lib.las_lstm_debug_stamps(stamps.data_ptr()) == 0
torch.manual_seed(0)
xproj = torch.randn(B, T, nd * 4 * H, device='cuda') * 0.5
kh = torch.randn(nd, H, 4 * H, device='cuda') * 0.05
wp = torch.empty(nd * H * 4 * H, dtype=torch.bfloat16, device='cuda')
for d in range(nd):
    hip.check(lib.las_lstm_pack_recurrent(hip.p(kh[d]), H, hip.p(wp[d * H * 4 * H:]), hip.stream()))
length = torch.full((B,), T, dtype=torch.int32, device='cuda')
y = torch.empty(B, T, nd * H, dtype=torch.bfloat16, device='cuda')
cbuf = torch.empty(B, T, nd * H, device='cuda')
cl = torch.empty(nd, B, H, device='cuda'); hl = torch.empty(nd, B, H, device='cuda')
ws = ops.lstm_workspace(B, H, nd)
for it in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.check(lib.las_lstm_recurrent_fwd(hip.p(xproj), hip.p(wp), hip.p(length), hip.p(y), hip.p(cbuf), hip.p(cl), hip.p(hl),
                                         hip.p(ws), B, T, H, nd, hip.stream()))
    e1.record(); torch.cuda.synchronize()
print('us/step', e0.elapsed_time(e1) * 1e3 / T)
st = stamps.cpu().numpy().reshape(64, 5, 32, 8)
blocks = [b for b in range(64) if st[b, 0, 4, 0] != 0]
print('blocks with stamps', len(blocks))
names = ['xp issue', 'poll', 'barrier', 'lds+mfma', 'gate math', 'send', 'hbm stores', 'loop']
for b in blocks[:3] + blocks[-1:]:
    for w in range(4):
        a = st[b, w, 2:30].astype(np.float64)
        d = [np.mean(a[:, k + 1] - a[:, k]) for k in range(7)] + [np.mean(a[1:, 0] - a[:-1, 7])]
        step = np.mean(a[1:, 0] - a[:-1, 0])
        print('block %2d wave %d step %.0f ticks: ' % (b, w, step) + '  '.join('%s %.0f' % (n, x) for n, x in zip(names, d)))
