# timing ablations of the recurrent forward kernel (diagnostics build, -DLAS_ABLATE); results are wrong by design
import os, sys, ctypes as C, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
from phones_las_amd.las import ops
lib = hip.lib()
H, B, nd = 256, 64, 2
T = int(os.environ.get('T', 200))
xproj = torch.randn(B, T, nd * 4 * H, device='cuda') * 0.5
x0 = xproj.clone()
kh = torch.randn(nd, H, 4 * H, device='cuda') * 0.05
wp = torch.empty(nd * H * 4 * H, dtype=torch.bfloat16, device='cuda')
for d in range(nd):
    hip.check(lib.las_lstm_pack_recurrent(hip.p(kh[d]), H, hip.p(wp[d * H * 4 * H:]), hip.stream()))
length = torch.full((B,), T, dtype=torch.int32, device='cuda')
y = torch.empty(B, T, nd * H, dtype=torch.bfloat16, device='cuda')
cbuf = torch.empty(B, T, nd * H, device='cuda')
cl = torch.empty(nd, B, H, device='cuda'); hl = torch.empty(nd, B, H, device='cuda')
ws = ops.lstm_workspace(B, H, nd)
names = {1: 'no HBM stores', 2: 'no xproj loads', 4: 'no gate math', 8: 'no MFMA', 16: 'no polls', 32: 'no LDS scatter', 64: 'no granule stores', 128: 'no progress store', 256: 'no h LDS write', 512: 'no barrier', 1024: 'no fail check'}
for bits in (0, 2047, 128, 128 + 1, 128 + 16, 128 + 8, 128 + 4, 128 + 31):
    lib.las_lstm_debug_ablate(bits)
    ts = []
    for it in range(3):
        xproj.copy_(x0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(lib.las_lstm_recurrent_fwd(hip.p(xproj), hip.p(wp), hip.p(length), hip.p(y), hip.p(cbuf), hip.p(cl), hip.p(hl),
                                             hip.p(ws), B, T, H, nd, hip.stream()))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print('%-50s %.2f us/step' % ('bits %d' % bits, min(ts) * 1e3 / T))
