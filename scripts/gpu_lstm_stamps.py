# Phase times of the recurrent kernels (workgroup 0, first 256 steps) from the diagnostics build (see gpu_dec_stamps.py)
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
from phones_las_amd.las import ops
lib = hip.lib()
H = int(os.environ.get('H', 256)); B = 64; nd = 2; T = 800
torch.manual_seed(0)
xproj = torch.randn(B, T, nd * 4 * H, device='cuda') * 0.5
kh = torch.randn(nd, H, 4 * H, device='cuda') * 0.05
wp = torch.empty(nd * H * 4 * H, dtype=torch.bfloat16, device='cuda')
for d in range(nd):
    hip.check(lib.las_lstm_pack_recurrent(hip.p(kh[d]), H, hip.p(wp[d * H * 4 * H:]), hip.stream()))
khb = kh.view(nd, H, 4, H).transpose(2, 3).reshape(nd, H, 4 * H).to(torch.bfloat16).contiguous()
length = torch.full((B,), T, dtype=torch.int32, device='cuda')
y = torch.empty(B, T, nd * H, dtype=torch.bfloat16, device='cuda')
cbuf = torch.empty(B, T, nd * H, device='cuda')
cl = torch.empty(nd, B, H, device='cuda'); hl = torch.empty(nd, B, H, device='cuda')
dy = torch.randn(B, T, nd * H, device='cuda') * 0.1
dz = torch.empty(B, T, nd * 4 * H, dtype=torch.bfloat16, device='cuda')
ws = ops.lstm_workspace(B, H, nd)
for it in range(2):
    hip.check(lib.las_lstm_recurrent_fwd(hip.p(xproj), hip.p(wp), hip.p(length), hip.p(y), hip.p(cbuf), hip.p(cl), hip.p(hl),
                                         hip.p(ws), B, T, H, nd, hip.stream()))
    hip.check(lib.las_lstm_recurrent_bwd(hip.p(xproj), hip.p(cbuf), hip.p(dy), None, None, hip.p(khb), hip.p(length), hip.p(dz),
                                         hip.p(ws), B, T, H, nd, hip.stream()))
torch.cuda.synchronize()
lib.las_debug_read_lstm_stamps.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(2 * 256 * 8, np.uint64)
hip.check(lib.las_debug_read_lstm_stamps(buf.ctypes.data, buf.size))
NAMES = {'forward': ['request x.K_x; poll the peers\' h', 'previous stores + LDS barrier', 'MFMA issue (+ K-split hand-over)', 'wait for product + x; gates; h granules',
                     'HBM stores issued'],
         'backward': ['poll the partial dh', 'request next step\'s saved values (8-wave form: the helpers do)', 'gate derivatives -> dz (LDS [+ HBM])', 'LDS barrier',
                      'product + partial-tile granules', 'coefficients of the next step (8-wave form: the helpers do)']}
for which, off in (('forward', 0), ('backward', 256)):
    st = buf.reshape(512, 8)[off:off + 256].astype(np.int64)
    names = NAMES[which]
    d = np.diff(st[:, :len(names) + 1], axis=1) / 100.0
    print('%s (H=%d): step time %.3f us' % (which, H, np.diff(st[20:250, 0]).mean() / 100.0))
    for k, n in enumerate(names):
        print('  %-44s %6.3f us  (min %5.2f max %5.2f)' % (n, d[20:250, k].mean(), d[20:250, k].min(), d[20:250, k].max()))
