import os, sys, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
from phones_las_amd.las import ops
lib = hip.lib()
H, nd = 256, 2
B, T = int(os.environ.get('B', 2)), int(os.environ.get('T', 6))
lens = [int(x) for x in os.environ.get('LENS', '6,3').split(',')]
torch.manual_seed(0)
xproj = torch.randn(B, T, nd * 4 * H, device='cuda') * 0.5
kh = torch.randn(nd, H, 4 * H, device='cuda') * 0.05
wp = torch.empty(nd * H * 4 * H, dtype=torch.bfloat16, device='cuda')
for d in range(nd):
    hip.check(lib.las_lstm_pack_recurrent(hip.p(kh[d]), H, hip.p(wp[d * H * 4 * H:]), hip.stream()))
length = torch.tensor(lens, dtype=torch.int32, device='cuda')
y = torch.empty(B, T, nd * H, dtype=torch.bfloat16, device='cuda')
cbuf = torch.empty(B, T, nd * H, device='cuda'); cl = torch.empty(nd, B, H, device='cuda'); hl = torch.empty(nd, B, H, device='cuda')
ws = ops.lstm_workspace(B, H, nd)
print('launch', B, T, lens, flush=True)
hip.check(lib.las_lstm_recurrent_fwd(hip.p(xproj), hip.p(wp), hip.p(length), hip.p(y), hip.p(cbuf), hip.p(cl), hip.p(hl), hip.p(ws), B, T, H, nd, hip.stream()))
torch.cuda.synchronize()
print('ok status', ws[:16].view(torch.int32).tolist(), float(y.float().abs().mean()))
