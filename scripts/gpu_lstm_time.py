# times the recurrent forward/backward kernels alone (layer shapes of metric-M); LAS_LSTM_PREFETCH selects the mode
import os, sys, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
from phones_las_amd.las import ops
lib = hip.lib()
H = int(os.environ.get('H', 256)); B = 64; nd = 2
HEAT = int(os.environ.get('HEAT', 0))      # HEAT=n: n dense bf16 products (8192^3, ~1 ms each) in front of every timed launch
FLUSH = int(os.environ.get('FLUSH', 0))    # FLUSH=1: a 1-GiB buffer is rewritten in front of every timed launch (L2 / MALL hold nothing of the operands)
big = torch.empty(1 << 30, dtype=torch.uint8, device='cuda') if FLUSH else None
if HEAT:
    ha = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16); hb = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
for T in (800, 400, 200):
    torch.manual_seed(0)
    xproj = torch.randn(B, T, nd * 4 * H, device='cuda') * 0.5
    x0 = xproj.clone()
    kh = torch.randn(nd, H, 4 * H, device='cuda') * 0.05
    wp = torch.empty(nd * H * 4 * H, dtype=torch.bfloat16, device='cuda')
    for d in range(nd):
        hip.check(lib.las_lstm_pack_recurrent(hip.p(kh[d]), H, hip.p(wp[d * H * 4 * H:]), hip.stream()))
    khb = kh.view(nd, H, 4, H).transpose(2, 3).reshape(nd, H, 4 * H).to(torch.bfloat16).contiguous()   # columns u*4+g
    length = torch.full((B,), T, dtype=torch.int32, device='cuda')
    y = torch.empty(B, T, nd * H, dtype=torch.bfloat16, device='cuda')
    cbuf = torch.empty(B, T, nd * H, device='cuda')
    cl = torch.empty(nd, B, H, device='cuda'); hl = torch.empty(nd, B, H, device='cuda')
    dy = torch.randn(B, T, nd * H, device='cuda') * 0.1
    dz = torch.empty(B, T, nd * 4 * H, dtype=torch.bfloat16, device='cuda')
    ws = ops.lstm_workspace(B, H, nd)
    tf, tb = [], []
    for it in range(6):
        xproj.copy_(x0)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for _ in range(HEAT):
            ha @ hb
        if FLUSH:
            big.add_(1)
        e[0].record()
        hip.check(lib.las_lstm_recurrent_fwd(hip.p(xproj), hip.p(wp), hip.p(length), hip.p(y), hip.p(cbuf), hip.p(cl), hip.p(hl),
                                             hip.p(ws), B, T, H, nd, hip.stream()))
        e[1].record()
        for _ in range(HEAT):
            ha @ hb
        if FLUSH:
            big.add_(1)
        e[3].record()
        hip.check(lib.las_lstm_recurrent_bwd(hip.p(xproj), hip.p(cbuf), hip.p(dy), None, None, hip.p(khb), hip.p(length), hip.p(dz),
                                             hip.p(ws), B, T, H, nd, hip.stream()))
        e[2].record(); torch.cuda.synchronize()
        ops.check_lstm_status(B, H, nd)
        tf.append(e[0].elapsed_time(e[1])); tb.append(e[3].elapsed_time(e[2]))
    print('H %d T %d mode %s: fwd %.3f ms (%.2f us/step)  bwd %.3f ms (%.2f us/step)  checksum %.6f %.6f' % (
        H, T, os.environ.get('LAS_LSTM_PREFETCH', 'default'), min(tf), min(tf) * 1e3 / T, min(tb), min(tb) * 1e3 / T,
        float(y.float().abs().mean()), float(dz.float().abs().mean())))
