# The fused LSTM weight-gradient product (las_gemm_tn_lstm, 128 x 512 tiles) alone at the metric-L layer shapes, for the library
# LAS_HIP_LIB names: the product build and the ablation builds of gemm_tn_ring_kernel (LAS_CXXFLAGS=-DTNR_NO_MFMA / -DTNR_NO_DSREAD /
# -DTNR_NO_GLDS): what the K loop waits for.  Results of the ablation builds are garbage by construction; only the time is read.
import os, sys, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
lib = hip.lib()
B, H = 64, 512
tag = os.environ.get('TAG', 'product')
for name, T, D in (('L1', 800, 1024), ('L2', 400, 2048), ('L3', 200, 2048), ('L0', 800, 40)):
    BT = B * T
    torch.manual_seed(0)
    x = (torch.randn(BT, D, device='cuda') * 0.5).to(torch.bfloat16)
    y = (torch.randn(BT, 2 * H, device='cuda') * 0.5).to(torch.bfloat16)
    dz = (torch.randn(BT, 8 * H, device='cuda') * 0.1).to(torch.bfloat16)
    gk = torch.zeros(D + H, 4 * H, device='cuda'); gb = torch.zeros(4 * H, device='cuda')
    tiles = -(-(D + H + 1) // 128) * -(-(4 * H) // 128)
    split = max(1, min(32, BT // 512, 2 * round(704 / tiles)))          # ops.py's choice for the wide tiles
    ws = torch.empty(lib.las_gemm_tn_lstm_workspace_bytes(D, H, max(split, 2)) // 4 + 1, device='cuda')
    ts = []
    for it in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(lib.las_gemm_tn_lstm(hip.p(x), D, D, hip.p(y), 2 * H, H, -1, T, hip.p(dz), 8 * H, hip.p(gk), hip.p(gb),
                                       BT, split | 0x10000, hip.p(ws), hip.stream()))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    fl = 2.0 * (D + H + 1) * 4 * H * BT
    print('%-10s %s TN M=%d N=%d K=%d split %d: %.1f us  %.0f TFLOP/s' % (tag, name, D + H + 1, 4 * H, BT, split, min(ts) * 1e3, fl / min(ts) / 1e9))
