# times the fused LSTM weight-gradient product (las_gemm_tn_lstm) and the input projections (las_gemm_nt) alone at
# the layer shapes of metric-M; LAS_HIP_LIB selects the library under test, SPLIT overrides the K split
import os, sys, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
lib = hip.lib()
B, H = 64, 256
for li, (name, T, D) in enumerate((('layer0', 800, 40), ('layer1', 800, 512), ('layer2', 400, 1024))):
    BT = B * T
    torch.manual_seed(0)
    x = (torch.randn(BT, D, device='cuda') * 0.5).to(torch.bfloat16)
    y = (torch.randn(BT, 2 * H, device='cuda') * 0.5).to(torch.bfloat16)
    dz = (torch.randn(BT, 8 * H, device='cuda') * 0.1).to(torch.bfloat16)
    gk = torch.zeros(D + H, 4 * H, device='cuda'); gb = torch.zeros(4 * H, device='cuda')
    split = os.environ.get('SPLIT', str(max(1, min(32, BT // 2048)))).split(',')
    split = int(split[li % len(split)])
    ts = []
    ws = None
    if os.environ.get('WS', '1') != '0':
        ws = torch.empty(lib.las_gemm_tn_lstm_workspace_bytes(D, H, split) // 4 + 1, device='cuda')
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(lib.las_gemm_tn_lstm(hip.p(x), D, D, hip.p(y), 2 * H, H, -1, T, hip.p(dz), 8 * H, hip.p(gk), hip.p(gb),
                                       BT, split, hip.p(ws), hip.stream()))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    fl = 2.0 * (D + H + 1) * 4 * H * BT
    print('%s TN M=%d N=%d K=%d split %d: %.1f us  %.0f TFLOP/s  checksum %.4f' % (
        name, D + H + 1, 4 * H, BT, split, min(ts) * 1e3, fl / min(ts) / 1e9, float(gk.abs().mean() / len(ts))))
    # projection x K_x -> [BT, 8H] bf16
    w = (torch.randn(8 * H, max(D, 64), device='cuda') * 0.05).to(torch.bfloat16)
    xx = x if D >= 64 else (torch.randn(BT, 64, device='cuda')).to(torch.bfloat16)
    Dk = xx.shape[1]
    out = torch.empty(BT, 8 * H, dtype=torch.bfloat16, device='cuda')
    ts = []
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.gemm_nt(xx, w, out, BT, 8 * H, Dk, lda=Dk, ldb=Dk, ldc=8 * H, out_bf16=True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print('%s NT M=%d N=%d K=%d: %.1f us  %.0f TFLOP/s' % (name, BT, 8 * H, Dk, min(ts) * 1e3, 2.0 * BT * 8 * H * Dk / min(ts) / 1e9))

for name, M, N, K in (('dX layer2->1', 25600, 1024, 2048), ('dX layer1->0', 51200, 512, 2048), ('proj layer1 fp32', 51200, 2048, 512)):
    a = (torch.randn(M, K, device='cuda') * 0.1).to(torch.bfloat16)
    b = (torch.randn(N, K, device='cuda') * 0.1).to(torch.bfloat16)
    out = torch.empty(M, N, device='cuda')
    ts = []
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.gemm_nt(a, b, out, M, N, K, lda=K, ldb=K, ldc=N)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print('%s NT M=%d N=%d K=%d fp32 out: %.1f us  %.0f TFLOP/s' % (name, M, N, K, min(ts) * 1e3, 2.0 * M * N * K / min(ts) / 1e9))
