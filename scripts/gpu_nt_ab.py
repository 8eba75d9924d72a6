# A/B timing of the NT products of a metric-M step: run twice, LAS_GEMM_RING=1 (LDS-DMA ring kernel) and =0 (register-staged
# 128x128 kernel); prints min / median us over interleaved repetitions and TFLOP/s.
import os, sys, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
shapes = (('proj L1 fp32', 51200, 2048, 512, False), ('proj L2 fp32', 25600, 2048, 1024, False),
          ('dX L2', 25600, 1024, 2048, False), ('dX L1', 51200, 512, 2048, False), ('keys bf16', 12800, 256, 1024, True),
          ('dmem acc', 12800, 1024, 256, False))
tot = 0.0
for name, M, N, K, bf in shapes:
    a = (torch.randn(M, K, device='cuda') * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16 if bf else torch.float32)
    bias = torch.randn(N, device='cuda')
    ts = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.gemm_nt(a, b, out, M, N, K, lda=K, ldb=K, ldc=N, bias=bias, out_bf16=bf)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[2:])
    tot += ts[len(ts) // 2]
    print('%-14s M=%d N=%d K=%d: min %.1f us  median %.1f us  %.0f TFLOP/s (median)' % (
        name, M, N, K, ts[0] * 1e3, ts[len(ts) // 2] * 1e3, 2.0 * M * N * K / ts[len(ts) // 2] / 1e9))
print('ring=%s total (median) %.1f us' % (os.environ.get('LAS_GEMM_RING', '1'), tot * 1e3))
