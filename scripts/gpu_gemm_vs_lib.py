# las_gemm_nt against the vendor library (torch.matmul -> hipBLASLt / rocBLAS) at the bulk NT shapes of metric-M: what a
# tuned library kernel reaches on these shapes (a yardstick for the ring kernel, not a path of the product)
import sys, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
def t_of(fn, n=8):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts) * 1e3
for name, M, N, K, obf in (('proj layer1', 51200, 2048, 512, True), ('proj layer2', 25600, 2048, 1024, True),
                           ('dX layer2->1', 25600, 1024, 2048, False), ('dX layer1->0', 51200, 512, 2048, False),
                           ('keys', 12800, 256, 512, True), ('square 8192', 8192, 8192, 8192, True)):
    a = (torch.randn(M, K, device='cuda') * 0.1).to(torch.bfloat16)
    b = (torch.randn(N, K, device='cuda') * 0.1).to(torch.bfloat16)
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16 if obf else torch.float32)
    mine = t_of(lambda: hip.gemm_nt(a, b, out, M, N, K, lda=K, ldb=K, ldc=N, out_bf16=obf))
    bt = b.t()
    ref = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    libt = t_of(lambda: torch.matmul(a, bt, out=ref))
    fl = 2.0 * M * N * K
    err = float((out.float() - ref.float()).abs().max() / ref.float().abs().max())
    print('%-14s M=%6d N=%5d K=%5d  ring %7.1f us %6.0f TF/s | library (bf16 out) %7.1f us %6.0f TF/s | max rel diff %.1e' % (
        name, M, N, K, mine, fl / mine / 1e6, libt, fl / libt / 1e6, err))
