# host enqueue time against total time of 50 back-to-back dX products (is the launch path the limit?)
import os, sys, time, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
M, N, K = 25600, 1024, 2048
a = (torch.randn(M, K, device='cuda') * 0.5).to(torch.bfloat16)
b = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, device='cuda')
for i in range(3): hip.gemm_nt(a, b, out, M, N, K, lda=K, ldb=K, ldc=N)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50): hip.gemm_nt(a, b, out, M, N, K, lda=K, ldb=K, ldc=N)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('ring=%s host enqueue %.1f us per launch, total %.1f us per launch' % (os.environ.get('LAS_GEMM_RING'), (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
