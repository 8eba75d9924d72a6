# round 5: the NT ring's ping-pong schedule against the one-barrier form and the vendor library (a yardstick, not a product path):
# interleaved in one process through las_set_knob, correctness of every variant against the float64 product on a sample of rows
import os, sys, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
shapes = (('dX L1 (M)', 51200, 512, 2048), ('dX L2 (M)', 25600, 1024, 2048), ('xKx L1 (M)', 51200, 2048, 512), ('xKx L2 (M)', 25600, 2048, 1024),
          ('dX L1 (L)', 51200, 1024, 4096), ('dX L2 (L)', 25600, 2048, 4096), ('xKx L1 (L)', 51200, 4096, 1024), ('xKx L2 (L)', 25600, 4096, 2048))
variants = (('ring', 0), ('pp256', 1), ('pp all', 2))
for name, M, N, K in shapes:
    a = (torch.randn(M, K, device='cuda') * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device='cuda')
    bias = torch.randn(N, device='cuda')
    rows = torch.randint(0, M, (64,), device='cuda')
    ref = a[rows].double() @ b.double().t() + bias.double()
    res = {}
    for rep in range(8):
        for vname, pp in variants:
            hip.set_knob('LAS_GEMM_PP', pp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            out.fill_(float('nan'))
            e0.record()
            hip.gemm_nt(a, b, out, M, N, K, lda=K, ldb=K, ldc=N, bias=bias)
            e1.record(); torch.cuda.synchronize()
            err = float((out[rows].double() - ref).abs().max() / ref.abs().max())
            res.setdefault(vname, []).append((e0.elapsed_time(e1), err))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib_out = torch.matmul(a, b.t())
        e1.record(); torch.cuda.synchronize()
        res.setdefault('library (bf16 out)', []).append((e0.elapsed_time(e1), 0.0))
    line = '%-11s M=%5d N=%4d K=%4d:' % (name, M, N, K)
    for vname in [v[0] for v in variants] + ['library (bf16 out)']:
        ts = sorted(t for t, _ in res[vname][2:])
        errs = max(e for _, e in res[vname])
        line += '  %s %.0f us %.0f TF%s' % (vname, ts[len(ts) // 2] * 1e3, 2.0 * M * N * K / ts[len(ts) // 2] / 1e9, '' if errs < 2e-3 else ' ERR %.2e' % errs)
    print(line)
hip.set_knob('LAS_GEMM_PP', 1)
