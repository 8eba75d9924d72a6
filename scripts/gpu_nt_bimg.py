# round 6: the NT product with the weight operand from its MFMA-B image (las_gemm_nt_bimg) against the ring kernels and the vendor
# library (a yardstick, not a product path): interleaved in one process, every variant checked against the float64 product on sampled rows
import os, sys, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
shapes = (('dX L1 (M)', 51200, 512, 2048), ('dX L2 (M)', 25600, 1024, 2048), ('xKx L1 (M)', 51200, 2048, 512), ('xKx L2 (M)', 25600, 2048, 1024),
          ('dX L1 (L)', 51200, 1024, 4096), ('dX L2 (L)', 25600, 2048, 4096), ('xKx L1 (L)', 51200, 4096, 1024), ('xKx L2 (L)', 25600, 4096, 2048))
lib = hip.lib()
for name, M, N, K in shapes:
    a = (torch.randn(M, K, device='cuda') * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    img = torch.empty(N * K, dtype=torch.bfloat16, device='cuda')
    hip.check(lib.las_pack_mfma_b_bf16(hip.p(b), K, N, K, hip.p(img), hip.stream()))
    out = torch.empty(M, N, device='cuda')
    bias = torch.randn(N, device='cuda')
    rows = torch.cat([torch.randint(0, M, (60,), device='cuda'), torch.tensor([0, 255, 256, M - 1], device='cuda')])
    ref = a[rows].double() @ b.double().t() + bias.double()
    res = {}
    for rep in range(8):
        for vname in ('ring', 'image'):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            out.fill_(float('nan'))
            e0.record()
            if vname == 'ring':
                hip.gemm_nt(a, b, out, M, N, K, lda=K, ldb=K, ldc=N, bias=bias)
            else:
                hip.check(lib.las_gemm_nt_bimg(hip.p(a), K, hip.p(img), hip.p(out), N, hip.p(bias), M, N, K, 0, hip.stream()))
            e1.record(); torch.cuda.synchronize()
            err = float((out[rows].double() - ref).abs().max() / ref.abs().max())
            bad = bool(torch.isnan(out).any())
            res.setdefault(vname, []).append((e0.elapsed_time(e1), err if not bad else 9.0))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib_out = torch.matmul(a, b.t())
        e1.record(); torch.cuda.synchronize()
        res.setdefault('library (bf16 out)', []).append((e0.elapsed_time(e1), 0.0))
    line = '%-11s M=%5d N=%4d K=%4d:' % (name, M, N, K)
    for vname in ('ring', 'image', 'library (bf16 out)'):
        ts = sorted(t for t, _ in res[vname][2:])
        errs = max(e for _, e in res[vname])
        line += '  %s %.0f us %.0f TF%s' % (vname, ts[len(ts) // 2] * 1e3, 2.0 * M * N * K / ts[len(ts) // 2] / 1e9, '' if errs < 2e-3 else ' ERR %.2e' % errs)
    print(line)
