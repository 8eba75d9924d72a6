# las_ctc_loss of this tree against the round-5 library (ab_base/: a built git worktree) on cfg4's shape and on small ragged ones:
# per-example losses and d(logits) must be BIT-IDENTICAL (the round-6 kernel moves the same arithmetic differently); then the time.
#   python scripts/gpu_ctc_ab.py run <lib.so> <out.pt>      python scripts/gpu_ctc_ab.py cmp a.pt b.pt
import os, sys, ctypes, torch
def run(libpath, out):
    sys.path.insert(0, '.')
    from phones_las_amd import hip                     # (for p() / check() / stream() only: the library under test is loaded by hand --
    lib = ctypes.CDLL(libpath)                         #  an older build lacks entry points hip.py binds)
    lib.las_ctc_workspace_bytes.restype = ctypes.c_size_t
    V, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    lib.las_ctc_loss.argtypes = [V, L, V, L, V, V, I, I, I, I, I, F, F, V, V, V, V, V]
    hip.check = lambda rc: None if rc == 0 else (_ for _ in ()).throw(RuntimeError('rc %d' % rc))
    res = {}
    for name, (B, T, C, Cp, U) in {'cfg4': (64, 100, 65, 72, 80), 'small': (5, 17, 9, 16, 6), 'long': (3, 400, 40, 40, 150)}.items():
        g = torch.Generator().manual_seed(7)
        logits = (torch.randn(B, T, Cp, generator=g) * 2).cuda()
        labels = torch.randint(1, C, (B, U), generator=g).to(torch.int32).cuda()
        ll = torch.randint(1, U + 1, (B,), generator=g).to(torch.int32); ll[0] = U
        tl = torch.randint(max(1, T // 2), T + 1, (B,), generator=g).to(torch.int32); tl[0] = T
        d_ll, d_tl = ll.cuda(), tl.cuda()
        ws = torch.empty(lib.las_ctc_workspace_bytes(B, T, Cp, U), dtype=torch.uint8, device='cuda')
        loss = torch.zeros(1, device='cuda'); per = torch.empty(B, device='cuda')
        dl = torch.empty(B, T, Cp, dtype=torch.bfloat16, device='cuda')
        ts = []
        for rep in range(6):
            loss.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hip.check(lib.las_ctc_loss(hip.p(logits), Cp, hip.p(labels), U, hip.p(d_ll), hip.p(d_tl), B, T, C, U, 0, 1.0 / B, 0.3 / B,
                                       hip.p(ws), hip.p(loss), hip.p(per), hip.p(dl), hip.stream()))
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[name] = (per.cpu(), dl.cpu(), sorted(ts)[len(ts) // 2])
        print(libpath.split('/')[-3] if 'ab_base' in libpath else 'this tree', name, 'B=%d T=%d C=%d U=%d: %.0f us' % (B, T, C, U, res[name][2]))
    torch.save(res, out)
def cmp(a, b):
    ra, rb = torch.load(a), torch.load(b)
    for k in ra:
        print(k, 'per-example losses equal:', torch.equal(ra[k][0], rb[k][0]), ' d(logits) equal:', torch.equal(ra[k][1], rb[k][1]))
if sys.argv[1] == 'run': run(sys.argv[2], sys.argv[3])
else: cmp(sys.argv[2], sys.argv[3])
