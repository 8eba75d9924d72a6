# times one fused decoder step (forward) alone, captured 100x in a graph: LAS_DEC_STREAM=0/1, varying memory length
import os, sys, ctypes as C, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
lib = hip.lib()
B, Hd, M, V = 64, 256, 512, 64
bf, f32 = torch.bfloat16, torch.float32
for Tm, parts in ((200, 4), (200, 1), (200, 2), (200, 8), (48, 4), (8, 4)):
    dev = 'cuda'
    keys = torch.randn(B, Tm, Hd, device=dev).to(bf)
    vals = torch.randn(B, Tm, M, device=dev).to(bf)
    mem_len = torch.full((B,), Tm, dtype=torch.int32, device=dev)
    z = torch.randn(B, 4 * Hd, device=dev)
    tok = torch.randn(V, 4 * Hd, device=dev).to(bf)
    ids = torch.randint(0, V, (B, 4), dtype=torch.int32, device=dev)
    bias = torch.zeros(4 * Hd, device=dev)
    cs = torch.zeros(B, 2, Hd, device=dev)
    gates = torch.empty(B, 4 * Hd, device=dev)
    h = torch.empty(B, Hd, dtype=bf, device=dev)
    Tmp = (Tm + 7) // 8 * 8
    align = torch.zeros(B, Tmp, device=dev); align_bf = torch.zeros(B, Tmp, dtype=bf, device=dev)
    ctx = torch.empty(B, M, dtype=bf, device=dev)
    s = hip.DecStep()
    s.B, s.Hd, s.M, s.Tm, s.attention, s.mode = B, Hd, M, Tm, hip.ATT_LUONG, hip.DEC_FUSED
    s.z, s.bias = hip.addr(z), hip.addr(bias)
    s.tok_rows, s.tok_ids, s.tok_stride = hip.addr(tok), hip.addr(ids), 4
    s.c_prev, s.ldcp = hip.addr(cs), 2 * Hd
    s.gates_out, s.ldg = hip.addr(gates), 4 * Hd
    s.c_out, s.ldco = hip.addr(cs, Hd), 2 * Hd
    s.h_out, s.ldh = hip.addr(h), Hd
    s.keys, s.values, s.mem_len = hip.addr(keys), hip.addr(vals), hip.addr(mem_len)
    s.align_out, s.align_bf16, s.lda = hip.addr(align), hip.addr(align_bf), Tmp
    s.ctx_out, s.ldc = hip.addr(ctx), M
    s.drop_keep, s.feed_width = 1.0, V + M
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            hip.check(lib.las_decoder_step_fwd(C.byref(s), parts, hip.stream()))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(100):
                hip.check(lib.las_decoder_step_fwd(C.byref(s), parts, hip.stream()))
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 10.0)
    print('stream=%s Tm %3d parts %d: %.1f us per step   ctx checksum %.5f' % (os.environ.get('LAS_DEC_STREAM', '1'), Tm, parts, min(ts),
                                                                             float(ctx.float().abs().mean())))
