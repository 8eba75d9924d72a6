# round 5: (1) does a CU-masked stream confine a kernel's workgroups to the chosen XCDs?  (2) a recurrent launch alone, beside a
# GEMM loop that shares its XCDs, and beside the same loop confined to OTHER XCDs (B = 48: 120 chain workgroups fit 4 XCDs)
import ctypes, os, sys, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
from phones_las_amd.las import ops
lib = hip.lib()


def masked(xcds):
    h = ctypes.c_void_p()
    hip.check(lib.las_stream_create_masked(xcds, ctypes.byref(h)))
    return torch.cuda.ExternalStream(h.value)


for xcds in (0xff, 0x0f, 0xf0, 0x1f, 0xe0, 0x01):
    st = masked(xcds)
    counts = torch.zeros(8, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        hip.check(lib.las_xcd_histogram(hip.p(counts), 256, 50, hip.stream()))
    torch.cuda.synchronize()
    print('mask %02x -> workgroups per XCD %s' % (xcds, counts.tolist()))

H, nd = 256, 2
B = int(os.environ.get('B', 48)); T = 800
torch.manual_seed(0)
xproj = torch.randn(B, T, nd * 4 * H, device='cuda') * 0.5
x0 = xproj.clone()
kh = torch.randn(nd, H, 4 * H, device='cuda') * 0.05
wp = torch.empty(nd * H * 4 * H, dtype=torch.bfloat16, device='cuda')
for d in range(nd):
    hip.check(lib.las_lstm_pack_recurrent(hip.p(kh[d]), H, hip.p(wp[d * H * 4 * H:]), hip.stream()))
khb = kh.view(nd, H, 4, H).transpose(2, 3).reshape(nd, H, 4 * H).to(torch.bfloat16).contiguous()
length = torch.full((B,), T, dtype=torch.int32, device='cuda')
y = torch.empty(B, T, nd * H, dtype=torch.bfloat16, device='cuda')
cbuf = torch.empty(B, T, nd * H, device='cuda')
cl = torch.empty(nd, B, H, device='cuda'); hl = torch.empty(nd, B, H, device='cuda')
dy = torch.randn(B, T, nd * H, device='cuda') * 0.1
dz = torch.empty(B, T, nd * 4 * H, dtype=torch.bfloat16, device='cuda')
ws = ops.lstm_workspace(B, H, nd)
# the GEMM beside: dX-shaped NT product (51200 x 512 x 2048), repeated
ga = torch.randn(51200, 2048, device='cuda').to(torch.bfloat16)
gb = torch.randn(512, 2048, device='cuda').to(torch.bfloat16)
gc = torch.empty(51200, 512, device='cuda')


def chain(stream):
    tf, tb = [], []
    for it in range(4):
        xproj.copy_(x0)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            hip.check(lib.las_lstm_recurrent_fwd(hip.p(xproj), hip.p(wp), hip.p(length), hip.p(y), hip.p(cbuf), hip.p(cl), hip.p(hl),
                                                 hip.p(ws), B, T, H, nd, hip.stream()))
            e[1].record()
            hip.check(lib.las_lstm_recurrent_bwd(hip.p(xproj), hip.p(cbuf), hip.p(dy), None, None, hip.p(khb), hip.p(length), hip.p(dz),
                                                 hip.p(ws), B, T, H, nd, None, hip.stream()))
            e[2].record()
        yield
        torch.cuda.synchronize()
        ops.check_lstm_status(B, H, nd)
        tf.append(e[0].elapsed_time(e[1])); tb.append(e[1].elapsed_time(e[2]))
    print('    fwd %.3f ms (%.3f us/step)  bwd %.3f ms (%.3f us/step)' % (min(tf), min(tf) * 1e3 / T, min(tb), min(tb) * 1e3 / T))
    yield 'done'


def run(label, chain_stream, gemm_stream, n_gemm):
    print(label)
    for _ in chain(chain_stream):
        if _ == 'done':
            break
        if gemm_stream is not None:
            with torch.cuda.stream(gemm_stream):
                hip.check(lib.las_stream_delay(20, hip.stream()))
                for _k in range(n_gemm):
                    hip.gemm_nt(ga, gb, gc, 51200, 512, 2048)


plain, side = torch.cuda.Stream(), torch.cuda.Stream()
run('chain alone, unmasked', plain, None, 0)
run('chain unmasked + GEMM loop unmasked', plain, side, 12)
lo, hi = masked(0x0f), masked(0xf0)
run('chain on XCDs 0-3 alone', lo, None, 0)
run('chain on XCDs 0-3 + GEMM loop on XCDs 4-7', lo, hi, 6)
run('chain on XCDs 0-3 + GEMM loop unmasked', lo, side, 12)
