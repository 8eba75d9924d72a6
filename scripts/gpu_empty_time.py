import sys, torch
sys.path.insert(0, '.')
from phones_las_amd import hip
lib = hip.lib()
c = torch.zeros(4, dtype=torch.int32, device='cuda')
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3): hip.check(lib.las_counter_add(hip.p(c), 1, hip.stream()))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(200): hip.check(lib.las_counter_add(hip.p(c), 1, hip.stream()))
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 5.0)
print('tiny dependent kernel in a graph: %.2f us per node' % min(ts))
