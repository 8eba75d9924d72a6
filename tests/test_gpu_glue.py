"""The step-glue entry points of include/las_hip.h (las_fill_many, las_train_op_begin, las_grad_l2_norms_acc, las_total_loss,
las_stream_delay) against plain torch, through the C-ABI."""
import ctypes as C
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fill_many_windows_and_casts():
    from phones_las_amd import hip
    dev = 'cuda'
    a = torch.full((5, 7, 12), 3.0, dtype=torch.bfloat16, device=dev)           # zero a strided 2-D window of 2-byte elements
    b = torch.full((33,), 5.0, dtype=torch.float32, device=dev)                  # a contiguous fp32 tensor
    c = torch.full((4, 3, 10), 9.0, dtype=torch.float32, device=dev)            # copy fp32 -> fp32 into a strided window
    d = torch.full((4, 6, 16), 9.0, dtype=torch.bfloat16, device=dev)           # cast fp32 -> bf16 into a strided window
    e = torch.full((2, 8), 1.0, dtype=torch.float32, device=dev)                 # copy with src = None: zeros
    src_c = torch.randn(4, 10, device=dev)
    src_d = torch.randn(4, 5, device=dev)
    src_strided = torch.randn(4, 9, device=dev)[:, 2:7]                          # a strided SOURCE into a contiguous destination
    f = torch.empty(4, 5, dtype=torch.float32, device=dev)
    hip.fill_many(zero=[a[:, 2, :5], b], copy=[(c[:, 1], src_c), (d[:, 3, 4:9], src_d), (e[1], None), (f, src_strided)])
    torch.cuda.synchronize()
    ref = torch.full((5, 7, 12), 3.0, dtype=torch.bfloat16, device=dev)
    ref[:, 2, :5] = 0
    assert torch.equal(a, ref)
    assert float(b.abs().max()) == 0.0
    assert torch.equal(c[:, 1], src_c) and float((c[:, 0] - 9).abs().max()) == 0.0 and float((c[:, 2] - 9).abs().max()) == 0.0
    assert torch.equal(d[:, 3, 4:9], src_d.to(torch.bfloat16)) and float((d[:, 3, :4].float() - 9).abs().max()) == 0.0
    assert float(e[1].abs().max()) == 0.0 and float((e[0] - 1).abs().max()) == 0.0
    assert torch.equal(f, src_strided)
    # more jobs than one launch takes; shapes that cannot be expressed raise
    many = [torch.ones(3, device=dev) for _ in range(2 * hip.FILL_MAX_JOBS + 1)]
    hip.fill_many(zero=many)
    torch.cuda.synchronize()
    assert all(float(t.abs().max()) == 0.0 for t in many)
    with pytest.raises(hip.LasError):
        hip.fill_many(zero=[torch.ones(4, 4, 4, device=dev)[:, :, 1]])


def test_train_op_begin_norms_acc_and_total_loss():
    from phones_las_amd import hip
    lib = hip.lib()
    dev = 'cuda'
    sizes = [5, 1, 40003, 7]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    total, nseg = int(offs[-1]), len(sizes)
    gen = torch.Generator().manual_seed(1)
    g0, p0 = torch.randn(total, generator=gen).to(dev), torch.randn(total, generator=gen).to(dev)
    seg = torch.from_numpy(offs).to(dev)
    l2 = 1e-3
    # status words: two clean workspaces and one with a bit set
    ws = [torch.zeros(16, dtype=torch.int32, device=dev) for _ in range(3)]
    ptrs = torch.tensor([w.data_ptr() for w in ws], dtype=torch.int64, device=dev)
    flag = torch.full((1,), 7.0, device=dev)
    sumsq = torch.full((nseg,), float('nan'), device=dev)
    psq = torch.full((2,), float('nan'), device=dev)
    hip.check(lib.las_train_op_begin(hip.p(ptrs), 3, hip.p(flag), hip.p(sumsq), nseg, hip.p(psq), 2, hip.stream()))
    torch.cuda.synchronize()
    assert float(flag) == 0.0 and float(sumsq.abs().max()) == 0.0 and float(psq.abs().max()) == 0.0
    ws[1][0] = 8
    hip.check(lib.las_train_op_begin(hip.p(ptrs), 3, hip.p(flag), hip.p(sumsq), nseg, hip.p(psq), 2, hip.stream()))
    torch.cuda.synchronize()
    assert float(flag) == 1.0
    # the accumulating norms pass twice = twice the sums of the zeroing one (same gradients, L2 already added once)
    ga, gb = g0.clone(), g0.clone()
    s1, q1 = torch.empty(nseg, device=dev), torch.empty(1, device=dev)
    nws = torch.zeros(lib.las_grad_l2_norms_ws_bytes(nseg, total), dtype=torch.uint8, device=dev)     # fixed-order form for one of them
    hip.check(lib.las_grad_l2_norms(hip.p(ga), hip.p(p0), hip.p(seg), nseg, total, l2, hip.p(s1), hip.p(q1), None, 0, hip.stream()))
    hip.check(lib.las_grad_l2_norms_acc(hip.p(gb), hip.p(p0), hip.p(seg), nseg, total, l2, hip.p(sumsq), hip.p(psq),
                                        hip.p(nws), nws.numel(), hip.stream()))
    torch.cuda.synchronize()
    assert torch.allclose(sumsq, s1, rtol=1e-6) and abs(float(psq[0]) - float(q1)) <= 1e-6 * float(q1) and float(psq[1]) == 0.0
    assert torch.equal(ga, gb)
    assert int(nws[:4].view(torch.int32).item()) == 0          # the arrival counter is back at zero for the next launch
    ref = [float(((g0 + l2 * p0)[offs[i]:offs[i + 1]].double() ** 2).sum()) for i in range(nseg)]
    assert np.allclose(s1.cpu().double().numpy(), ref, rtol=1e-4)
    # total loss = audio + half_l2 * (psq[0] + psq[1])
    audio = torch.tensor([2.5], device=dev)
    psq[1] = 3.0
    out = torch.empty(1, device=dev)
    hip.check(lib.las_total_loss(hip.p(audio), hip.p(psq), 2, 0.5 * l2, hip.p(out), hip.stream()))
    hip.check(lib.las_total_loss(None, hip.p(psq), 1, 0.5 * l2, hip.p(flag), hip.stream()))
    torch.cuda.synchronize()
    assert abs(float(out) - (2.5 + 0.5 * l2 * (float(psq[0]) + 3.0))) < 1e-5
    assert abs(float(flag) - 0.5 * l2 * float(psq[0])) < 1e-6


def test_stream_delay_holds_its_stream_only():
    from phones_las_amd import hip
    lib = hip.lib()
    side = torch.cuda.Stream()
    x = torch.zeros(1, device='cuda')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        hip.check(lib.las_stream_delay(500, side.cuda_stream))
        e1.record()
    t0 = time.perf_counter()
    x.add_(1.0)                                     # the main stream is not held
    torch.cuda.current_stream().synchronize()
    side.synchronize()
    assert 0.3 < e0.elapsed_time(e1) < 5.0          # about 0.5 ms on the side stream
    assert float(x) == 1.0
    assert lib.las_stream_delay(5000, side.cuda_stream) != 0       # out of range: refused
    assert lib.las_stream_delay(0, side.cuda_stream) == 0


def test_add_cast_bf16_windows():
    from phones_las_amd import hip
    lib = hip.lib()
    dev = 'cuda'
    a = torch.randn(5, 3, 40, device=dev)          # row r = a[r, 1, 8:28]: strided window
    b = torch.randn(5, 64, device=dev)
    out = torch.full((5, 7, 24), 9.0, dtype=torch.bfloat16, device=dev)
    hip.check(lib.las_add_cast_bf16(hip.addr(a, 1 * 40 + 8), 3 * 40, hip.addr(b, 16), 64, hip.addr(out, 2 * 24 + 2), 7 * 24, 5, 20, hip.stream()))
    torch.cuda.synchronize()
    ref = (a[:, 1, 8:28] + b[:, 16:36]).to(torch.bfloat16)
    assert torch.equal(out[:, 2, 2:22], ref)
    assert float((out[:, 2, :2].float() - 9).abs().max()) == 0.0 and float((out[:, 2, 22:].float() - 9).abs().max()) == 0.0
    assert float((out[:, 1].float() - 9).abs().max()) == 0.0
    hip.check(lib.las_add_cast_bf16(hip.addr(a, 1 * 40 + 8), 3 * 40, None, 0, hip.addr(out, 2 * 24 + 2), 7 * 24, 5, 20, hip.stream()))
    torch.cuda.synchronize()
    assert torch.equal(out[:, 2, 2:22], a[:, 1, 8:28].to(torch.bfloat16))
    assert lib.las_add_cast_bf16(None, 0, None, 0, hip.addr(out), 24, 5, 20, hip.stream()) != 0


def test_rccl_behind_the_c_abi_one_rank():
    """SURVEY 8(b) / VERDICT r3 #7b: las_dp_unique_id / las_dp_init / las_dp_allreduce_bucket / las_dp_finalize over RCCL, resolved
    at run time (the copy torch has loaded is reused).  One GPU: a 1-rank communicator, the in-place SUM over a gradient bucket
    on a side stream is the identity, asynchronous on that stream, and the flat-buffer offset form (grads + begin) works."""
    import ctypes as C
    from phones_las_amd import hip
    lib = hip.lib()
    assert lib.las_dp_available() == 1
    uid = (C.c_char * 128)()
    hip.check(lib.las_dp_unique_id(uid))
    assert any(bytes(uid))
    comm = C.c_void_p()
    hip.check(lib.las_dp_init(uid, 0, 1, C.byref(comm)))
    assert comm.value
    g = torch.randn(1 << 20, device='cuda')
    ref = g.clone()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    hip.check(lib.las_dp_allreduce_bucket(comm, C.c_void_p(g.data_ptr() + 4 * 1000), g.numel() - 1000, C.c_void_p(st.cuda_stream)))
    st.synchronize()
    assert torch.equal(g, ref)
    assert lib.las_dp_init(uid, 3, 2, C.byref(C.c_void_p())) != 0          # rank out of range: an error code, no crash
    hip.check(lib.las_dp_finalize(comm))


def test_stream_concurrency_probe_and_the_product_stream():
    """The streamed input products need a stream whose kernels run BESIDE an earlier kernel of the launching stream.  After a
    process has created many streams they share hardware queues: las_stream_concurrency_probe tells (a stream against itself: the
    setter can never run beside the waiter), and ops._concurrent_stream() hands out a stream that passes the probe."""
    from phones_las_amd import hip
    from phones_las_amd.las import ops
    lib = hip.lib()
    crowd = [torch.cuda.Stream() for _ in range(48)]          # what a long test session / a process with many models has created
    for st in crowd:
        with torch.cuda.stream(st):
            torch.zeros(1, device='cuda')
    main = torch.cuda.current_stream()
    words = torch.zeros(2, dtype=torch.int32, device='cuda')
    hip.check(lib.las_stream_concurrency_probe(main.cuda_stream, main.cuda_stream, hip.p(words), 500))
    torch.cuda.synchronize()
    assert int(words[1].item()) == 2                          # same queue: in submission order, the waiter gives up
    side = ops._concurrent_stream()
    assert ops.STREAM_X                                       # a stream was found
    hip.check(lib.las_stream_concurrency_probe(main.cuda_stream, side.cuda_stream, hip.p(words), 2000))
    torch.cuda.synchronize()
    assert int(words[1].item()) == 1



def test_image_job_tables_outlive_any_number_of_other_tables():
    """A captured HIP graph replays las_refresh_images with the ADDRESS of its job table baked in, so a table must never be freed.
    Round 6: the cache used to be emptied at 256 tables; in a process that had built enough different models (the whole GPU
    suite, once the image GEMM's weight images added tables) a later tensor took a freed table's memory and the next graph replay
    walked garbage jobs (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in tests/test_gpu_step_forms.py)."""
    from phones_las_amd import hip
    src = torch.randn(4, 8, device='cuda')
    first = torch.empty(4, 8, dtype=torch.bfloat16, device='cuda')
    with hip.image_batch():
        hip.cast_bf16(src, 4, 8, first, 4, 8)
    key0 = [k for k, t in hip._image_tables.items() if int.from_bytes(k[8:16], 'little') == first.data_ptr()]
    assert len(key0) == 1
    table0 = hip._image_tables[key0[0]]
    ptr0 = table0.data_ptr()
    outs = [torch.empty(4, 8, dtype=torch.bfloat16, device='cuda') for _ in range(400)]
    for o in outs:                                   # 400 more tables (one per destination address)
        with hip.image_batch():
            hip.cast_bf16(src, 4, 8, o, 4, 8)
    torch.cuda.synchronize()
    assert hip._image_tables[key0[0]] is table0 and table0.data_ptr() == ptr0
    assert bytes(table0.cpu().numpy().tobytes()) == key0[0]          # ... and still holds its jobs
