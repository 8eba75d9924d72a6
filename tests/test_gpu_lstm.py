"""GPU parity of the listener kernels (las_lstm_recurrent_fwd/bwd + the GEMMs around them) against the
oracle's dynamic_rnn in its bf16 storage model (oracle/las_oracle.py, mxu='bf16') on identical inputs.

Tolerances (stated per SURVEY.md §4): forward outputs are bf16 values: |err| <= 2 bf16 ulp of the
max-abs (1.6e-2 relative) plus fp32-vs-fp64 accumulation; states/gates fp32: 3e-3.  Gradients pass
through bf16 dz on the device and in the oracle's 'bf16' model alike: 1e-2 of the per-tensor max-abs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DT = torch.float64


def _setup(B, T, D, H, lengths, seed=0, nd=2):
    from oracle import las_oracle as O
    rng = np.random.default_rng(seed)
    x = torch.tensor(rng.standard_normal((B, T, D)), dtype=DT)
    length = torch.tensor(lengths)
    for b in range(B):
        x[b, lengths[b]:] = 0
    x = O.q_bf16(x)
    var = {}
    for d in ['fw', 'bw'][:nd]:
        var[f'L/{d}/lstm_cell/kernel'] = torch.tensor(rng.uniform(-0.2, 0.2, (D + H, 4 * H)).astype(np.float32), dtype=DT)
        var[f'L/{d}/lstm_cell/bias'] = torch.tensor(rng.uniform(-0.3, 0.3, (4 * H,)).astype(np.float32), dtype=DT)
    return x, length, var


def _relerr(got, ref):
    ref = ref.detach()
    return float((got.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-12))


@pytest.mark.parametrize('B,T,D,H,lengths', [
    (3, 10, 24, 64, [10, 4, 7]),
    (17, 12, 40, 128, [12, 1, 5, 12, 3, 9, 2, 11, 6, 8, 10, 4, 12, 7, 1, 12, 5]),
    (2, 6, 64, 256, [6, 3]),
    (21, 40, 32, 256, [40, 3, 17, 40, 1, 25, 8, 33, 12, 40, 5, 29, 2, 38, 9, 21, 40, 7, 31, 15, 36]),
    (19, 14, 24, 512, [14, 3, 9, 14, 1, 7, 12, 5, 14, 2, 11, 6, 13, 4, 10, 8, 14, 1, 9]),
    (9, 11, 80, 256, [11, 2, 7, 11, 5, 1, 9, 3, 11]),              # three 32-deep chunks of the fused input projection (F = 80)
    (5, 9, 72, 128, [9, 4, 1, 9, 6]),
    (6, 7, 104, 256, [7, 3, 7, 1, 5, 2]),                          # wider than the fused projection takes: the separate product
    (140, 6, 16, 256, [6 - (i * 7) % 6 for i in range(140)]),      # the policy's own choice here: 8-row slices forward, 16 backward
    (600, 5, 16, 256, [5 - (i * 3) % 5 for i in range(600)]),      # 76 chains x 4 workgroups (+ companions): more than the device holds at once
])
@pytest.mark.parametrize('rows', [0, 8, 16])
def test_bilstm_forward_backward_vs_oracle(B, T, D, H, lengths, rows, monkeypatch):
    """rows: 0 = the library's choice of utterances per slice (4-row slices at these batch sizes),
    8 / 16 = forced through LAS_LSTM_ROWS -- all three layouts of the 64 / 128 / 256 / 512-unit kernels are covered."""
    from phones_las_amd import hip
    hip.set_knob('LAS_LSTM_ROWS', rows)              # (the library reads its environment once: las_set_knob is the tests' hook)
    try:
        _bilstm_case(B, T, D, H, lengths)
    finally:
        hip.set_knob('LAS_LSTM_ROWS', 0)


def _bilstm_case(B, T, D, H, lengths):
    from oracle import las_oracle as O
    from phones_las_amd.las import ops
    x, length, var = _setup(B, T, D, H, lengths)
    leaf = {k: v.clone().requires_grad_(True) for k, v in var.items()}
    xr = x.clone().requires_grad_(True)
    (fw, bw), (sfw, sbw) = O.bilstm(xr, length, leaf, 'L', O.q_bf16)
    ref_y = torch.cat([fw, bw], -1)

    dvar = {k: v.float().cuda() for k, v in var.items()}
    Dp = (D + 7) // 8 * 8
    xd = torch.zeros(B, T, Dp, dtype=torch.bfloat16, device='cuda')
    xd[..., :D] = x.to(torch.bfloat16).cuda()
    ld = length.to(torch.int32).cuda()
    tape = []
    (ofw, obw), (dsfw, dsbw) = ops.bilstm(xd, ld, H, 0.0, ops.TRAIN, variables=dvar, scope='L', tape=tape,
                                          in_features=D)
    y = ops.concat_outputs((ofw, obw))
    torch.cuda.synchronize()
    assert y.shape == (B, T, 2 * H)
    ops.check_lstm_status(B, H, 2)
    assert _relerr(y.float(), ref_y) < 1.6e-2
    for b in range(B):          # outputs beyond the length are exactly zero (Appendix A.3)
        if lengths[b] < T:
            assert float(y[b, lengths[b]:].float().abs().max()) == 0.0
    assert _relerr(dsfw.c, sfw[0]) < 3e-3 and _relerr(dsbw.c, sbw[0]) < 3e-3
    assert _relerr(dsfw.h, sfw[1]) < 1.6e-2 and _relerr(dsbw.h, sbw[1]) < 1.6e-2

    # backward: random cotangents on outputs and final states
    rng = np.random.default_rng(5)
    dy = torch.tensor(rng.standard_normal((B, T, 2 * H)), dtype=DT)
    dc = torch.tensor(rng.standard_normal((2, B, H)), dtype=DT)
    dh = torch.tensor(rng.standard_normal((2, B, H)), dtype=DT)
    obj = (ref_y * dy).sum() + (sfw[0] * dc[0]).sum() + (sbw[0] * dc[1]).sum() + (sfw[1] * dh[0]).sum() + (sbw[1] * dh[1]).sum()
    obj.backward()
    grads = {k: torch.zeros_like(v) for k, v in dvar.items()}
    dx = ops.bilstm_backward(tape[0], dy.float().cuda().contiguous(),
                             (dc.float().cuda().contiguous(), dh.float().cuda().contiguous()), grads)
    torch.cuda.synchronize()
    ops.check_lstm_status(B, H, 2)
    for k in grads:
        assert _relerr(grads[k], leaf[k].grad) < 1e-2, k
    assert _relerr(dx, xr.grad) < 1e-2
    return y, dx, grads


@pytest.mark.parametrize('rows', [0, 4, 8, 16])
@pytest.mark.parametrize('B,T', [(19, 14), (70, 9)])
def test_512_unit_chains_on_every_slice_height(rows, B, T, monkeypatch):
    """The 512-unit recurrences: 8 members of 64 units (one 16-unit block per wave over the whole K, the weights pinned in
    AccVGPRs, 7 peers) on 4-, 8- and 16-row slices; 8-row slices send the two rows' partial dh sums of a tile as ONE granule of
    two bf16 values.  B=70: several slices per direction with a ragged tail.  (The round-2 form of 16 members of 32 units and
    the fp32-granule exchange on 8-row slices were removed in round 6: measured slower, no default used them.)"""
    lengths = [T - (i * 5) % T for i in range(B)]
    test_bilstm_forward_backward_vs_oracle(B, T, 24, 512, lengths, rows, monkeypatch)


@pytest.mark.parametrize('H,B,T', [(256, 21, 40), (128, 21, 40), (256, 70, 9)])
@pytest.mark.parametrize('rows', [4, 8])
def test_backward_with_helper_waves_is_the_four_wave_backward_bit_for_bit(H, B, T, rows, monkeypatch):
    """Round 6: the eight-wave backward (helper waves load, fold the coefficients and store dz; LAS_LSTM_BWD_HW, default) computes
    what the four-wave form computes, in the same order: dz, dx-side gradients and state gradients are torch.equal; and both sit
    within the oracle's tolerance (test_bilstm_forward_backward_vs_oracle runs under either)."""
    from phones_las_amd import hip
    lengths = [T - (i * 5) % T for i in range(B)]
    out = []
    hip.set_knob('LAS_LSTM_ROWS', rows)
    try:
        for hw in (0, 1):
            hip.set_knob('LAS_LSTM_BWD_HW', hw)
            out.append(_bilstm_case(B, T, 24, H, lengths))
    finally:
        hip.set_knob('LAS_LSTM_BWD_HW', 1)
        hip.set_knob('LAS_LSTM_ROWS', 0)
    (y0, dx0, g0), (y1, dx1, g1) = out
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


def test_unidirectional_and_pyramid_view():
    from oracle import las_oracle as O
    from phones_las_amd.las import ops
    B, T, D, H = 4, 8, 16, 64
    x, length, var = _setup(B, T, D, H, [8, 5, 2, 7], nd=1)
    fw, (c, h) = O.bilstm(x, length, var, 'L', O.q_bf16, unidirectional=True)
    dvar = {k: v.float().cuda() for k, v in var.items()}
    y, st = ops.bilstm(x.to(torch.bfloat16).cuda(), length.to(torch.int32).cuda(), H, 0.0, ops.TRAIN, True,
                       variables=dvar, scope='L')
    assert _relerr(y.float(), fw) < 1.6e-2 and _relerr(st.c, c) < 3e-3
    stacked, l2 = ops.pyramidal_stack(y, length.to(torch.int32).cuda())
    ref, rl = O.pyramidal_stack(fw, length)
    assert stacked.shape == (B, T // 2, 2 * H) and l2.cpu().tolist() == rl.tolist()
    assert stacked.data_ptr() == y.data_ptr()          # zero-copy
    assert _relerr(stacked.float(), ref) < 1.6e-2


@pytest.mark.parametrize('B,T,D,H', [(13, 23, 40, 256), (5, 17, 39, 128), (10, 12, 80, 512), (70, 9, 24, 256)])
def test_fused_input_projection_matches_the_separate_product(B, T, D, H, monkeypatch):
    """Round 4: for narrow inputs (the features) x_t K_x + b is formed inside the recurrent kernel (las_lstm_recurrent_fwd_x).
    Same inputs through both forms: the saved gates agree to fp32 rounding of one more accumulation order, y to a bf16 ulp."""
    from phones_las_amd.las import ops
    lengths = [T - (i * 5) % T for i in range(B)]
    x, length, var = _setup(B, T, D, H, lengths)
    dvar = {k: v.float().cuda() for k, v in var.items()}
    Dp = (D + 7) // 8 * 8
    xd = torch.zeros(B, T, Dp, dtype=torch.bfloat16, device='cuda')
    xd[..., :D] = x.to(torch.bfloat16).cuda()
    ld = length.to(torch.int32).cuda()
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, 'FUSED_X', fused)
        tape = []
        (ofw, obw), st = ops.bilstm(xd, ld, H, 0.0, ops.TRAIN, variables=dvar, scope='L', tape=tape, in_features=D)
        torch.cuda.synchronize()
        ops.check_lstm_status(B, H, 2)
        assert (tape[0]['weights'].kx_chunks > 0) == fused
        out[fused] = (ops.concat_outputs((ofw, obw)).float().clone(), tape[0]['gates'].clone(), tape[0]['cbuf'].clone(), st)
    (ya, ga, ca, sa), (yb, gb, cb, sb) = out[True], out[False]
    mask = (torch.arange(T, device='cuda')[None, :] < ld[:, None])[..., None]
    assert float(((ga - gb) * mask).abs().max()) < 2e-2 * 1.0          # gates are in (-1, 1); bf16 h feeds back into them
    assert float((ya - yb).abs().max()) <= 2 ** -6                        # a couple of bf16 ulps of |h| < 1
    assert float(((ca - cb) * mask).abs().max()) < 3e-2
    for d in range(2):
        assert _relerr(sa[d].c, sb[d].c.double().cpu()) < 1e-2


def test_exchange_tags_need_no_memset_between_launches():
    """Round 4: the exchange buffer is zeroed once, at allocation; every launch offsets its tags by a base kept in the
    workspace header and moves the base past them when its last workgroup leaves.  Forward and backward launches of two
    layer shapes share one workspace here, interleaved, many times over: results stay bit-identical, the base grows, and a
    base about to wrap is started over (the last workgroup clears the buffer)."""
    from phones_las_amd.las import ops
    B, H = 21, 256
    ws = ops.lstm_workspace(B, H, 2)
    hdr = ws[:64].view(torch.int32)

    def run(T, D, seed):
        lengths = [T - (i * 5) % T for i in range(B)]
        x, length, var = _setup(B, T, D, H, lengths, seed=seed)
        dvar = {k: v.float().cuda() for k, v in var.items()}
        xd = x.to(torch.bfloat16).cuda()
        ld = length.to(torch.int32).cuda()
        tape = []
        (ofw, obw), _ = ops.bilstm(xd, ld, H, 0.0, ops.TRAIN, variables=dvar, scope='L', tape=tape, in_features=D)
        y = ops.concat_outputs((ofw, obw)).clone()
        dy = torch.ones(B, T, 2 * H, dtype=torch.float32, device='cuda')
        grads = {k: torch.zeros_like(v) for k, v in dvar.items()}
        dx = ops.bilstm_backward(tape[0], dy, None, grads)
        torch.cuda.synchronize()
        ops.check_lstm_status(B, H, 2)
        return y, dx.clone(), {k: v.clone() for k, v in grads.items()}

    ref = {(T, D): run(T, D, 3) for (T, D) in [(40, 32), (13, 512)]}
    base0 = int(hdr[4])
    assert base0 > 0 and int(hdr[5]) == 0                     # the arrival counter is back at zero after every launch
    for it in range(6):
        if it == 3:
            hdr[4] = 0x7ff00000 - 50                          # a base about to wrap: the next launches start it over
            torch.cuda.synchronize()
        for key in ref:
            y, dx, g = run(key[0], key[1], 3)
            assert torch.equal(y, ref[key][0]) and torch.equal(dx, ref[key][1])
            for k in g:
                assert torch.equal(g[k], ref[key][2][k]), k
        assert int(hdr[5]) == 0
    assert 0 < int(hdr[4]) < 10000                            # started over from zero


@pytest.mark.parametrize('B,T,D,H,ragged', [(21, 40, 512, 256, True), (64, 48, 512, 256, False), (64, 37, 1024, 256, True),
                                             (19, 23, 128, 256, True), (5, 7, 512, 256, True)])
def test_streamed_input_product_matches_the_product_before_the_recurrence(B, T, D, H, ragged, monkeypatch):
    """Round 4: the input product of the upper layers runs on a second stream BESIDE the recurrence (las_gemm_nt_stream: tiles
    of 16 utterances x 16 steps in consumption order, a counter per block; las_lstm_recurrent_fwd_ex(ready=...) waits on it).
    Same inputs through both forms, several times over (the hand-over is a race if anything is wrong with it): the saved
    gates, cell states and outputs must agree -- bit for bit where both products run on the same matrix-core instruction."""
    from phones_las_amd.las import ops
    monkeypatch.setattr(ops, 'STREAM_MIN_ROWS', 0)
    monkeypatch.setattr(ops, 'STREAM_ALWAYS', True)
    lengths = [T - (i * 5) % T if ragged else T for i in range(B)]
    x, length, var = _setup(B, T, D, H, lengths)
    dvar = {k: v.float().cuda() for k, v in var.items()}
    xd = (x * 0.25).to(torch.bfloat16).cuda()
    ld = length.to(torch.int32).cuda()
    out = {}
    for streamed in (False, True, True, True):
        monkeypatch.setattr(ops, 'STREAM_X', streamed)
        tape = []
        (ofw, obw), st = ops.bilstm(xd, ld, H, 0.0, ops.TRAIN, variables=dvar, scope='L', tape=tape, in_features=D)
        torch.cuda.synchronize()
        ops.check_lstm_status(B, H, 2)
        got = (ops.concat_outputs((ofw, obw)).float().clone(), tape[0]['gates'].clone(), tape[0]['cbuf'].clone())
        if streamed not in out:
            out[streamed] = got
        else:                                     # streamed runs repeat exactly
            for a, b in zip(got, out[streamed]):
                mask = (torch.arange(T, device='cuda')[None, :] < ld[:, None])[..., None]
                assert torch.equal(a * mask, b * mask)
    mask = (torch.arange(T, device='cuda')[None, :] < ld[:, None])[..., None]
    (ya, ga, ca), (yb, gb, cb) = out[True], out[False]
    assert float(((ga - gb) * mask).abs().max()) < 2e-2
    assert float((ya - yb).abs().max()) <= 2 ** -6
    assert float(((ca - cb) * mask).abs().max()) < 3e-2


def test_streamed_input_product_whose_chain_starts_late(monkeypatch):
    """the streamed producer's workgroups wait for the chain groups to publish their XCDs.  With a budget of 2^15 polls
    (some 40 ms) a recurrence that started later -- queued behind other work, short of CUs -- found every producer gone, burnt
    its whole bounded wait and had the optimiser step withheld.  Here the chain's stream is held back 120 ms (las_stream_delay)
    AFTER the product has been handed to its stream: the producers must still be there (their budget is the chain's own now),
    the launch must finish clean and give the bits of an undisturbed streamed launch."""
    from phones_las_amd import hip
    from phones_las_amd.las import ops
    monkeypatch.setattr(ops, 'STREAM_MIN_ROWS', 0)
    monkeypatch.setattr(ops, 'STREAM_ALWAYS', True)
    monkeypatch.setattr(ops, 'STREAM_X', True)
    B, T, D, H = 64, 48, 512, 256
    x, length, var = _setup(B, T, D, H, [T] * B)
    dvar = {k: v.float().cuda() for k, v in var.items()}
    xd = (x * 0.25).to(torch.bfloat16).cuda()
    ld = length.to(torch.int32).cuda()

    def run(delay_us):
        real = hip.lib().las_lstm_recurrent_fwd_ex
        calls = []

        def late(p, stream):                       # the recurrence's launch, behind a delay kernel on ITS stream
            for _ in range(delay_us // 1000):       # (las_stream_delay takes at most 1000 us per launch)
                hip.check(hip.lib().las_stream_delay(1000, stream))
            calls.append(1)
            return real(p, stream)
        monkeypatch.setattr(hip.lib(), 'las_lstm_recurrent_fwd_ex', late, raising=False)
        try:
            tape = []
            (ofw, obw), _ = ops.bilstm(xd, ld, H, 0.0, ops.TRAIN, variables=dvar, scope='L', tape=tape, in_features=D)
            torch.cuda.synchronize()
        finally:
            monkeypatch.setattr(hip.lib(), 'las_lstm_recurrent_fwd_ex', real, raising=False)
        assert calls, 'the streamed path launches the recurrence through las_lstm_recurrent_fwd_ex'
        ops.check_lstm_status(B, H, 2)             # raises on a timeout
        return ops.concat_outputs((ofw, obw)).float().clone(), tape[0]['gates'].clone()

    y0, g0 = run(0)
    y1, g1 = run(120000)
    assert ops.STREAM_X, 'the late start must not have switched streaming off'
    assert torch.equal(y0, y1) and torch.equal(g0, g1)


def test_1024_units_two_chunks_of_chain_groups(monkeypatch):
    """B = 64 bidirectional at 1024 units is 8 slices x 2 directions x 32 members = 512 chain workgroups -- two
    chunks of 8 groups, more than the device holds at once; forward and backward against the oracle."""
    test_bilstm_forward_backward_vs_oracle(64, 5, 16, 1024, [5 - (i * 3) % 5 for i in range(64)], 0, monkeypatch)


@pytest.mark.parametrize('mode', ['dropout', 'split'])
def test_streamed_input_product_with_one_operand_per_direction(mode, monkeypatch):
    """las_gemm_nt_stream_dirs: the streamed product when the two directions read DIFFERENT operands -- the two masked copies of
    the input under DropoutWrapper (independent masks for the fw and bw cells), or each direction's own column range of the
    layer below (the stacked listener's split inputs).  Against the products before the recurrence, repeated (race check)."""
    from phones_las_amd.las import ops
    monkeypatch.setattr(ops, 'STREAM_MIN_ROWS', 0)
    monkeypatch.setattr(ops, 'STREAM_ALWAYS', True)
    B, T, D, H = 19, 23, 128, 256
    lengths = [T - (i * 5) % T for i in range(B)]
    x, length, var = _setup(B, T, D, H, lengths)
    dvar = {k: v.float().cuda() for k, v in var.items()}
    xd = (x * 0.25).to(torch.bfloat16).cuda()
    if mode == 'split':
        xd = torch.cat([xd, torch.flip(xd, dims=[2])], -1).contiguous()          # [B, T, 2 D]: the directions' own columns
    ld = length.to(torch.int32).cuda()
    kw = dict(rng=(77, 40)) if mode == 'dropout' else dict(split_inputs=True)
    out = {}
    for streamed in (False, True, True):
        monkeypatch.setattr(ops, 'STREAM_X', streamed)
        tape = []
        (ofw, obw), st = ops.bilstm(xd, ld, H, 0.25 if mode == 'dropout' else 0.0, ops.TRAIN, variables=dvar, scope='L', tape=tape, **kw)
        torch.cuda.synchronize()
        ops.check_lstm_status(B, H, 2)
        got = (ops.concat_outputs((ofw, obw)).float().clone(), tape[0]['gates'].clone(), tape[0]['cbuf'].clone())
        mask = (torch.arange(T, device='cuda')[None, :] < ld[:, None])[..., None]
        if streamed not in out:
            out[streamed] = got
        else:
            for a, b in zip(got, out[streamed]):
                assert torch.equal(a * mask, b * mask)
    (ya, ga, ca), (yb, gb, cb) = out[True], out[False]
    assert float(((ga - gb) * mask).abs().max()) < 2e-2
    assert float((ya - yb).abs().max()) <= 2 ** -6
    assert float(((ca - cb) * mask).abs().max()) < 3e-2
    assert float(ga.abs().max()) > 0.1
