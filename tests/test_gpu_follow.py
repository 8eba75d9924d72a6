"""Follower products (round 5; las_gemm_nt_follow, include/las_hip.h): the next layer's input projection behind a forward
recurrence (las/ops.py:75-87: the stacking that makes layer l + 1 wait for layer l) and dX = dz K_x^T behind a backward one,
each formed in two halves, one per direction of the recurrence, as soon as that direction's chain has passed the rows.

* the kernel alone against a float64 product: the clean-up pass on its own, the persistent follower with the chain's words
  scripted by the host (placement, zero fill, progress), a follower that finds no chain (every tile is left to the clean-up);
* every way of dividing the work gives the SAME BITS ((P + Q) + bias whichever half came first);
* the pyramidal listener with followers against the same listener with the products behind the recurrences."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _layout(ngroups, ntiles):
    g16 = (ngroups + 15) // 16 * 16
    p0 = 16 + g16
    return p0, p0 + g16, p0 + 2 * g16, p0 + 2 * g16 + ntiles


def _case(B, T_out, stack, H, N, kind, ragged, seed=0, R=4):
    """A = the chain's output viewed [B, T_out, stack * 2 * H] (kind 0) or dz [B, T_out, 2 * 4H'] (kind 1, stack 1)."""
    g = torch.Generator().manual_seed(seed)
    T_chain = T_out * stack
    length = torch.tensor([T_chain - (i * 7) % (T_chain // 3) if ragged else T_chain for i in range(B)], dtype=torch.int32)
    if kind == 0:
        seg, nseg, Kfull = H, stack, stack * 2 * H
        a_dir, a_seg = H, 2 * H
    else:
        seg, nseg, Kfull = H, 1, 2 * H          # (H plays the role of 4H here)
        a_dir, a_seg = H, 0
    A = (torch.randn(B, T_out, Kfull, generator=g) * 0.5).to(torch.bfloat16)
    Bw = (torch.randn(N, Kfull, generator=g) * 0.1).to(torch.bfloat16)
    bias = torch.randn(N, generator=g) if kind == 0 else None
    return dict(B=B, T_out=T_out, stack=stack, N=N, kind=kind, R=R, length=length, A=A, Bw=Bw, bias=bias, seg=seg, nseg=nseg,
                a_dir=a_dir, a_seg=a_seg, Kfull=Kfull)


def _struct(c, dev, C_, words, workgroups):
    from phones_las_amd import hip
    f = hip.Follow()
    f.A, f.Bw, f.C, f.bias = hip.addr(dev['A']), hip.addr(dev['Bw']), hip.addr(C_), hip.addr(dev['bias'])
    f.lda = f.ldb = c['Kfull']
    f.ldc = c['N']
    f.a_dir, f.a_seg, f.b_dir, f.b_seg, f.nseg, f.seg_len = c['a_dir'], c['a_seg'], c['a_dir'], c['a_seg'], c['nseg'], c['seg']
    f.N, f.B, f.T_out, f.T_chain, f.stack, f.rows_per_slice, f.ndir, f.kind = (c['N'], c['B'], c['T_out'], c['T_out'] * c['stack'], c['stack'],
                                                                               c['R'], 2, c['kind'])
    f.length, f.words, f.workgroups = hip.addr(dev['length']), hip.addr(words), workgroups
    return f


def _run(c, dev, script=None, follower=True, cleanup=True):
    """One follower + clean-up pair.  script(words, layout) writes what a chain would have published.  Returns (C, words)."""
    from phones_las_amd import hip
    lib = hip.lib()
    n = lib.las_gemm_nt_follow_words(c['B'], c['T_out'], c['N'], c['R'], 2)
    words = torch.zeros(n, dtype=torch.int32, device='cuda')
    C_ = torch.full((c['B'], c['T_out'], c['N']), float('nan'), device='cuda')
    if script is not None:
        script(words)
    torch.cuda.synchronize()
    if follower:
        hip.check(lib.las_gemm_nt_follow(ctypes.byref(_struct(c, dev, C_, words, 64)), 0, hip.stream()))
    if cleanup:
        hip.check(lib.las_gemm_nt_follow(ctypes.byref(_struct(c, dev, C_, words, 0)), 1, hip.stream()))
    torch.cuda.synchronize()
    return C_, words


def _reference(c):
    ref = c['A'].double().reshape(-1, c['Kfull']) @ c['Bw'].double().t()
    if c['bias'] is not None:
        ref = ref + c['bias'].double()
    return ref.reshape(c['B'], c['T_out'], c['N'])


@pytest.mark.parametrize('B,T_out,stack,H,N,kind,ragged', [
    (32, 100, 2, 64, 256, 0, True),          # forward, stacked view (two K segments per direction), ragged, a partial last time tile
    (32, 128, 1, 128, 384, 0, False),        # forward, layer 0 -> 1 (one segment), N = 3 column tiles
    (64, 70, 1, 256, 200, 1, True),          # backward: dX = dz K_x^T, no bias, N not a multiple of the tile
    (30, 64, 1, 64, 128, 1, True),           # a batch that does not fill its last slice (nslices = 8)
])
def test_follower_kernel_alone(B, T_out, stack, H, N, kind, ragged):
    from phones_las_amd import hip
    lib = hip.lib()
    assert lib.las_gemm_nt_follow_supported(B, N, H, 4, 2) == 1
    c = _case(B, T_out, stack, H, N, kind, ragged)
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in c.items()}
    ref = _reference(c)
    nslices = (B + 3) // 4
    sbo = 64
    ntb, nct = (T_out + sbo - 1) // sbo, (N + 127) // 128
    p0, z0, s0, total = _layout(2 * nslices, nslices * ntb * nct)
    assert total == lib.las_gemm_nt_follow_words(B, T_out, N, 4, 2)

    # (1) the clean-up pass alone forms everything
    C1, w1 = _run(c, dev, follower=False)
    scale = float(ref.abs().max())
    assert float((C1.double().cpu() - ref).abs().max()) <= 2e-3 * scale
    assert bool((w1[s0:total] == 0x300).all())                         # both halves done, none claimed by a follower

    # (2) a follower that finds no chain leaves (bounded wait) and the clean-up pass forms everything: same bits
    C2, w2 = _run(c, dev)
    assert torch.equal(C1, C2)

    # (3) the chain's words scripted as "all done" (every group on the XCD the layout expects): the follower forms
    #     everything, the clean-up pass finds nothing to do: same bits again
    nblk = (T_out * stack + 63) // 64

    def all_done(words):
        for g in range(2 * nslices):
            words[16 + g] = ((g % 8) + 1) | (16 << 8) | (1 << 16)
            words[z0 + g] = 1
            words[p0 + g] = 16 * nblk
    C3, w3 = _run(c, dev, script=all_done)
    assert torch.equal(C1, C3)
    st = w3[s0:total]
    assert bool(((st & 0x300) == 0x300).all())
    # ... and it really was the follower (unless the device placed no workgroup on some XCD): most tiles carry claim bits
    assert int(((st & 3) == 3).sum()) >= int(0.5 * st.numel())
    C3b, _ = _run(c, dev, script=all_done, cleanup=False)
    claimed = ((st & 3) == 3)
    if bool(claimed.all()):
        assert torch.equal(C1, C3b)

    # (4) only ONE direction's chains make progress: the follower stores (some of) that direction's halves, gives the other
    #     direction up after its bounded wait, and the clean-up pass adds what is missing: same bits
    def half_done(words):
        all_done(words)
        for g in range(nslices, 2 * nslices):
            words[p0 + g] = 0
    C4, w4 = _run(c, dev, script=half_done)
    assert torch.equal(C1, C4)
    st4 = w4[s0:total]
    assert bool(((st4 & 0x300) == 0x300).all())
    if not ragged:                       # (a tile beyond every length of its slice needs no step of the chain: taken at once)
        assert bool(((st4 & 2) == 0).all())

    # (5) a group that runs on ANOTHER XCD than its twin (or spread over several): its slice is left to the clean-up pass
    def elsewhere(words):
        all_done(words)
        words[16 + nslices] = 0xff | (16 << 8) | (1 << 16)
        words[16 + 1] = (((1 % 8) + 1) % 8 + 1) | (16 << 8) | (1 << 16)
    C5, w5 = _run(c, dev, script=elsewhere)
    assert torch.equal(C1, C5)
    st5 = w5[s0:total].view(nslices, ntb * nct)
    assert bool(((st5 & 0x300) == 0x300).all()) and bool((st5[0] == 0x300).all()) and bool((st5[1] == 0x300).all())


def _listener_run(cfg, follow, monkeypatch, steps=2, cleanup_only=False):
    import bench
    from phones_las_amd import model_helper as mh
    from phones_las_amd.las import ops
    monkeypatch.setattr(ops, 'FOLLOW', follow)
    if cleanup_only:
        monkeypatch.setattr(ops, 'FOLLOW_WGS', 0)
    c = bench.CONFIGS[cfg]
    model = mh.LasModel(bench.build_params(c))
    feats, labels = bench.synthetic_batch(c, 1234, torch.device('cuda', 0))
    losses = []
    for _ in range(steps):
        model.vars.grad.zero_()
        audio, logits, dlogits = model.forward_train(feats, labels, num_steps=c['U'])
        model.backward(dlogits)
        losses.append(float(audio))
    torch.cuda.synchronize()
    assert not model.read_and_clear_status()
    return logits.float().clone(), model.vars.grad.clone(), losses


@pytest.mark.parametrize('cfg', ['metric-M-ragged', 'metric-L'])
def test_listener_with_followers_matches_the_products_behind_the_recurrences(cfg, monkeypatch):
    """Forward + backward of the benchmarked models with the follower products against the same passes with the products
    behind the recurrences (a full-K product adds the two directions' halves in one accumulator chain, the follower adds
    two fp32 partial products: differences of fp32 rounding only, amplified through the layers above to the bf16 storage
    level); the follower run twice gives the same bits; so does the clean-up pass doing all of the work."""
    lg0, g0, l0 = _listener_run(cfg, False, monkeypatch)
    lg1, g1, l1 = _listener_run(cfg, True, monkeypatch)
    lg2, g2, l2 = _listener_run(cfg, True, monkeypatch)
    lg3, g3, l3 = _listener_run(cfg, True, monkeypatch, cleanup_only=True)
    assert torch.equal(lg1, lg2) and torch.equal(g1, g2) and l1 == l2
    assert torch.equal(lg1, lg3) and torch.equal(g1, g3)
    assert abs(l0[-1] - l1[-1]) < 2e-3 * abs(l0[-1])
    assert float((lg0 - lg1).abs().max()) < 2e-2 * float(lg0.abs().max())
    assert float((g0 - g1).abs().max()) < 2e-2 * float(g0.abs().max())
    assert float((g0 - g1).norm()) < 1e-2 * float(g0.norm())
