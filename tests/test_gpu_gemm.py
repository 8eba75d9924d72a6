"""GPU parity of the MFMA GEMMs (las_gemm_nt / las_gemm_tn, include/las_hip.h) against a float64
product of the same bf16-rounded operands.  Tolerance: fp32 accumulation of K<=4096 bf16 products,
|err| <= 2e-3 * sum|a||b| bound in practice -> rtol 1e-3 on the max-abs scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).to(torch.bfloat16)


def _close(got, ref, tol=2e-3):
    scale = ref.abs().max().item() + 1e-6
    err = (got.double().cpu() - ref).abs().max().item()
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 72, 40), (64, 1024, 1280), (3, 5, 8), (257, 129, 328),
                                   (1024, 256, 128), (1100, 300, 136), (2304, 512, 328),
                                   # the LDS-DMA ring kernel (K % 64 == 0, M >= 1024, N >= 128): one and many K tiles, ragged M / N edges
                                   (1024, 128, 128), (1300, 200, 192), (2500, 1000, 512), (4096, 512, 2048), (1025, 129, 64 * 7)])
def test_gemm_nt_matches_fp64(M, N, K):
    from phones_las_amd import hip
    A, B = _mk((M, K), 1), _mk((N, K), 2)
    bias = torch.randn(N)
    ref = A.double() @ B.double().t() + bias.double()
    Ad, Bd = A.cuda(), B.cuda()
    C = torch.empty(M, N, device='cuda')
    hip.gemm_nt(Ad, Bd, C, M, N, K, bias=bias.cuda())
    _close(C, ref)
    # bf16 output, accumulate, split-K
    Cb = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    hip.gemm_nt(Ad, Bd, Cb, M, N, K, bias=bias.cuda(), out_bf16=True)
    _close(Cb.float(), ref, 1e-2)
    C2 = torch.ones(M, N, device='cuda')
    hip.gemm_nt(Ad, Bd, C2, M, N, K, accumulate=True)
    _close(C2, ref - bias.double() + 1.0)
    C3 = torch.full((M, N), 7.0, device='cuda')
    hip.gemm_nt(Ad, Bd, C3, M, N, K, bias=bias.cuda(), split_k=3)
    _close(C3, ref)


def test_gemm_nt_asymmetric_identity_and_strides():
    # A = I with an asymmetric B catches a transposed C write (guide §3)
    from phones_las_amd import hip
    n = 64
    A = torch.eye(n).to(torch.bfloat16).cuda()
    B = (torch.arange(n * n).reshape(n, n) % 251).float().to(torch.bfloat16)
    C = torch.empty(n, n, device='cuda')
    hip.gemm_nt(A, B.cuda(), C, n, n, n)
    assert torch.equal(C.cpu(), B.float().t())
    # sub-matrix views through lda/ldb/ldc
    big = _mk((96, 160), 3).cuda()
    W = _mk((40, 64), 4).cuda()
    out = torch.zeros(96, 80, device='cuda')
    hip.gemm_nt(big[:, 32:], W, out[:, 16:], 96, 40, 64, lda=160, ldb=64, ldc=80)
    ref = big[:, 32:96].double().cpu() @ W.double().cpu().t()
    _close(out[:, 16:56], ref)
    assert float(out[:, :16].abs().max()) == 0.0 and float(out[:, 56:].abs().max()) == 0.0


def test_gemm_nt_batched():
    from phones_las_amd import hip
    nb, M, N, K = 5, 33, 70, 48
    A, B = _mk((nb, M, K), 5), _mk((nb, N, K), 6)
    C = torch.empty(nb, M, N, device='cuda')
    hip.gemm_nt(A.cuda(), B.cuda(), C, M, N, K, lda=K, ldb=K, ldc=N, batch=nb, sa=M * K, sb=N * K, sc=M * N)
    _close(C, torch.einsum('bmk,bnk->bmn', A.double(), B.double()))


@pytest.mark.parametrize('M,N,K,split', [(128, 128, 256, 1), (40, 1024, 1000, 4), (520, 72, 77, 2), (8, 8, 5, 1)])
def test_gemm_tn_matches_fp64(M, N, K, split):
    from phones_las_amd import hip
    lda, ldb = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    A, B = _mk((K, lda), 7), _mk((K, ldb), 8)
    ref = A[:, :M].double().t() @ B[:, :N].double()
    C = torch.zeros(M, N, device='cuda')
    hip.gemm_tn(A.cuda(), B.cuda(), C, M, N, K, lda=lda, ldb=ldb, split_k=split)
    _close(C, ref)
    hip.gemm_tn(A.cuda(), B.cuda(), C, M, N, K, lda=lda, ldb=ldb, split_k=split)   # accumulates
    _close(C, 2 * ref)


@pytest.mark.parametrize('M,N,K', [(1536, 1024, 256), (300, 200, 64), (4100, 136, 1024)])
def test_gemm_nt_masked_is_the_product_through_the_dropout_backward(M, N, K):
    """las_gemm_nt_masked (the mask of a cell's input dropout in the dX product's epilogue; one direction stores, the other
    accumulates) against las_gemm_nt into two buffers + las_dropout_bwd over them: the same bits."""
    from phones_las_amd import hip
    lib = hip.lib()
    A0, A1 = _mk((M, K), 21).cuda(), _mk((M, K), 22).cuda()
    B0, B1 = _mk((N, K), 23).cuda(), _mk((N, K), 24).cuda()
    keep, seed, st = 0.8, 4242, 16
    p0, p1 = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
    hip.gemm_nt(A0, B0, p0, M, N, K, lda=K, ldb=K, ldc=N)
    hip.gemm_nt(A1, B1, p1, M, N, K, lda=K, ldb=K, ldc=N)
    ref = torch.empty(M, N, device='cuda')
    hip.check(lib.las_dropout_bwd(hip.p(p0), hip.p(p1), hip.p(ref), M, N, keep, seed, st, st + 1, hip.stream()))
    out = torch.full((M, N), float('nan'), device='cuda')
    hip.check(lib.las_gemm_nt_masked(hip.p(A0), K, hip.p(B0), K, hip.p(out), N, M, N, K, 0, keep, seed, st, hip.stream()))
    hip.check(lib.las_gemm_nt_masked(hip.p(A1), K, hip.p(B1), K, hip.p(out), N, M, N, K, 1, keep, seed, st + 1, hip.stream()))
    torch.cuda.synchronize()
    assert float((out == 0).float().mean()) > 0.02              # both masks dropped some elements
    assert torch.equal(out == 0, ref == 0)
    assert float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


@pytest.mark.parametrize('nb,M,N,K', [(3, 200, 1024, 80), (5, 50, 256, 80), (2, 7, 40, 3), (1, 130, 136, 33)])
def test_gemm_tn_store_overwrites(nb, M, N, K):
    """las_gemm_tn_store: C = A^T B into a buffer full of junk (NaN included), batched -- the speller's d(keys) and
    d(memory) products, whose K is the U decoder steps."""
    from phones_las_amd import hip
    lda, ldb = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    A, B = _mk((nb, K, lda), 17), _mk((nb, K, ldb), 18)
    ref = torch.einsum('bkm,bkn->bmn', A[..., :M].double(), B[..., :N].double())
    C = torch.full((nb, M, N), float('nan'), device='cuda')
    hip.gemm_tn(A.cuda(), B.cuda(), C, M, N, K, lda=lda, ldb=ldb, ldc=N, batch=nb, sa=K * lda, sb=K * ldb, sc=M * N, store=True)
    _close(C, ref)
    Cb = torch.full((nb, M, N), float('nan'), device='cuda', dtype=torch.bfloat16)      # bf16 output: one rounding of the fp32 sums
    hip.gemm_tn(A.cuda(), B.cuda(), Cb, M, N, K, lda=lda, ldb=ldb, ldc=N, batch=nb, sa=K * lda, sb=K * ldb, sc=M * N, store=True)
    assert torch.equal(Cb, C.to(torch.bfloat16))
    with pytest.raises(hip.LasError):
        hip.check(hip.lib().las_gemm_tn_store(None, lda, None, ldb, None, N, M, N, 0, 0, 0, 0, 1, 0, 0, 0, 0, hip.stream()))


@pytest.mark.parametrize('shift', [-1, 1])
def test_gemm_tn_shifted_rows(shift):
    # dK_h = sum_t h_{t-1}^T dz_t without a shifted copy: rows cross no utterance boundary
    from phones_las_amd import hip
    Bn, T, M, N = 3, 7, 64, 72
    A, Bm = _mk((Bn * T, M), 9), _mk((Bn * T, N), 10)
    Ash = torch.zeros(Bn, T, M, dtype=torch.float64)
    Av = A.double().reshape(Bn, T, M)
    if shift == -1:
        Ash[:, 1:] = Av[:, :-1]
    else:
        Ash[:, :-1] = Av[:, 1:]
    ref = Ash.reshape(Bn * T, M).t() @ Bm.double()
    C = torch.zeros(M, N, device='cuda')
    hip.gemm_tn(A.cuda(), Bm.cuda(), C, M, N, Bn * T, a_shift=shift, period=T, split_k=2)
    _close(C, ref)


def test_gemm_tn_batched():
    from phones_las_amd import hip
    nb, K, M, N = 4, 23, 40, 136
    A, B = _mk((nb, K, M), 11), _mk((nb, K, N), 12)
    C = torch.zeros(nb, M, N, device='cuda')
    hip.gemm_tn(A.cuda(), B.cuda(), C, M, N, K, lda=M, ldb=N, ldc=N, batch=nb, sa=K * M, sb=K * N, sc=M * N)
    _close(C, torch.einsum('bkm,bkn->bmn', A.double(), B.double()))


def test_cast_and_colsum():
    from phones_las_amd import hip
    src = torch.randn(37, 50)
    dst = torch.full((56, 40), 9.0, dtype=torch.bfloat16, device='cuda')
    hip.cast_bf16(src.cuda(), 37, 50, dst, 56, 40, transpose=True)
    ref = torch.zeros(56, 40)
    ref[:50, :37] = src.t()
    assert torch.equal(dst.cpu().float(), ref.to(torch.bfloat16).float())
    dst2 = torch.full((37, 56), 9.0, dtype=torch.bfloat16, device='cuda')
    hip.cast_bf16(src.cuda(), 37, 50, dst2, 37, 56)
    ref2 = torch.zeros(37, 56)
    ref2[:, :50] = src
    assert torch.equal(dst2.cpu().float(), ref2.to(torch.bfloat16).float())
    X = _mk((1000, 72), 13)
    out = torch.zeros(72, device='cuda')
    hip.colsum_bf16(X.cuda(), 1000, 72, out)
    _close(out, X.double().sum(0), 1e-5)


def test_bad_arguments_raise():
    from phones_las_amd import hip
    A = torch.zeros(8, 12, dtype=torch.bfloat16, device='cuda')
    C = torch.zeros(8, 8, device='cuda')
    with pytest.raises(hip.LasError):
        hip.gemm_nt(A, A, C, 8, 8, 12)          # K not a multiple of 8
    with pytest.raises(hip.LasError):
        hip.gemm_nt(A.cpu(), A, C, 8, 8, 8)     # CPU tensor: no fallback


@pytest.mark.parametrize('use_ws,wide', [(True, False), (True, True), (False, False)])
@pytest.mark.parametrize('D,H,B,T,shift', [(40, 64, 3, 20, -1), (128, 64, 2, 37, 1), (0, 128, 2, 16, -1), (40, 256, 8, 100, 1)])
def test_gemm_tn_lstm_fused_weight_gradients(D, H, B, T, shift, use_ws, wide):
    """las_gemm_tn_lstm: dK_x, dK_h (row-shifted h) and db of one LSTM direction in one product, against float64.
    dz arrives with gate-interleaved columns (u*4+g); outputs are in TF column order (g*H+u).  wide: the 128 x 512 output tiles,
    asked for per call through LAS_TN_SPLIT_WIDE in split_k (taken where 4H is a multiple of 512, ignored elsewhere)."""
    from phones_las_amd import hip
    K = B * T
    Dp = max(D, 8)
    x, y, dz = _mk((K, Dp), 3), _mk((K, 2 * H), 4), _mk((K, 8 * H), 5)      # y / dz hold two directions side by side
    yi, dzi = y[:, H:], dz[:, 4 * H:]                                         # the second direction's columns
    # reference in TF column order
    dz_tf = dzi.double().view(K, H, 4).permute(0, 2, 1).reshape(K, 4 * H)
    ysh = torch.zeros(K, H, dtype=torch.float64)
    for k in range(K):
        t = k % T + shift
        if 0 <= t < T:
            ysh[k] = yi[k + shift].double()
    ref_k = torch.cat([x[:, :D].double().t() @ dz_tf, ysh.t() @ dz_tf], 0)
    ref_b = dz_tf.sum(0)
    # outputs ACCUMULATE: start from a known non-zero state; split 11 leaves empty K slices on the small shapes
    gk = torch.full((D + H, 4 * H), 0.5, device='cuda')
    gb = torch.full((4 * H,), -0.25, device='cuda')
    xd, yd, dzd = x.cuda(), y.cuda(), dz.cuda()
    split = 11
    ws = None
    if use_ws:
        need = hip.lib().las_gemm_tn_lstm_workspace_bytes(D, H, split)
        assert need == 4 * split * (D + H + 1) * 4 * H
        ws = torch.full((need // 4,), float('nan'), device='cuda')           # every word must be overwritten before use
    hip.check(hip.lib().las_gemm_tn_lstm(hip.p(xd) if D else None, Dp, D, hip.addr(yd, H), 2 * H, H, shift, T,
                                         hip.addr(dzd, 4 * H), 8 * H, hip.p(gk), hip.p(gb), K, split | (0x10000 if wide else 0), hip.p(ws),
                                         hip.stream()))
    torch.cuda.synchronize()
    _close(gk - 0.5, ref_k)
    _close(gb + 0.25, ref_b)


def test_gemm_nt_ring_asymmetric_identity_strides_and_batch():
    """The LDS-DMA ring kernel (gemm_nt_ring_kernel: 256 x 128 tiles of v_mfma_f32_32x32x16_bf16): A = I against an
    asymmetric B catches a transposed or permuted C write and a wrong source-chunk swizzle exactly (integer data);
    sub-matrix views through lda / ldb / ldc; a batch of two; and the same product through the register-staged kernel
    (LAS_GEMM_RING is read once per process, so the reference here is the float64 product)."""
    from phones_las_amd import hip
    n = 1024
    A = torch.eye(n).to(torch.bfloat16).cuda()
    B = ((torch.arange(256 * n).reshape(256, n) * 7) % 253).float().to(torch.bfloat16)
    C = torch.empty(n, 256, device='cuda')
    hip.gemm_nt(A, B.cuda(), C, n, 256, n)
    assert torch.equal(C.cpu(), B.float().t())
    big = _mk((1500, 400), 3).cuda()
    W = _mk((200, 256), 4).cuda()
    out = torch.zeros(1500, 260, device='cuda')
    hip.gemm_nt(big[:, 64:], W[:, :192], out[:, 20:], 1500, 200, 192, lda=400, ldb=256, ldc=260)
    ref = big[:, 64:256].double().cpu() @ W[:, :192].double().cpu().t()
    _close(out[:, 20:220], ref)
    assert float(out[:, :20].abs().max()) == 0.0 and float(out[:, 220:].abs().max()) == 0.0
    nb, M, N, K = 2, 1100, 136, 128
    Ab, Bb = _mk((nb, M, K), 5), _mk((nb, N, K), 6)
    Cb = torch.empty(nb, M, N, device='cuda')
    hip.gemm_nt(Ab.cuda(), Bb.cuda(), Cb, M, N, K, lda=K, ldb=K, ldc=N, batch=nb, sa=M * K, sb=N * K, sc=M * N)
    _close(Cb, torch.einsum('bmk,bnk->bmn', Ab.double(), Bb.double()))


@pytest.mark.parametrize('M,N,K', [(1024, 256, 128), (1300, 512, 384), (2049, 256, 2048), (4096, 1024, 512), (5000, 256, 4096)])
def test_gemm_nt_with_the_weight_operand_from_its_image(M, N, K):
    """las_gemm_nt_bimg (round 6): the weight operand as its LAS_IMAGE_PACK_MFMA_B image, fetched into registers fragment by fragment
    (every load of the K loop is hand-counted inline assembly -- several repetitions: a fragment consumed before it has landed shows
    up as a wrong sum now and then).  One to 32 rounds of the four-stage loop, ragged M, bias, accumulate; image from a bf16 matrix
    (las_pack_mfma_b_bf16) and from fp32 through the image job (the [N, K] window, and the two K halves of one image)."""
    from phones_las_amd import hip
    lib = hip.lib()
    A, W = _mk((M, K), 3), _mk((N, K), 4) * 0.25
    bias = torch.randn(N)
    ref = A.double() @ W.double().t() + bias.double()
    Ad, Wd = A.cuda(), W.cuda()
    img = torch.empty(N * K, dtype=torch.bfloat16, device='cuda')
    hip.check(lib.las_pack_mfma_b_bf16(hip.p(Wd), K, N, K, hip.p(img), hip.stream()))
    img2 = torch.empty_like(img)
    hip.pack_mfma_b(Wd.float(), N, K, img2, lds=K)
    assert torch.equal(img, img2)
    img3 = torch.zeros_like(img)                        # two windows of one image: the K halves
    with hip.image_batch():
        hip.pack_mfma_b(Wd.float(), N, K // 2, img3, lds=K, image_k=K, k0=0, dst_rows=N, dst_cols=K // 2)
        hip.pack_mfma_b(Wd.float()[:, K // 2:], N, K // 2, img3, lds=K, image_k=K, k0=K // 2, dst_rows=N, dst_cols=K // 2)
    assert torch.equal(img, img3)
    for rep in range(6):
        C = torch.full((M, N), float('nan'), device='cuda')
        hip.gemm_nt_bimg(Ad, img, C, M, N, K, bias=bias.cuda())
        _close(C, ref)
    C2 = torch.ones(M, N, device='cuda')
    hip.gemm_nt_bimg(Ad, img, C2, M, N, K, accumulate=True)
    _close(C2, ref - bias.double() + 1.0)
    with pytest.raises(hip.LasError):
        hip.gemm_nt_bimg(Ad, img, C, M, N - 16, K)


def test_weight_image_with_the_gate_interleaving():
    """The image job's perm_h (the K_x images of las/ops.py LayerWeights): source column g * H + u read at logical index u * 4 + g, on
    the image's row axis (transpose: x K_x) and on its K axis (a K window per direction: dZ K_x^T) -- against the images of the
    bf16 operand copies the ring kernels read (LAS_IMAGE_CAST with the same permutation)."""
    from phones_las_amd import hip
    lib = hip.lib()
    D, H, nd = 256, 64, 2
    g = torch.Generator().manual_seed(5)
    kernels = [torch.randn(D + H, 4 * H, generator=g).cuda() for _ in range(nd)]
    kxT = torch.empty(nd * 4 * H, D, dtype=torch.bfloat16, device='cuda')
    kx = torch.empty(D, nd * 4 * H, dtype=torch.bfloat16, device='cuda')
    kxT_img = torch.empty(nd * 4 * H * D, dtype=torch.bfloat16, device='cuda')
    kx_img = torch.empty(D * nd * 4 * H, dtype=torch.bfloat16, device='cuda')
    with hip.image_batch():
        for i, k in enumerate(kernels):
            hip.cast_bf16(k, D, 4 * H, kxT[i * 4 * H:], 4 * H, D, ldd=D, transpose=True, lds=4 * H, perm_h=H)
            hip.cast_bf16(k, D, 4 * H, kx[:, i * 4 * H:], D, 4 * H, ldd=nd * 4 * H, lds=4 * H, perm_h=H)
            hip.pack_mfma_b(k, 4 * H, D, kxT_img[i * 4 * H * D:(i + 1) * 4 * H * D], lds=4 * H, transpose=True, perm_h=H, dst_rows=4 * H, dst_cols=D)
            hip.pack_mfma_b(k, D, 4 * H, kx_img, lds=4 * H, perm_h=H, image_k=nd * 4 * H, k0=i * 4 * H, dst_rows=D, dst_cols=4 * H)
    for mat, img in ((kxT, kxT_img), (kx, kx_img)):
        ref = torch.empty_like(img)
        hip.check(lib.las_pack_mfma_b_bf16(hip.p(mat), mat.shape[1], mat.shape[0], mat.shape[1], hip.p(ref), hip.stream()))
        assert torch.equal(img, ref)
