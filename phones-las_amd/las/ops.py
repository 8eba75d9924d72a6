"""MI355X host side of the reference's ``las/ops.py`` (lstm_cell, bilstm, pyramidal_bilstm).

Same names and argument meaning as /root/reference/las/ops.py:10-87; the TF graph ops are replaced by
calls into liblas_hip.so (hand-written HIP, include/las_hip.h).  TF builds variables implicitly inside
variable scopes; here the caller passes ``variables`` (name -> fp32 CUDA tensor, TF names/layouts) and,
for training, a ``tape`` list that receives one record per layer for ``*_backward``.

Layout contract: inputs [B,T,D] bf16 with D a multiple of 8 (zero padded), T even wherever a pyramid
stack follows; outputs [B,T,ndir*H] bf16 with fw in [0,H) and bw in [H,2H) — the tf.concat of
las/ops.py:81 is free.  pyramidal_stack (las/ops.py:49-65) is then a zero-copy view.
"""
import collections
import os

import torch

from .. import hip

__all__ = ['lstm_cell', 'bilstm', 'pyramidal_bilstm', 'pyramidal_stack', 'LSTMStateTuple', 'TRAIN', 'EVAL',
           'PREDICT']

TRAIN, EVAL, PREDICT = 'train', 'eval', 'infer'      # tf.estimator.ModeKeys values
LSTMStateTuple = collections.namedtuple('LSTMStateTuple', ['c', 'h'])
LSTMCellSpec = collections.namedtuple('LSTMCellSpec', ['num_units', 'input_keep_prob'])


def lstm_cell(num_units, dropout, mode):
    """las/ops.py:10-20: LSTMCell(num_units, U(-0.075,0.075)); DropoutWrapper(input_keep_prob) in TRAIN."""
    dropout = dropout if mode == TRAIN else 0.0
    if not 0.0 <= dropout < 1.0:
        raise ValueError('dropout must be in [0, 1)')
    if num_units not in (64, 128, 256, 512, 1024):
        raise ValueError('the recurrent kernels are built for 64, 128, 256, 512 and 1024 units (got %d): LasModel runs other '
                         'widths zero-padded to the next of these (model_helper.physical_params)' % num_units)
    return LSTMCellSpec(num_units, 1.0 - dropout)


_WORKSPACES = {}


def lstm_workspace(B, H, nd):
    """Scratch for the cooperative recurrent kernels (status word + exchange buffer), cached per shape."""
    key = (B, H, nd, torch.cuda.current_device())
    ws = _WORKSPACES.get(key)
    if ws is None:
        n = hip.lib().las_lstm_workspace_bytes(B, H, nd)
        if n == 0:
            raise ValueError('num_units %d is not supported by the HIP recurrent kernels' % H)
        ws = torch.zeros(n, dtype=torch.uint8, device='cuda')
        _WORKSPACES[key] = ws
    return ws


def _note_timeout(st):
    """A forward launch that timed out while its input products were being streamed beside it: the hand-over needs the two
    kernels to run concurrently; if this process cannot give them that (see _product_stream), the steps that follow fall
    back to the product before the recurrence instead of failing one after the other."""
    global STREAM_X
    if (st & 1) and STREAM_X and _PRODUCT_STREAMS:
        STREAM_X = False
        import sys
        print('phones_las_amd: a recurrent forward launch timed out beside its streamed input product; streaming is off for the '
              'rest of this process (LAS_LSTM_STREAM=0 has the same effect)', file=sys.stderr)


def check_lstm_status(B, H, nd):
    """Raise if a bounded inter-workgroup wait of the last recurrent launch timed out (forces a sync)."""
    st = int(lstm_workspace(B, H, nd)[:4].view(torch.int32).item())
    if st:
        _note_timeout(st)
        raise hip.LasError('recurrent kernel reported an inter-workgroup timeout (status %d)' % st)


def check_all_lstm_status():
    """Raise if any recurrent launch since the last check reported an inter-workgroup timeout (forces a sync)."""
    for (B, H, nd, dev), ws in _WORKSPACES.items():
        st = int(ws[:4].view(torch.int32).item())
        if st:
            _note_timeout(st)
            raise hip.LasError('recurrent kernel (B=%d, H=%d) reported an inter-workgroup timeout (status %d)' % (B, H, st))


# workgroups the fused weight-gradient product of a direction aims for (K slices x output tiles); LAS_TN_WGS overrides
TN_WORKGROUPS = int(os.environ.get('LAS_TN_WGS', '704'))
# ... of the bottom layer, whose two directions' products run ALONE on the chip (nothing hides them), at the same time on two
# streams; a 512-thread workgroup with 96 KiB of LDS takes a CU
TN_WORKGROUPS_EXPOSED = int(os.environ.get('LAS_TN_WGS_EXPOSED', '704'))
TN_WIDE = os.environ.get('LAS_TN_WIDE', '1') != '0'         # (the library reads the same switch: 128 x 512 tiles of the fused product)
TN_SPLIT_WIDE = 0x10000                                      # LAS_TN_SPLIT_WIDE of las_hip.h: the tile shape travels in split_k


class Overlap:
    """A second HIP stream for launches that are off the critical path of the backward pass (weight-gradient
    GEMMs, bias sums): they fill the ~224 CUs the persistent recurrent kernels leave idle.  fork() makes the side
    stream wait for everything enqueued so far; join() makes the main stream wait for the side stream.  Tensors
    the side stream reads are kept alive until join()."""

    def __init__(self):
        self.side = torch.cuda.Stream()
        self.side2 = None           # a third stream, created on first use (fork(..., lane=1))
        self.side2_busy = False     # something was forked onto side2 since the last join()
        self.keep = []

    # how long side work that is going to run BESIDE a recurrent launch is held back (see fork)
    BESIDE_US = int(os.environ.get('LAS_SIDE_DELAY_US', '12'))

    def fork(self, *tensors, lane=0, beside_chain=False):
        """lane 1: another side stream, for two pieces of side work that should run at the same time (the two directions'
        weight-gradient products of the bottom listener layer, which nothing hides).
        beside_chain: the next launch on the main stream is a persistent recurrent kernel and this side work becomes
        runnable at the same moment: it is held back a few microseconds (las_stream_delay) so that the chain's workgroups
        are resident first -- a chain that finds CUs taken by a GEMM workgroup starts late as a whole (measured: 1.03-1.07
        instead of 0.96-1.00 ms for an 800-step backward launch when the product beside it was dispatched first)."""
        if lane and self.side2 is None:
            # (an UNPROBED stream: one that is probed to run beside the first side stream -- streams share four hardware queues --
            # measured the same in round 5, 6.012 against 6.024 ms: side by side the two exposed products take as long as one
            # after the other, and a process whose streams hold all four queues leaves none for RCCL's)
            self.side2 = torch.cuda.Stream()
        side = self.side2 if lane else self.side
        if lane:
            self.side2_busy = True
        side.wait_stream(torch.cuda.current_stream())
        self.keep.extend(tensors)
        if beside_chain and self.BESIDE_US > 0:
            hip.check(hip.lib().las_stream_delay(self.BESIDE_US, side.cuda_stream))
        import contextlib
        ctx = contextlib.ExitStack()
        ctx.enter_context(torch.cuda.stream(side))
        ctx.enter_context(hip.ws_lane(2 if lane else 1))       # the side streams' own product workspaces
        return ctx

    def mark(self):
        """Events after everything forked so far, on every side stream in use; .wait() makes the current stream wait for exactly
        that much of them."""
        streams = [self.side] + ([self.side2] if self.side2_busy else [])     # (an idle side2 holds nothing to wait for -- and
        return _Marks([s.record_event() for s in streams])                    #  may not be part of a graph capture yet)

    def join(self):
        torch.cuda.current_stream().wait_stream(self.side)
        if self.side2 is not None:
            torch.cuda.current_stream().wait_stream(self.side2)
        self.side2_busy = False
        self.keep.clear()

    def join_side(self):
        """The main stream waits for what the first side stream holds so far (nothing is released: join() does that)."""
        torch.cuda.current_stream().wait_stream(self.side)


class _Marks:
    def __init__(self, events):
        self.events = events

    def wait(self):
        for ev in self.events:
            torch.cuda.current_stream().wait_event(ev)


class _NoOverlap:
    """Same interface, everything on the current stream."""

    def fork(self, *tensors, lane=0, beside_chain=False):
        import contextlib
        return contextlib.nullcontext()

    def mark(self):
        return _Marks([])

    def join(self):
        pass

    def join_side(self):
        pass


# LAS_LSTM_FUSED_X=0: the bottom layer's input projection as a separate product again (diagnostics, A/B timing)
FUSED_X = os.environ.get('LAS_LSTM_FUSED_X', '1') != '0'
# LAS_LSTM_STREAM=0: the upper layers' input products run to completion BEFORE their recurrence again.  Default (256-unit
# layers; at 512 units the product beside the chain cost more than it hid: metric-L 17.83 against 17.69 ms): the product
# (las_gemm_nt_stream) runs on a second stream BESIDE the recurrence and hands its rows over step block by step block.
STREAM_X = os.environ.get('LAS_LSTM_STREAM', '1') != '0'
STREAM_512 = os.environ.get('LAS_LSTM_STREAM_512', '0') != '0'      # (diagnostics: stream at 512 units too)
STREAM_MIN_ROWS = int(os.environ.get('LAS_LSTM_STREAM_MIN_ROWS', '4096'))      # smaller products are not worth the hand-over
STREAM_ALWAYS = os.environ.get('LAS_LSTM_STREAM_ALWAYS', '0') != '0'           # (diagnostics / tests: stream whatever the recurrence's length)
STREAM_ROWS8_MIN_K = int(os.environ.get('LAS_STREAM_ROWS8_MIN_K', str(1 << 30)))      # (diagnostics, see _stream_setup)
STREAM_MAX_WORKGROUPS = 192       # the recurrence (members + companions, a CU each) must leave CUs to the product beside it
_PRODUCT_STREAMS = {}
_ALIASED_STREAMS = []


def _product_stream():
    """The stream the streamed input products run on: one per (device, LAUNCHING stream); its work is always joined by the
    launching stream.  The recurrence waits INSIDE its kernel for tiles this stream's kernel produces, so the two must never share
    a hardware queue -- a process that has created many streams gets them mapped onto a handful of queues, and a product queued
    behind the recurrence that waits for it is a (bounded) deadlock: every streamed launch timed out in a test session that had
    created ~100 streams before this one.  (A high-priority stream has queues of its own but its workgroups are placed before the
    recurrence's: measured, 41 ms per step.)  So the stream is PROBED when it is created (_concurrent_stream) -- against the
    stream the recurrence will be launched on, which is why the cache is keyed by it (round 5: bench.StepForms of a second
    model warmed up on another side stream than the first one's, the cached product stream shared THAT stream's queue, and its
    streamed launches timed out: found by tests/test_gpu_step_forms.py).  During a graph capture nothing can be probed: the
    capture takes the stream that an eager step on this device found (callers run the step eagerly first)."""
    dev = torch.cuda.current_device()
    cur = torch.cuda.current_stream()
    key = (dev, cur.cuda_stream)
    st = _PRODUCT_STREAMS.get(key)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            st = next((v for (d, _), v in _PRODUCT_STREAMS.items() if d == dev), None)
            if st is None:
                st = torch.cuda.Stream()
        else:
            # a stream that already serves another launching stream of this device is tried first (fewer streams, fewer queues)
            words = torch.zeros(2, dtype=torch.int32, device='cuda')
            for (d, _), cand in list(_PRODUCT_STREAMS.items()):
                if d != dev or cand.cuda_stream == cur.cuda_stream:
                    continue
                hip.check(hip.lib().las_stream_concurrency_probe(cur.cuda_stream, cand.cuda_stream, hip.p(words), 2000))
                cur.wait_stream(cand)
                if int(words[1].item()) == 1:
                    st = cand
                    break
            if st is None:
                st = _concurrent_stream()
        _PRODUCT_STREAMS[key] = st
    return st


def _stream_beside(others, tries=16):
    """A new stream whose kernels run beside kernels of every stream in `others` (las_stream_concurrency_probe); the last
    candidate when none is found (correct all the same: work is only ordered more than it had to be)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream()               # (cannot probe inside a capture: an eager step creates the stream first)
    words = torch.zeros(2, dtype=torch.int32, device='cuda')
    cur = torch.cuda.current_stream()
    cand = None
    for _ in range(tries):
        cand = torch.cuda.Stream()
        good = True
        for o in others:
            if o.cuda_stream == cand.cuda_stream:
                good = False
                break
            o.wait_stream(cur)                   # (`words` was cleared on the current stream)
            hip.check(hip.lib().las_stream_concurrency_probe(o.cuda_stream, cand.cuda_stream, hip.p(words), 2000))
            o.wait_stream(cand)
            cur.wait_stream(o)
            if int(words[1].item()) != 1:
                good = False
                break
        if good:
            return cand
        _ALIASED_STREAMS.append(cand)
    return cand


def _concurrent_stream(tries=8):
    """A stream whose kernels run BESIDE kernels enqueued earlier on the current stream (las_stream_concurrency_probe), or None.
    A new stream takes the next hardware queue in turn, so a few tries find one that is not the current stream's; the streams
    that failed are kept (dropping them would hand their queue to the next try)."""
    global STREAM_X
    if torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream()               # (cannot probe inside a capture: callers create the stream in an eager step first)
    main = torch.cuda.current_stream()
    words = torch.zeros(2, dtype=torch.int32, device='cuda')
    for _ in range(tries):
        cand = torch.cuda.Stream()
        hip.check(hip.lib().las_stream_concurrency_probe(main.cuda_stream, cand.cuda_stream, hip.p(words), 2000))
        main.wait_stream(cand)
        if int(words[1].item()) == 1:
            return cand
        _ALIASED_STREAMS.append(cand)
    import sys
    print('phones_las_amd: no stream found that runs beside the current one (%d tried): input products are not streamed in this '
          'process' % tries, file=sys.stderr)
    STREAM_X = False
    return torch.cuda.Stream()


def _dirs(unidirectional):
    return ['fw'] if unidirectional else ['fw', 'bw']


class LayerWeights:
    """bf16 operand images of one (Bi)LSTM layer's fp32 master weights (rebuilt after each update)."""

    def __init__(self, variables, scope, D, Dp, H, unidirectional, cell_path='/{dir}/lstm_cell'):
        self.D, self.Dp, self.H = D, Dp, H
        self.dirs = _dirs(unidirectional)
        nd = len(self.dirs)
        dev = 'cuda'
        self.names = [(scope + cell_path.format(dir=d) + '/kernel', scope + cell_path.format(dir=d) + '/bias')
                      for d in self.dirs]
        self.kxT = torch.empty(nd * 4 * H, Dp, dtype=torch.bfloat16, device=dev)       # B^T-form for x*K_x
        self.kx = torch.empty(max(D, 1), nd * 4 * H, dtype=torch.bfloat16, device=dev)  # B^T-form for dZ*K_x^T
        self.kh = torch.empty(nd, H, 4 * H, dtype=torch.bfloat16, device=dev)           # K_h, interleaved columns (bwd)
        self.khp = torch.empty(nd, H * 4 * H, dtype=torch.bfloat16, device=dev)         # fragment-major (fwd)
        self.bias = torch.empty(nd * 4 * H, dtype=torch.float32, device=dev)
        # fused input projection (las_lstm_recurrent_fwd_x): narrow inputs -- the bottom layer's features -- enter the recurrent
        # kernel itself; K_x as register-resident fragment images.  LAS_LSTM_FUSED_X=0 keeps the separate x K_x product.
        self.kx_chunks = hip.lib().las_lstm_fused_input_chunks(H, Dp) if (FUSED_X and D > 0) else 0
        self.kxp = (torch.empty(nd, (H // 16) * self.kx_chunks * 4 * 512, dtype=torch.bfloat16, device=dev)
                    if self.kx_chunks else None)
        # las_gemm_nt_bimg (round 6): K_x as LAS_IMAGE_PACK_MFMA_B images -- [nd * 4H, Dp] for x K_x, [D, nd * 4H] for dZ K_x^T --
        # where the shapes are ones that kernel takes and wins on (a wave fetches the weight fragments straight into registers)
        N4 = nd * 4 * H
        self.kxT_img = (torch.empty(N4 * Dp, dtype=torch.bfloat16, device=dev)
                        if (not self.kx_chunks and D == Dp and hip.gemm_nt_bimg_wanted(1 << 20, N4, Dp)) else None)
        self.kx_img = (torch.empty(D * N4, dtype=torch.bfloat16, device=dev)
                       if (D > 0 and D == Dp and hip.gemm_nt_bimg_wanted(1 << 20, D, N4)) else None)
        self.refresh(variables)

    def refresh(self, variables):
        D, Dp, H = self.D, self.Dp, self.H
        nd = len(self.dirs)
        for i, (kn, bn) in enumerate(self.names):
            k, b = variables[kn], variables[bn]
            assert k.shape == (D + H, 4 * H), (kn, tuple(k.shape), (D + H, 4 * H))
            # gate-interleaved column order (u*4+g) for everything the recurrent kernels touch per step
            hip.cast_bf16(k, D, 4 * H, self.kxT[i * 4 * H:], 4 * H, Dp, ldd=Dp, transpose=True, lds=4 * H, perm_h=H)
            hip.cast_bf16(k, D, 4 * H, self.kx[:, i * 4 * H:], D, 4 * H, ldd=nd * 4 * H, lds=4 * H, perm_h=H)
            hip.cast_bf16(k[D:], H, 4 * H, self.kh[i], H, 4 * H, ldd=4 * H, lds=4 * H, perm_h=H)
            hip.pack_recurrent(k[D:], H, self.khp[i])
            if self.kx_chunks:
                hip.pack_input(k, D, H, self.kx_chunks, self.kxp[i])
            if self.kxT_img is not None:        # rows [i * 4H, (i + 1) * 4H) of the image: row n = gate-interleaved column of K_x, k = input feature
                hip.pack_mfma_b(k, 4 * H, D, self.kxT_img[i * 4 * H * Dp:(i + 1) * 4 * H * Dp], lds=4 * H, transpose=True, perm_h=H,
                                dst_rows=4 * H, dst_cols=Dp)
            if self.kx_img is not None:         # K range [i * 4H, (i + 1) * 4H) of the image: row n = input feature, k = gate-interleaved column
                hip.pack_mfma_b(k, D, 4 * H, self.kx_img, lds=4 * H, perm_h=H, image_k=nd * 4 * H, k0=i * 4 * H, dst_rows=D, dst_cols=4 * H)
            hip.bias_interleave(b, H, self.bias[i * 4 * H:(i + 1) * 4 * H])


def _stream_setup(a, lda, a_dir_stride, weights, xproj, sequence_length, B, T, H, Dp, nd, dev):
    """x K_x BESIDE the recurrence (las_gemm_nt_stream[_dirs]) when the shapes allow: returns (ready flags, column tiles per
    direction, launch_product) or None.  a_dir_stride: 0 = both directions read `a`; else direction d reads a + d * stride."""
    lib = hip.lib()
    # the recurrence's slices must not straddle the product's 16-utterance blocks, and the chain must leave CUs to the product
    # ... and the recurrence must be long enough for the product to finish beside it: the product gets the third of the chip the
    # chains leave (~0.28 PFLOP/s); beside a recurrence shorter than ~1.6 x that it is the recurrence that waits (metric-M's layer 2,
    # T = 400: 0.49 ms streamed against 0.33 + 0.14 one after the other since the image kernel; layer 1, T = 800: 0.79 against 0.84)
    long_enough = STREAM_ALWAYS or T * 0.835e-6 > 1.6 * (2.0 * B * T * nd * 4 * H * Dp) / 0.28e15
    streamed = (STREAM_X and long_enough and (H == 256 or (H == 512 and STREAM_512)) and B * T >= STREAM_MIN_ROWS and lib.las_gemm_nt_stream_supported(nd * 4 * H, Dp, nd) == 1
                and 16 % max(lib.las_lstm_slice_rows(B, H, nd), 1) == 0
                and 0 < lib.las_lstm_fwd_workgroups(B, H, nd) <= STREAM_MAX_WORKGROUPS)
    if not streamed:
        return None
    _product_stream()                 # (created -- and probed against the current stream -- before the first streamed launch)
    if not STREAM_X:
        return None
    # counters zeroed here, the product on its own stream (held back a few microseconds so that the chain's workgroups are
    # resident first), the recurrence consumes the rows as they become visible
    rows = lib.las_lstm_slice_rows(B, H, nd)
    # (LAS_STREAM_ROWS8_MIN_K=k: 8-row chain slices under products with K >= k -- half the chain workgroups, more CUs for the
    # product.  Measured in round 5, metric-M's layer 2 (K = 1024): 0.523 ms against 0.501 on 4-row slices, the step 6.08 against
    # 6.05: the product beside the chain is not short of CUs.  Off by default.)
    if rows == 4 and Dp >= STREAM_ROWS8_MIN_K:
        rows = 8
    n = lib.las_gemm_nt_stream_flags(B, T, nd, rows)
    ready = weights.__dict__.setdefault('_ready', {}).get((B, T, rows))
    if ready is None:
        ready = weights._ready[(B, T, rows)] = torch.zeros(n, dtype=torch.int32, device=dev)
    hip.fill_many(zero=[ready])
    cleared = torch.cuda.Event()
    cleared.record()

    def launch_product():
        # (called once the recurrence has been ENQUEUED: an idle GPU must not start the product first -- its persistent
        # workgroups would hold every CU while they wait for the recurrence to say where its groups run)
        side = _product_stream()
        side.wait_event(cleared)
        with torch.cuda.stream(side):
            hip.check(lib.las_stream_delay(Overlap.BESIDE_US, hip.stream()))
            tok = hip.prof_begin('gemm_nt', 2.0 * B * T * nd * 4 * H * Dp)
            hip.check(lib.las_gemm_nt_stream_dirs(hip.p(a), lda, a_dir_stride, hip.p(weights.kxT), Dp, hip.p(xproj), nd * 4 * H,
                                                  hip.p(weights.bias), hip.p(sequence_length), B, T, nd * 4 * H, Dp, nd, rows,
                                                  hip.p(ready), hip.stream()))
            hip.prof_end(tok)
        return side
    return (ready, 4 * H // 128, launch_product, rows)


def bilstm(inputs, sequence_length, num_units, dropout, mode, unidirectional=False, *, variables=None,
           scope='', weights=None, tape=None, in_features=None, rng=None, split_inputs=False, after_projection=None):
    """las/ops.py:23-46.  inputs [B,T,Dp] bf16; sequence_length int32 [B] (CUDA).
    Returns (outputs, state) with the reference's structure: bidirectional -> ((fw, bw), (state_fw,
    state_bw)); unidirectional -> (fw, state).  fw/bw are views of one [B,T,ndir*H] buffer
    (use ``concat_outputs`` for the tf.concat of las/ops.py:81).
    split_inputs: direction i reads only columns [i*D, (i+1)*D) of ``inputs`` (the per-direction MultiRNNCell
    stacks of the non-pyramidal listener, las/model.py:111-133)."""
    cell = lstm_cell(num_units, dropout, mode)
    keep = cell.input_keep_prob
    B, T, Dfull = inputs.shape
    H = num_units
    nd_in = 1 if unidirectional else 2
    Dp = Dfull // nd_in if split_inputs else Dfull
    D = in_features if in_features is not None else Dp
    if weights is None:
        weights = LayerWeights(variables, scope, D, Dp, H, unidirectional)
    nd = len(weights.dirs)
    dev = inputs.device
    xproj = torch.empty(B, T, nd * 4 * H, dtype=torch.float32, device=dev)
    dropped = None
    stream_ready = None
    # narrow inputs (the features): x_t K_x + b is formed inside the recurrent kernel -- no product, no fp32 round trip
    fused = (weights.kx_chunks > 0 and not split_inputs and Dp == weights.Dp and
             (keep == 1.0 or nd == 1 or (nd == 2 and Dp % 8 == 0)))
    fused_x = None                    # (x, ldx, stride between the directions' copies)
    if fused and keep == 1.0:
        fused_x = (inputs, Dfull, 0)
    elif keep < 1.0 or split_inputs:
        # one A operand per direction: DropoutWrapper(input_keep_prob) draws independent masks for the fw and bw
        # cells, fresh per time step (las/ops.py:14-18); split_inputs gives each direction its own columns
        if keep < 1.0 and rng is None:
            raise ValueError('dropout needs rng=(seed, first_stream_id)')
        dropped = []
        pair = None
        if keep < 1.0 and nd == 2 and not split_inputs and Dp % 8 == 0:
            # both cells read the same input through their own masks: the two copies in one pass over it
            seed, stream0 = rng
            pair = torch.empty(2, B, T, Dp, dtype=torch.bfloat16, device=dev)
            hip.check(hip.lib().las_dropout_bf16_pair(hip.p(inputs), Dfull, hip.p(pair[0]), hip.p(pair[1]), Dp, B * T, Dp, keep,
                                                      seed, stream0, stream0 + 1, hip.stream()))
        # one streamed product for both directions where their operands lie a fixed stride apart: the two masked copies
        # (pair), or the directions' own column ranges of the layer below (split_inputs without dropout)
        if not fused:
            if pair is not None:
                stream_ready = _stream_setup(pair, Dp, B * T * Dp, weights, xproj, sequence_length, B, T, H, Dp, nd, dev)
            elif split_inputs and keep == 1.0 and nd == 2:
                stream_ready = _stream_setup(inputs, Dfull, Dp, weights, xproj, sequence_length, B, T, H, Dp, nd, dev)
        for i in range(nd):
            src = inputs[..., i * Dp:(i + 1) * Dp] if split_inputs else inputs
            if pair is not None:
                a, lda = pair[i], Dp
            elif keep < 1.0:
                seed, stream0 = rng
                xd = torch.empty(B, T, Dp, dtype=torch.bfloat16, device=dev)
                hip.check(hip.lib().las_dropout_bf16(hip.p(src), Dfull, hip.p(xd), Dp, B * T, Dp, keep, seed,
                                                     stream0 + i, hip.stream()))
                a, lda = xd, Dp
            else:
                a, lda = src, Dfull
            if not fused and stream_ready is None:
                if weights.kxT_img is not None and hip.gemm_nt_bimg_wanted(B * T, 4 * H, Dp):
                    # (direction i's rows of the image are one contiguous block: the image is row-tile major)
                    hip.gemm_nt_bimg(a, weights.kxT_img[i * 4 * H * Dp:(i + 1) * 4 * H * Dp], xproj[..., i * 4 * H:], B * T, 4 * H, Dp,
                                     lda=lda, ldc=nd * 4 * H, bias=weights.bias[i * 4 * H:])
                else:
                    hip.gemm_nt(a, weights.kxT[i * 4 * H:], xproj[..., i * 4 * H:], B * T, 4 * H, Dp, lda=lda, ldb=Dp,
                                ldc=nd * 4 * H, bias=weights.bias[i * 4 * H:])
            dropped.append((a, lda))
        if fused:
            fused_x = (dropped[0][0], Dp, B * T * Dp if nd == 2 else 0)
    else:
        stream_ready = _stream_setup(inputs, Dp, 0, weights, xproj, sequence_length, B, T, H, Dp, nd, dev)
        if stream_ready is None:
            if weights.kxT_img is not None and hip.gemm_nt_bimg_wanted(B * T, nd * 4 * H, Dp):
                hip.gemm_nt_bimg(inputs, weights.kxT_img, xproj, B * T, nd * 4 * H, Dp, lda=Dp, ldc=nd * 4 * H, bias=weights.bias)
            else:
                hip.gemm_nt(inputs, weights.kxT, xproj, B * T, nd * 4 * H, Dp, lda=Dp, ldb=Dp, ldc=nd * 4 * H,
                            bias=weights.bias)
    if after_projection is not None:
        after_projection()          # (LasModel: side-stream work that should run beside this layer's recurrence starts here)
    y = torch.empty(B, T, nd * H, dtype=torch.bfloat16, device=dev)
    cbuf = torch.empty(B, T, nd * H, dtype=torch.float32, device=dev)
    c_last = torch.empty(nd, B, H, dtype=torch.float32, device=dev)
    h_last = torch.empty(nd, B, H, dtype=torch.float32, device=dev)
    a = hip.LstmFwd()
    a.xproj, a.wpacked, a.length, a.y, a.cbuf = hip.addr(xproj), hip.addr(weights.khp), hip.addr(sequence_length), hip.addr(y), hip.addr(cbuf)
    a.c_last, a.h_last, a.workspace = hip.addr(c_last), hip.addr(h_last), hip.addr(lstm_workspace(B, H, nd))
    a.B, a.T, a.H, a.ndir = B, T, H, nd
    flops = 2.0 * B * T * nd * H * 4 * H                                   # the recurrent product h_{t-1} K_h of every step
    if fused_x is not None:
        xa, ldx, xdir = fused_x
        a.x, a.ldx, a.x_dir_stride, a.Dp = hip.addr(xa), ldx, xdir, Dp
        a.kx_packed, a.bias = hip.addr(weights.kxp), hip.addr(weights.bias)
        flops += 2.0 * B * T * nd * Dp * 4 * H                            # x_t K_x of every step as well
    elif stream_ready is not None:
        a.ready, a.ready_count, a.rows_per_slice = hip.addr(stream_ready[0]), stream_ready[1], stream_ready[3]
    tok = hip.prof_begin('lstm_fwd', flops)
    import ctypes
    hip.check(hip.lib().las_lstm_recurrent_fwd_ex(ctypes.byref(a), hip.stream()))
    hip.prof_end(tok)
    if stream_ready is not None:
        # (the product is done long before the recurrence is; everything later on this stream is ordered behind both,
        # so the operands need no record_stream)
        torch.cuda.current_stream().wait_stream(stream_ready[2]())
    if tape is not None:
        tape.append(dict(kind='bilstm', inputs=inputs, length=sequence_length, gates=xproj, cbuf=cbuf, y=y,
                         weights=weights, B=B, T=T, H=H, D=D, Dp=Dp, nd=nd, dropped=dropped, keep=keep, rng=rng,
                         split=split_inputs, Dfull=Dfull))
    states = tuple(LSTMStateTuple(c_last[i], h_last[i]) for i in range(nd))
    if unidirectional:
        return y, states[0]
    return (y[..., :H], y[..., H:]), states


def concat_outputs(outputs):
    """tf.concat(outputs, -1) of las/ops.py:81 — free: both halves already live in one buffer."""
    if isinstance(outputs, tuple):
        fw, bw = outputs
        base = fw._base
        if base is None or bw._base is not base or base.shape[-1] != fw.shape[-1] + bw.shape[-1]:
            return torch.cat(outputs, -1)
        return base
    return outputs


# LAS_MASKED_DX=0: the input-dropout backward as its own pass over two partial dX buffers again (diagnostics, A/B timing)
MASKED_DX = os.environ.get('LAS_MASKED_DX', '1') != '0'


def bilstm_backward(rec, dy, d_state, grads, need_dx=True, overlap=None, defer_weight_grads=False, exposed=False):
    """Reverse-mode AD of one bilstm() call.  dy [B,T,nd*H] fp32 (gradient of the concatenated outputs),
    d_state: None or (dc_last, dh_last) each [nd,B,H] fp32.  Accumulates into ``grads`` (name -> fp32
    tensor, same shapes as the variables) and returns dX [B,T,D'] fp32 (or None).  defer_weight_grads: returns
    (dX, launch) instead and leaves the weight-gradient products to launch() (the caller forks them later).
    exposed: no recurrence follows that would hide the weight-gradient products (the bottom layer): the two directions'
    products then run on two streams at once instead of one after the other."""
    B, T, H, D, Dp, nd = rec['B'], rec['T'], rec['H'], rec['D'], rec['Dp'], rec['nd']
    w = rec['weights']
    dev = dy.device
    dz = torch.empty(B, T, nd * 4 * H, dtype=torch.bfloat16, device=dev)
    dc_last, dh_last = d_state if d_state is not None else (None, None)
    x, y = rec['inputs'], rec['y']
    BT = B * T
    Df_w = D if D % 8 == 0 else (Dp if (Dp % 8 == 0 and all(a.shape[-1] == Dp for a, _ in (rec.get('dropped') or [(x, 0)]))) else -1)
    dx = None
    dropped, keep = rec.get('dropped'), rec.get('keep', 1.0)
    split_in = rec.get('split', False)
    split = max(1, min(32, BT // 2048))
    # The fused product takes an input width that is a multiple of 8.  A feature count that is not (39-dim MFCCs: cfg1, cfg5)
    # runs it at the PADDED width -- the zero pad columns of x give zero gradient rows -- into a scratch kernel gradient whose
    # rows are then added where they belong.  (The separate products this replaces went through the 64-row kernel unsplit once
    # the K slices stopped meeting in atomics: 2 x 1.26 ms exposed behind the last recurrence of a cfg5 step.)
    Df = D if D % 8 == 0 else (Dp if (Dp % 8 == 0 and all(a.shape[-1] == Dp for a, _ in (dropped or [(x, 0)]))) else -1)
    if Df >= 0:
        # fused product: about 700 workgroups in flight measured best on MI355X (128 x 128 output tiles, K cut in slices)
        tiles = -(-(Df + H + 1) // 128) * -(-(4 * H) // 128)
        split = max(1, min(32, BT // 512, round((TN_WORKGROUPS_EXPOSED if exposed else TN_WORKGROUPS) / tiles)))
        wide = TN_WIDE and (4 * H) % 512 == 0 and (exposed or H >= 512)
        if wide:                                 # 128 x 512 output tiles (round 5): half the workgroups per K slice, so twice the slices
            split = max(1, min(32, BT // 512, 2 * split))
    keepalive = [dz, x, y] + [a for a, _ in (dropped or [])]

    def weight_grads():
        for i, (kn, bn) in enumerate(w.names):
            with (overlap or _NoOverlap()).fork(*keepalive, lane=(i % 2 if exposed else 0), beside_chain=(i == 0 and not exposed)):
                gk, gb = grads[kn], grads[bn]
                dzi = dz.view(BT, nd * 4 * H)[:, i * 4 * H:]
                xa, lda = (dropped[i] if dropped is not None else (x, Dp))
                yi = y.view(BT, nd * H)[:, i * H:]
                if Df >= 0:
                    # dK_x, dK_h and db of this direction in one product (dz read once); the K slices meet in a workspace
                    # owned by this layer (its products run one after the other on one stream)
                    # (one workspace per stream the products may run on)
                    need = hip.lib().las_gemm_tn_lstm_workspace_bytes(Df, H, max(split, 2))
                    wss = w.__dict__.setdefault('_tn_ws', {})
                    ws = wss.get(i % 2 if exposed else 0)
                    if ws is None or ws.numel() * 4 < need:
                        ws = wss[i % 2 if exposed else 0] = torch.empty(max(1, need // 4), dtype=torch.float32, device=dev)
                    gdst = gk
                    if Df != D:                 # padded input width: the kernel gradient at [Df + H, 4H], folded back below
                        pads = w.__dict__.setdefault('_gk_pad', {})
                        gdst = pads.get(i)
                        if gdst is None:
                            gdst = pads[i] = torch.empty(Df + H, 4 * H, dtype=torch.float32, device=dev)
                        gdst.zero_()
                    tok = hip.prof_begin('gemm_tn_lstm', 2.0 * (Df + H + 1) * 4 * H * BT)
                    hip.check(hip.lib().las_gemm_tn_lstm(hip.p(xa) if Df > 0 else None, lda, Df, hip.p(yi), nd * H, H,
                                                         (-1 if i == 0 else 1), T, hip.p(dzi), nd * 4 * H, hip.p(gdst), hip.p(gb),
                                                         BT, split | (TN_SPLIT_WIDE if wide else 0), hip.p(ws), hip.stream()))
                    hip.prof_end(tok)
                    if Df != D:
                        gk[:D].add_(gdst[:D])
                        gk[D:].add_(gdst[Df:])
                    continue
                if D > 0:
                    hip.gemm_tn(xa, dzi, gk, D, 4 * H, BT, lda=lda, ldb=nd * 4 * H, ldc=4 * H, split_k=split, c_perm_h=H)
                hip.gemm_tn(yi, dzi, gk[D:], H, 4 * H, BT, lda=nd * H, ldb=nd * 4 * H, ldc=4 * H,
                            a_shift=(-1 if i == 0 else 1), period=T, split_k=split, c_perm_h=H)
                hip.colsum_bf16(dzi, BT, 4 * H, gb, ldx=nd * 4 * H, perm_h=H)

    tok = hip.prof_begin('lstm_bwd', 2.0 * B * T * nd * H * 4 * H)       # dh_{t-1} = dz_t K_h^T of every step
    hip.check(hip.lib().las_lstm_recurrent_bwd(hip.p(rec['gates']), hip.p(rec['cbuf']), hip.p(dy), hip.p(dc_last),
                                               hip.p(dh_last), hip.p(w.kh), hip.p(rec['length']), hip.p(dz),
                                               hip.p(lstm_workspace(B, H, nd)), B, T, H, nd, hip.stream()))
    hip.prof_end(tok)
    # critical path first: dX feeds the next (lower) layer's recurrence
    if need_dx:
        if dropped is None:
            dx = torch.empty(B, T, D, dtype=torch.float32, device=dev)
            if w.kx_img is not None and hip.gemm_nt_bimg_wanted(BT, D, nd * 4 * H):
                hip.gemm_nt_bimg(dz, w.kx_img, dx, BT, D, nd * 4 * H, lda=nd * 4 * H, ldc=D)
            else:
                hip.gemm_nt(dz, w.kx, dx, BT, D, nd * 4 * H, lda=nd * 4 * H, ldb=nd * 4 * H, ldc=D)
        else:
            # per direction dX_i = dZ_i K_x,i^T, then through that direction's input dropout (masks regenerated from
            # the counter-based generator); summed when both directions read the same input, side by side otherwise
            parts = []
            lib = hip.lib()
            if keep < 1.0 and not split_in and nd == 2 and BT > 64 and D > 64 and MASKED_DX:
                # both directions read the same input through their own masks: dX = sum_d mask_d * (dZ_d K_x,d^T) / keep, the
                # mask applied in the products' epilogues (the first stores, the second accumulates)
                seed, stream0 = rec['rng']
                dx = torch.empty(B, T, D, dtype=torch.float32, device=dev)
                for i in range(nd):
                    tok2 = hip.prof_begin('gemm_nt', 2.0 * BT * D * 4 * H)
                    hip.check(lib.las_gemm_nt_masked(hip.p(dz[..., i * 4 * H:]), nd * 4 * H, hip.p(w.kx[:, i * 4 * H:]), nd * 4 * H,
                                                     hip.p(dx), D, BT, D, 4 * H, int(i > 0), keep, seed, stream0 + i, hip.stream()))
                    hip.prof_end(tok2)
            for i in range(nd if dx is None else 0):
                pi = torch.empty(B, T, D, dtype=torch.float32, device=dev)
                hip.gemm_nt(dz[..., i * 4 * H:], w.kx[:, i * 4 * H:], pi, BT, D, 4 * H, lda=nd * 4 * H, ldb=nd * 4 * H,
                            ldc=D)
                parts.append(pi)
            if keep < 1.0 and dx is None:
                seed, stream0 = rec['rng']
                if split_in or nd == 1:
                    for i in range(nd):
                        hip.check(lib.las_dropout_bwd(hip.p(parts[i]), None, hip.p(parts[i]), BT, D, keep, seed,
                                                      stream0 + i, 0, hip.stream()))
                else:
                    dx = torch.empty(B, T, D, dtype=torch.float32, device=dev)
                    hip.check(lib.las_dropout_bwd(hip.p(parts[0]), hip.p(parts[1]), hip.p(dx), BT, D, keep, seed, stream0,
                                                  stream0 + 1, hip.stream()))
            if dx is None:
                dx = torch.cat(parts, -1) if split_in else (parts[0] if nd == 1 else parts[0] + parts[1])
    if defer_weight_grads:
        return dx, weight_grads
    weight_grads()
    return dx


def stacked_bilstm(inputs, sequence_length, mode, hparams, *, weights, tape=None, in_features=None, seed=0,
                   after_first_layer=None):
    """The non-pyramidal listener of las/model.py:111-142: one MultiRNNCell stack per direction (layer l of a
    direction reads that direction's layer l-1 outputs only), outputs = concat(fw_top, bw_top), no time reduction.
    Returns ((outputs, lengths), state) with state = (fw_layer_states, bw_layer_states) (or one tuple when
    unidirectional)."""
    outputs = inputs
    per_layer = []
    for layer in range(hparams.num_layers):
        hooks = after_first_layer if (layer == 0 and after_first_layer is not None) else (None, None)
        out, st = bilstm(outputs, sequence_length, hparams.num_units, hparams.dropout, mode, hparams.unidirectional,
                         weights=weights[layer], tape=tape, in_features=in_features if layer == 0 else None,
                         rng=(seed, 16 + 2 * layer), split_inputs=(layer > 0), after_projection=hooks[0])
        outputs = concat_outputs(out)
        per_layer.append(st)
        if hooks[1] is not None:
            hooks[1]()
    if hparams.unidirectional:
        return (outputs, sequence_length), tuple(per_layer)
    return (outputs, sequence_length), (tuple(s[0] for s in per_layer), tuple(s[1] for s in per_layer))


def pyramidal_stack(outputs, sequence_length, new_len=None):
    """las/ops.py:49-65 on the concatenated buffer: [B,T,C] -> [B,T/2,2C] (view), len -> len//2 + len%2.
    new_len: the stacked lengths when the caller already has them (pyramidal_bilstm forms all levels in one launch)."""
    B, T, C = outputs.shape
    if T % 2:
        raise ValueError('time dimension must be padded to an even length before pyramidal_stack')
    if new_len is None:
        new_len = torch.empty_like(sequence_length)
        hip.check(hip.lib().las_pyramid_lengths(hip.p(sequence_length), hip.p(new_len), B, hip.stream()))
    return outputs.view(B, T // 2, 2 * C), new_len


def pyramidal_bilstm(inputs, sequence_length, mode, hparams, *, variables=None, weights=None, tape=None,
                     in_features=None, seed=0, after_first_layer=None):
    """las/ops.py:68-87.  ``weights``: optional list of LayerWeights per layer (cached images)."""
    outputs = inputs
    state = None
    D = in_features
    levels = None
    if hparams.num_layers > 1:          # the stacked lengths of every level, before the first layer starts
        levels = torch.empty(hparams.num_layers - 1, sequence_length.shape[0], dtype=torch.int32, device=sequence_length.device)
        hip.check(hip.lib().las_pyramid_lengths_multi(hip.p(sequence_length), hip.p(levels), sequence_length.shape[0],
                                                      hparams.num_layers - 1, hip.stream()))
    for layer in range(hparams.num_layers):
        w = weights[layer] if weights is not None else None
        hooks = after_first_layer if (layer == 0 and after_first_layer is not None) else (None, None)
        out, state = bilstm(outputs, sequence_length, hparams.num_units, hparams.dropout, mode,
                            hparams.unidirectional, variables=variables,
                            scope='listener/bilstm_{}'.format(layer), weights=w, tape=tape, in_features=D,
                            rng=(seed, 16 + 2 * layer), after_projection=hooks[0])
        outputs = concat_outputs(out)
        if hooks[1] is not None:
            hooks[1]()
        if layer != 0:
            outputs, sequence_length = pyramidal_stack(outputs, sequence_length, new_len=levels[layer - 1])
            if tape is not None:
                tape.append(dict(kind='stack'))
        D = None
    return (outputs, sequence_length), state
