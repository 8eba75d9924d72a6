#!/usr/bin/env python3
"""Headline benchmark: utterances/s of a full LAS train step (forward + backward + L2 + per-tensor clip +
Adam [+ RCCL gradient all-reduce]) on synthetic (B=64 per GPU, T=800, F=40) batches — BASELINE.json's metric,
SURVEY.md §8(d) "metric-M": 3-layer pBiLSTM-256 + Luong + 1x256 decoder, V=64, U=80, bf16 operands.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU; weak scaling (each rank trains its own B=64 shard; gradients are summed with one
all-reduce between the local per-tensor clip and Adam, the CrossShardOptimizer order of model_helper.py:405-417).
Rank 0 prints ONE JSON line.  The step is captured once into HIP graphs (torch.cuda.CUDAGraph) and replayed:
the same kernels, no host launch overhead.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"

CONFIGS = {
    'metric-M': dict(F=40, L=3, H=256, Hd=256, V=64, att='luong', T=800, U=80, B=64),
    'metric-L': dict(F=40, L=4, H=512, Hd=512, V=64, att='bahdanau', T=800, U=80, B=64),
    'tiny': dict(F=40, L=2, H=64, Hd=64, V=64, att='luong', T=64, U=8, B=16),
    # the same model at the reference's default stochastic settings (train.py:48,71)
    'metric-M-stochastic': dict(F=40, L=3, H=256, Hd=256, V=64, att='luong', T=800, U=80, B=64, dropout=0.2, sampling=0.1),
}


def lstm_gemm_flops_per_utt(c):
    """forward LSTM-GEMM FLOPs per utterance (input + recurrent), x3 for training."""
    F, L, H, T = c['F'], c['L'], c['H'], c['T']
    tot_in = tot_rec = 0.0
    D, Tl = F, T
    for l in range(L):
        tot_in += 2 * Tl * 2 * D * 4 * H
        tot_rec += 2 * Tl * 2 * H * 4 * H
        D = 2 * H * (1 if l == 0 else 2)
        if l >= 1:
            Tl //= 2
    return tot_in, tot_rec


def build_params(c, lr=1e-3, l2=1e-6):
    from phones_las_amd.utils import params_utils as pu
    hp = pu.get_default_hparams()
    for k, v in dict(num_channels=c['F'], encoder_layers=c['L'], encoder_units=c['H'], use_pyramidal=True,
                     unidirectional=False, decoder_layers=1, decoder_units=c['Hd'], target_vocab_size=c['V'],
                     attention_type=c['att'], bottom_only=True, pass_hidden_state=True, dropout=c.get('dropout', 0.0),
                     sampling_probability=c.get('sampling', 0.0), learning_rate=lr, l2_reg_scale=l2).items():
        hp.set_hparam(k, v)
    return pu.get_encoder_decoder_hparams(hp)


def synthetic_batch(c, seed, device):
    """SURVEY.md §8(d) dense variant: x ~ N(0,1) fp32 [B,T,F], len = T, U tokens incl. EOS."""
    import numpy as np
    rng = np.random.default_rng(seed)
    B, T, F, V, U = c['B'], c['T'], c['F'], c['V'], c['U']
    x = rng.standard_normal((B, T, F)).astype(np.float32)
    y = rng.integers(3, V, size=(B, U - 1))
    tin = np.concatenate([np.full((B, 1), 1), y], 1).astype(np.int32)
    tout = np.concatenate([y, np.full((B, 1), 2)], 1).astype(np.int32)
    feats = {'encoder_inputs': torch.from_numpy(x).to(device),
             'source_sequence_length': torch.full((B,), T, dtype=torch.int32, device=device)}
    labels = {'targets_inputs': torch.from_numpy(tin).to(device), 'targets_outputs': torch.from_numpy(tout).to(device),
              'target_sequence_length': torch.full((B,), U, dtype=torch.int32, device=device)}
    return feats, labels


def time_dominant_kernel(c, reps=5):
    """HIP-event timing of the dominant kernel (the layer-1 forward recurrence: the longest serial chain)
    on its own, on torch's current stream (the stream the library launches on)."""
    from phones_las_amd import hip
    from phones_las_amd.las import ops
    B, T, H = c['B'], c['T'], c['H']
    dev = 'cuda'
    xproj0 = torch.randn(B, T, 8 * H, device=dev) * 0.5
    khp = (torch.randn(2, H * 4 * H, device=dev) * 0.05).to(torch.bfloat16)
    length = torch.full((B,), T, dtype=torch.int32, device=dev)
    y = torch.empty(B, T, 2 * H, dtype=torch.bfloat16, device=dev)
    cbuf = torch.empty(B, T, 2 * H, device=dev)
    cl = torch.empty(2, B, H, device=dev)
    hl = torch.empty(2, B, H, device=dev)
    ws = ops.lstm_workspace(B, H, 2)
    lib = hip.lib()
    times = []
    for i in range(reps + 1):
        xproj = xproj0.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(lib.las_lstm_recurrent_fwd(hip.p(xproj), hip.p(khp), hip.p(length), hip.p(y), hip.p(cbuf), hip.p(cl),
                                             hip.p(hl), hip.p(ws), B, T, H, 2, hip.stream()))
        e1.record()
        e1.synchronize()
        if i:
            times.append(e0.elapsed_time(e1))
    ops.check_lstm_status(B, H, 2)
    ms = sum(times) / len(times)
    flops = B * T * 2 * 2 * H * 4 * H          # recurrent GEMM h_{t-1} K_h of both directions
    return ms, flops


def cpu_baseline(c, sample_b=32, threads=None):
    """The oracle (a CPU port of the reference's per-time-step graph; TF 1.15 itself cannot run here) timed in
    fp32 on the host cores for ONE train step over `sample_b` utterances of the same shape."""
    from oracle import las_oracle as O
    # the per-time-step ops are small: more threads than ~16 only add synchronisation cost
    threads = threads or min(os.cpu_count(), 16)
    torch.set_num_threads(threads)
    O.set_dtype(torch.float32)
    try:
        hp = O.HP(encoder=O.EncoderHP(num_layers=c['L'], num_units=c['H']), num_channels=c['F'],
                  decoder=O.DecoderHP(num_layers=1, num_units=c['Hd'], target_vocab_size=c['V'],
                                      attention_type=c['att'], bottom_only=True, pass_hidden_state=True))
        params = {k: v.float() for k, v in O.init_params(hp).items()}
        batch = O.synthetic_batch(sample_b, c['T'], c['F'], c['V'], c['U'])
        batch['encoder_inputs'] = batch['encoder_inputs'].float()
        t0 = time.time()
        out = O.train_step(hp, params, None, None, 1, batch)
        zeros = {k: torch.zeros_like(v) for k, v in params.items()}
        O.adam_apply(params, zeros, zeros, out['clipped'], 1, 1e-3)
        dt = time.time() - t0
    finally:
        O.set_dtype(torch.float64)
    return {'value': round(sample_b / dt, 4), 'unit': 'utterances/s', 'cores': threads, 'kind': 'port',
            'sample': '1 full train step (fwd+bwd+clip+Adam) on %d utterances of T=%d, fp32 torch-CPU oracle, %.1f s'
                      % (sample_b, c['T'], dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='metric-M', choices=sorted(CONFIGS))
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying HIP graphs')
    ap.add_argument('--launch', default='auto', choices=['auto', 'graph'],
                    help="auto: an untimed probe picks graph replay or eager launches, whichever is faster; graph: always replay")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dp-overlap', action='store_true',
                    help='exchange the gradients in two buckets, the first one beside the lower layers\' backward '
                         '(with --gpus 1 the all-reduces run on a 1-rank RCCL group: plumbing check)')
    ap.add_argument('--cpu-sample', type=int, default=32)
    args = ap.parse_args()
    c = CONFIGS[args.config]

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if rank != 0:
        os.dup2(2, 1)          # only rank 0 owns stdout (one JSON line, last); library banners of the others go to stderr
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                             % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from phones_las_amd import model_helper as mh
    group = None
    if args.dp_overlap and world == 1:          # plumbing check on one GPU: the exchange runs on a 1-rank RCCL group
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        group = torch.distributed.group.WORLD
    model = mh.LasModel(build_params(c), world_size=world, process_group=group)
    overlap_exchange = bool(args.dp_overlap) and len(model.enable_exchange_overlap()) == 2
    feats, labels = synthetic_batch(c, 1234 + rank, dev)
    feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])   # resident bf16 [B,T,F'] batch
    U = c['U']
    loss_buf = torch.zeros(1, device=dev)

    def part_a():     # zero grads, forward, loss, backward, L2 term + per-tensor norms (+ clip before the all-reduce)
        model.vars.grad.zero_()
        audio, _, dlogits = model.forward_train(feats, labels, num_steps=U)
        model.backward(dlogits)
        if world > 1:
            model.clip_gradients()
        else:
            model.gradient_norms()
        loss_buf.copy_(audio + model.l2_loss(from_norms=True))

    def part_b():     # (clip +) Adam + refresh of the bf16 weight images
        if world > 1:
            model.adam_update()
        else:
            model.clip_adam_update()
        model.refresh_images()

    def reduce():
        if world > 1 or group is not None:
            torch.distributed.all_reduce(model.vars.grad)

    # --dp-overlap: the exchange in two buckets, the first one (top listener layer + speller) handed to RCCL while the
    # lower layers' backward is still running: three graphs with the asynchronous all-reduces between them
    def part_a1():
        model.vars.grad.zero_()
        audio, _, dlogits = model.forward_train(feats, labels, num_steps=U)
        audio_buf.copy_(audio)
        model.backward_exchange_begin(dlogits, exchange=False)

    def part_a2():
        model.backward_exchange_end([], exchange=False)
        loss_buf.copy_(audio_buf + model.l2_loss(from_norms=True))

    def part_b_dp():
        model.adam_update()
        model.refresh_images()

    # eager warm-up on a side stream (also what graph capture needs)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        part_a(); reduce(); part_b()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()

    audio_buf = torch.zeros(1, device=dev)
    if overlap_exchange:
        b0, b1 = model.vars.buckets
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):              # eager run of the three parts (allocations, job tables) before capture
            part_a1(); part_a2(); part_b_dp()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if args.no_graph:
            run_a1, run_a2, run_b = part_a1, part_a2, part_b_dp
        else:
            g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                part_a1()
            with torch.cuda.graph(g2, pool=g1.pool()):
                part_a2()
            with torch.cuda.graph(g3, pool=g1.pool()):
                part_b_dp()
            run_a1, run_a2, run_b = g1.replay, g2.replay, g3.replay

        def step():
            run_a1()
            w0 = model.all_reduce_gradients(b0, async_op=True)
            run_a2()
            w1 = model.all_reduce_gradients(b1, async_op=True)
            w0.wait(); w1.wait()
            run_b()
    elif args.no_graph:
        def step():
            part_a(); reduce(); part_b()
    else:
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            part_a()
        with torch.cuda.graph(gb):
            part_b()

        def step_graph():
            ga.replay(); reduce(); gb.replay()

        def step_eager():
            part_a(); reduce(); part_b()

        step = step_graph

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    used_graph = not args.no_graph
    if not args.no_graph and not overlap_exchange and args.launch == 'auto':
        # The step has about 130 launches.  Replaying them as HIP graphs takes the host out of the picture; launching them
        # eagerly lets the host run ahead of the GPU, which is a little faster when the host is quick and idle (graph
        # nodes carry a fixed cost).  Untimed probe of both, the faster one runs the timed steps; ranks agree on it.
        def probe(fn, n=8):
            fn(); barrier()
            t_ = time.perf_counter()
            for _ in range(n):
                fn()
            barrier()
            return time.perf_counter() - t_
        tg, te = probe(step_graph), probe(step_eager)
        if world > 1:
            tt = torch.tensor([tg, te], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            tg, te = float(tt[0]), float(tt[1])
        if te < 0.995 * tg:
            step, used_graph = step_eager, False

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss_buf.item())

    model.check_device_status()        # a bounded inter-workgroup wait that timed out would invalidate the numbers
    if rank == 0:
        ms = dt / args.steps * 1e3
        utt_s = c['B'] * world * args.steps / dt
        f_in, f_rec = lstm_gemm_flops_per_utt(c)
        kms, kflops = time_dominant_kernel(c)
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
        if args.config == 'metric-M' and os.path.exists(pmc):
            traffic = json.load(open(pmc))['traffic_bytes']        # PMC passes of the same kernel (scripts/gpu_pmc.sh)
        achieved = kflops / (kms * 1e-3) / 1e12
        step_tflops = 3 * (f_in + f_rec) * utt_s / world / 1e12
        out = {
            'metric': 'utterances/s LAS train step (B=64 per GPU, T=800, F=40)', 'value': round(utt_s, 2),
            'unit': 'utterances/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': '%s: %d-layer pBiLSTM-%d + %s attention + 1x%d LSTM decoder, V=%d, U=%d, dense '
                                   'T=%d, F=%d, full train step' % (args.config, c['L'], c['H'], c['att'], c['Hd'],
                                                                    c['V'], c['U'], c['T'], c['F']),
                       'global_batch': c['B'] * world, 'parallelism': 'dp%d' % world,
                       'hip_graph': used_graph, 'exchange': 'two buckets, overlapped' if overlap_exchange else 'one all-reduce',
                       'final_loss': round(final_loss, 4)},
            'roofline': {'bound': 'mfma', 'kernel': 'lstm_fwd_kernel<%d, %d> (layer-1 shape, both directions)' % (
                             c['H'], __import__('phones_las_amd.hip', fromlist=['lib']).lib().las_lstm_slice_rows(c['B'], c['H'], 2)),
                         'achieved': round(achieved, 3), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / PEAK_BF16_TFLOPS, 6), 'traffic': traffic,
                         'kernel_ms': round(kms, 3),
                         'whole_step_lstm_gemm_tflops_per_gpu': round(step_tflops, 3),
                         'whole_step_frac': round(step_tflops / PEAK_BF16_TFLOPS, 6)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(c, args.cpu_sample)
    if world > 1 or group is not None:
        torch.cuda.synchronize()
        torch.distributed.destroy_process_group()
    if rank == 0:
        # the JSON line must be the LAST line on stdout: RCCL writes its NCCL_DEBUG=VERSION banner to the C stdout
        # buffer, which would otherwise be flushed after this print at process exit
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
