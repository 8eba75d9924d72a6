#!/usr/bin/env python3
"""Headline benchmark: utterances/s of a full LAS train step (forward + backward + L2 + per-tensor clip +
Adam [+ RCCL gradient all-reduce]) on synthetic (B=64 per GPU, T=800, F=40) batches — BASELINE.json's metric,
SURVEY.md §8(d) "metric-M": 3-layer pBiLSTM-256 + Luong + 1x256 decoder, V=64, U=80, bf16 operands.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU; weak scaling (each rank trains its own B=64 shard; gradients are summed with one
all-reduce between the local per-tensor clip and Adam, the CrossShardOptimizer order of model_helper.py:405-417).
`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run`; this parent never touches a GPU) and relays rank 0's JSON line as its last
line of stdout.  Rank 0 prints ONE JSON line.  The step is captured once into HIP graphs (torch.cuda.CUDAGraph)
and replayed, or launched eagerly, whichever an untimed probe finds faster.
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
    # the other BASELINE.json configs at their stated shapes (parity-test cases; timed here for DESIGN.md, not bench lines)
    'cfg1': dict(F=39, L=2, H=128, Hd=128, V=64, att='luong', T=300, U=40, B=4),
    'cfg4': dict(F=80, L=4, H=512, Hd=512, V=64, att='bahdanau', T=800, U=80, B=64, ctc=0.3),
    'cfg5': dict(F=39, L=3, H=256, Hd=256, V=197, att='bahdanau_monotonic', T=800, U=80, B=64, binf='binf_map.csv'),
    # the same model at the reference's default stochastic settings (train.py:48,71)
    'metric-M-stochastic': dict(F=40, L=3, H=256, Hd=256, V=64, att='luong', T=800, U=80, B=64, dropout=0.2, sampling=0.1),
    # SURVEY.md 8(d) "ragged variant" of the headline workload: len_i = 800 - 8 (i mod 26), U_i = 80 - (i mod 17)
    'metric-M-ragged': dict(F=40, L=3, H=256, Hd=256, V=64, att='luong', T=800, U=80, B=64, ragged=True),
    # the reference's DEFAULT architecture flags (train.py:33-48,71: no --use_pyramidal, 3 x 128 stacked BiLSTM -- 800 memory
    # frames --, 2 x 128 decoder with the attention around the whole cell stack, dropout 0.2, sampling 0.1) at the headline batch
    'default-arch': dict(F=40, L=3, H=128, Hd=128, V=64, att='luong', T=800, U=80, B=64, dropout=0.2, sampling=0.1,
                         pyramidal=False, dec_layers=2, bottom_only=False, pass_hidden=False),
    # ... and the same decoder depth in the --bottom_only wiring (AttentionMultiCell, las/model.py:36-69) on the pyramidal listener
    'two-cell-bottom-only': dict(F=40, L=3, H=256, Hd=256, V=64, att='luong', T=800, U=80, B=64, dec_layers=2),
}


def lstm_gemm_flops_per_utt(c):
    """forward LSTM-GEMM FLOPs per utterance (input + recurrent), x3 for training."""
    F, L, H, T = c['F'], c['L'], c['H'], c['T']
    tot_in = tot_rec = 0.0
    D, Tl = F, T
    for l in range(L):
        tot_in += 2 * Tl * 2 * D * 4 * H
        tot_rec += 2 * Tl * 2 * H * 4 * H
        if not c.get('pyramidal', True):       # one MultiRNNCell stack per direction: layer l reads H columns, no time reduction
            D = H
            continue
        D = 2 * H * (1 if l == 0 else 2)
        if l >= 1:
            Tl //= 2
    return tot_in, tot_rec


def build_params(c, lr=1e-3, l2=1e-6):
    from phones_las_amd.utils import params_utils as pu
    hp = pu.get_default_hparams()
    for k, v in dict(num_channels=c['F'], encoder_layers=c['L'], encoder_units=c['H'], use_pyramidal=c.get('pyramidal', True),
                     unidirectional=False, decoder_layers=c.get('dec_layers', 1), decoder_units=c['Hd'], target_vocab_size=c['V'],
                     attention_type=c['att'], bottom_only=c.get('bottom_only', True), pass_hidden_state=c.get('pass_hidden', True),
                     dropout=c.get('dropout', 0.0),
                     sampling_probability=c.get('sampling', 0.0), learning_rate=lr, l2_reg_scale=l2,
                     ctc_weight=c.get('ctc', -1.0)).items():
        hp.set_hparam(k, v)
    if c.get('binf'):           # cfg5: --binary_outputs --output_ipa --binf_projection with the reference's misc/binf_map.csv
        for k, v in dict(binary_outputs=True, binf_projection=True, binf_count=binf_matrix(c['binf']).shape[0]).items():
            hp.set_hparam(k, v)
    return pu.get_encoder_decoder_hparams(hp)


def binf_matrix(name):
    """[nf, V] map as the reference's load_binf2phone returned it for misc/<name> (data fixture tests/golden/binf_maps.json)."""
    import numpy as np
    m = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'binf_maps.json')))['maps'][name]
    return np.array([[int(ch) for ch in row] for row in m['rows']], dtype=np.float32)


def synthetic_batch(c, seed, device):
    """x ~ N(0,1) fp32 [B,T,F]; dense variant len = T, U tokens incl. EOS; ragged variant (c['ragged'])
    len_i = T - 8 (i mod 26) with the frames beyond zeroed, U_i = U - (i mod 17) incl. EOS, padded with EOS (id 2) as
    utils/dataset_utils.py:254-267 pads its batches."""
    import numpy as np
    rng = np.random.default_rng(seed)
    B, T, F, V, U = c['B'], c['T'], c['F'], c['V'], c['U']
    x = rng.standard_normal((B, T, F)).astype(np.float32)
    y = rng.integers(3, V, size=(B, U - 1))
    tin = np.concatenate([np.full((B, 1), 1), y], 1).astype(np.int32)
    tout = np.concatenate([y, np.full((B, 1), 2)], 1).astype(np.int32)
    lens = np.full((B,), T, dtype=np.int32)
    ulens = np.full((B,), U, dtype=np.int32)
    if c.get('ragged'):
        for i in range(B):
            lens[i] = T - 8 * (i % 26)
            ulens[i] = U - (i % 17)
            x[i, lens[i]:] = 0.0
            tout[i, ulens[i] - 1:] = 2
            tin[i, ulens[i]:] = 2
    feats = {'encoder_inputs': torch.from_numpy(x).to(device),
             'source_sequence_length': torch.from_numpy(lens).to(device)}
    labels = {'targets_inputs': torch.from_numpy(tin).to(device), 'targets_outputs': torch.from_numpy(tout).to(device),
              'target_sequence_length': torch.from_numpy(ulens).to(device)}
    return feats, labels


KERNEL_NAMES = {        # family (phones_las_amd.hip.KernelTimer) -> kernel symbol(s) in a rocprofv3 trace
    'lstm_fwd': 'lstm_fwd_kernel<%(H)d, %(rows)d, G, KX> (KX > 0: the bottom layer, input projection inside the chain)', 'lstm_bwd': 'lstm_bwd_kernel<%(H)d, %(rows)d, G>',
    'dec_persist_fwd': 'dec_persist_fwd_lean_kernel<att, tiles> (dec_persist_fwd_kernel: scheduled sampling, 512 units, two cells)',
    'dec_persist_bwd': 'dec_persist_bwd_kernel<wq, M/128>',
    'gemm_nt': 'gemm_nt_ring_kernel<256, 128|256, 32, ...> (+ gemm_kernel<..., false, ...> for the small shapes): x K_x, dX, keys, logits',
    'gemm_tn': 'gemm_tn_tr_kernel / gemm_kernel<..., true, ...> (speller weight gradients)',
    'gemm_tn_lstm': 'gemm_tn_ring_kernel + tn_reduce_kernel (dK_x, dK_h, db of a direction)',
}


def kernel_table(step_fn, c, steps=3):
    """Per-kernel-family time of `steps` eagerly launched train steps, HIP events around every launch on the stream it is
    launched on (weight-gradient products: the second stream).  Returns a list of dicts sorted by ms per step."""
    from phones_las_amd import hip
    step_fn()
    torch.cuda.synchronize()
    with hip.KernelTimer() as kt:
        for _ in range(steps):
            step_fn()
    rows = hip.lib().las_lstm_slice_rows(c['B'], c['H'], 2)
    out = []
    for fam, (n, ms, flops) in kt.table().items():
        ms_step = ms / steps
        name = KERNEL_NAMES.get(fam, fam)
        out.append({'name': name % dict(H=c['H'], rows=rows) if '%(' in name else name, 'family': fam, 'launches_per_step': n // steps, 'ms_per_step': round(ms_step, 4),
                    'algorithmic_flops': flops / steps,
                    'tflops': round(flops / steps / (ms_step * 1e-3) / 1e12, 2) if ms_step > 0 else None,
                    'frac': round(flops / steps / (ms_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 5) if ms_step > 0 else None})
    out.sort(key=lambda r: -r['ms_per_step'])
    return out


def csrc_digest():
    """sha256 over the kernel sources (phones-las_amd/csrc/*.hip, *.h, sorted by name): ties a PMC pass to a build.  (.git does
    not travel to the GPU box, so a commit id cannot be read there.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'phones-las_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(family):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC passes (scripts/gpu_pmc.sh ->
    profiles/r0N_pmc_traffic.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be
    read from inside this process: null when no committed pass names this kernel family.  Returns (bytes, source file,
    stale): stale = the kernel sources have changed since that pass was taken (its `csrc_digest` differs from this build's,
    or the file predates the digest)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0*_pmc_traffic.json')))
    for path in reversed(paths):
        d = json.load(open(path))
        k = d.get('kernels', {}).get(family)
        if k:
            return k['traffic_bytes_per_launch'], os.path.relpath(path, ROOT), d.get('csrc_digest') != csrc_digest()
    return None, None, None


def name_of(c):
    return next(k for k, v in CONFIGS.items() if v is c or v == c)


def host_cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(c, sample_b=16, fused_b=64, threads=None):
    """The reference's CPU path timed on this host next to the GPU number.  TensorFlow 1.15 cannot run here, so two
    stated stand-ins, both one full fp32 train step (fwd + autograd bwd + clip + Adam) on a
    bounded sample of the same workload: (i) the oracle's step-wise restatement (one matmul + element-wise ops per time
    step, the structure of dynamic_rnn / dynamic_decode) and (ii) the same step with the listener on torch.nn.LSTM's
    fused kernel (oracle/fused_cpu.py).  Both are timed after one untimed warm-up step on a small batch, (ii) at 16 threads
    AND at os.cpu_count() threads (per-time-step ops are small: more threads mostly add synchronisation, so neither is
    assumed to win); the fastest is quoted as `value` with the threads it used as `cores`."""
    from oracle import las_oracle as O, fused_cpu
    ncpu = os.cpu_count() or 1
    O.set_dtype(torch.float32)
    try:
        hp = O.HP(encoder=O.EncoderHP(num_layers=c['L'], num_units=c['H']), num_channels=c['F'],
                  decoder=O.DecoderHP(num_layers=1, num_units=c['Hd'], target_vocab_size=c['V'],
                                      attention_type=c['att'], bottom_only=True, pass_hidden_state=True))
        params = {k: v.float() for k, v in O.init_params(hp).items()}
        zeros = {k: torch.zeros_like(v) for k, v in params.items()}

        def run(fn, b, T=None):
            batch = O.synthetic_batch(b, T or c['T'], c['F'], c['V'], c['U'])
            batch['encoder_inputs'] = batch['encoder_inputs'].float()
            t0 = time.time()
            out = fn(batch)
            O.adam_apply(params, zeros, zeros, out['clipped'], 1, 1e-3)
            return time.time() - t0

        step_wise = lambda b: O.train_step(hp, params, None, None, 1, b)
        fused = lambda b: fused_cpu.train_step_fused(hp, params, b)
        runs = []                     # (label, threads, utterances, seconds)
        base_th = threads or min(ncpu, 16)
        torch.set_num_threads(base_th)
        run(fused, min(8, fused_b), T=min(c['T'], 64))            # primitive creation / thread pool, untimed
        runs.append(('fused_lstm', base_th, fused_b, run(fused, fused_b)))
        th = threads or min(ncpu, 16)
        torch.set_num_threads(th)
        run(step_wise, 2, T=min(c['T'], 64))                      # warm-up, untimed
        runs.append(('step_wise', th, sample_b, run(step_wise, sample_b)))
    finally:
        O.set_dtype(torch.float64)
    rate = lambda r: r[2] / r[3]
    all_core = None
    # The same fused stand-in at MORE host threads (32, 64, 128 and all of them), each leg in a child process with a hard
    # wall-clock cap: per-time-step ops on 100+ threads can be far slower than on 16 (a first version ran the all-core leg
    # in-process on a 192-thread host and did not come back in 14 minutes), and the baseline must stay bounded.  A leg that
    # does not finish is reported as such; the fastest leg that did is the quoted value (the honest denominator).
    import subprocess
    import threading
    nb = fused_b                 # the utterance count of the base run: thread count is the only thing that changes
    cap = 30                     # seconds of TIMED work a leg may take (interpreter start, import, init and warm-up do not count)
    sweep = []
    lost = None                  # the thread count at which the sweep was already 3x behind the best run
    for th in [t for t in (32, 64, 128) if t < ncpu and t != base_th] + ([ncpu] if ncpu != base_th else []):
        if lost is not None:     # more threads only add synchronisation to per-time-step ops: every round measured it; not re-learned
            leg = {'threads': th, 'note': 'skipped: the %d-thread leg was already more than 3x slower than the best run' % lost}
            sweep.append(leg)
            if th == ncpu:
                all_core = leg
            continue
        code = ('import sys, time, torch; sys.path.insert(0, %r); import bench; from oracle import las_oracle as O, fused_cpu\n'
                'c = bench.CONFIGS[%r]; torch.set_num_threads(%d); O.set_dtype(torch.float32)\n'
                'hp = O.HP(encoder=O.EncoderHP(num_layers=c["L"], num_units=c["H"]), num_channels=c["F"], decoder=O.DecoderHP(num_layers=1, '
                'num_units=c["Hd"], target_vocab_size=c["V"], attention_type=c["att"], bottom_only=True, pass_hidden_state=True))\n'
                'p = {k: v.float() for k, v in O.init_params(hp).items()}\n'
                'def run(b, T):\n'
                '    bt = O.synthetic_batch(b, T, c["F"], c["V"], c["U"]); bt["encoder_inputs"] = bt["encoder_inputs"].float()\n'
                '    t0 = time.time(); fused_cpu.train_step_fused(hp, p, bt); return time.time() - t0\n'
                'run(2, 64); print("READY", flush=True); print("SECONDS", run(%d, c["T"]), flush=True)\n' % (ROOT, name_of(c), th, nb))
        sec, measured_bound = None, None
        try:
            child = subprocess.Popen([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            lines = []
            reader = threading.Thread(target=lambda: [lines.append(l) for l in child.stdout], daemon=True)
            reader.start()
            t_start, t_ready = time.time(), None
            while True:
                if t_ready is None and any(l.startswith('READY') for l in lines):
                    t_ready = time.time()                    # the cap counts from here
                if any(l.startswith('SECONDS') for l in lines) or child.poll() is not None:
                    break
                if (t_ready is not None and time.time() - t_ready > cap) or (t_ready is None and time.time() - t_start > 180):
                    child.kill()
                    break
                time.sleep(0.05)
            child.wait()
            reader.join(timeout=5)
            got = [l for l in lines if l.startswith('SECONDS')]
            if got:
                sec = float(got[0].split()[1])
                leg = {'threads': th, 'utterances': nb, 'seconds': round(sec, 2), 'utt_s': round(nb / sec, 4)}
                runs.append(('fused_lstm', th, nb, sec))
            elif t_ready is not None:
                measured_bound = nb / cap                 # the timed section itself ran out of its cap: a measured upper bound
                leg = {'threads': th, 'utterances': nb, 'seconds': None,
                       'note': 'the timed step did not finish within %d s (< %.2f utterances/s): slower than the %d-thread run' % (cap, measured_bound, base_th)}
            else:
                leg = {'threads': th, 'note': 'the child did not get through import / init / warm-up (not a measurement: the sweep goes on)'}
        except Exception as e:      # noqa: BLE001
            leg = {'threads': th, 'note': 'failed: %s' % e}
        sweep.append(leg)
        if th == ncpu:
            all_core = leg
        # `lost` only from a MEASUREMENT: a finished step, or a timed section that exhausted its cap -- never from a start-up timeout
        rate_here = nb / sec if sec else measured_bound
        if rate_here is not None and rate_here * 3.0 < max(rate(r) for r in runs):
            lost = th
    best = max(runs, key=rate)
    return {'value': round(rate(best), 4), 'unit': 'utterances/s', 'cores': best[1], 'kind': 'port',
            'host_cpu_count': ncpu, 'host_cpu_model': host_cpu_model(), 'all_core_run': all_core, 'thread_sweep': sweep,
            'runs': [{'stand_in': r[0], 'threads': r[1], 'utterances': r[2], 'seconds': round(r[3], 2),
                      'utt_s': round(rate(r), 4)} for r in runs],
            'sample': 'one full fp32 train step (fwd+bwd+clip+Adam) of the same model on T=%d utterances, torch-CPU stand-ins '
                      'for TF 1.15, each after an untimed warm-up: step-wise oracle on %d utterances, fused torch.nn.LSTM listener '
                      'on %d utterances at %d threads and on the same utterance count at each of the `thread_sweep` counts (child processes; a leg may take '
                      '30 s of TIMED work -- start-up and warm-up do not count --, the sweep stops once a MEASURED leg is 3x behind; `all_core_run` = the leg on every host thread); value = the fastest run that finished' % (c['T'], sample_b, fused_b, base_th)}


def choose_step_form(candidates, probe, read_and_clear_status, any_rank, rank=0):
    """The untimed probe of bench.py: times every candidate form of the step (name -> (step function, uses graphs, overlapped
    exchange)) with `probe` (which returns the SLOWEST rank's time on every rank), reads AND CLEARS the persistent kernels'
    status words after each one on every rank, drops a form that timed out on ANY rank (`any_rank`), and picks the fastest of the
    rest -- forms within 1 % of the fastest tie, and a tie goes to eager launches (eight probe steps of a 6-ms step scatter by that
    much, and a graph picked on a 0.5 % edge has replayed slower than eager launches as often as not); candidates are walked in
    sorted order so that every rank takes the same decisions.  With every form dropped the plain eager form is the fall-back.
    Returns (chosen name, {name: seconds}, {name: why it was dropped})."""
    probed, dropped = {}, {}
    if len(candidates) > 1:
        for name in sorted(candidates):
            t_ = probe(candidates[name][0])
            bad = read_and_clear_status()
            if any_rank(bool(bad)):
                dropped[name] = 'persistent-kernel timeout in the probe (status %s on rank %d)' % (bad, rank)
            else:
                probed[name] = t_
        if probed:
            best = min(probed.values())
            chosen = min(sorted(n for n in probed if probed[n] <= 1.01 * best), key=lambda n: (n.endswith('_graph'), probed[n]))
        else:
            chosen = 'plain_eager' if 'plain_eager' in candidates else sorted(candidates)[0]
    else:
        chosen = next(iter(candidates))
    return chosen, probed, dropped


# Step forms bench.py may time; each is pinned to LasModel.train_step by tests/test_gpu_step_forms.py (same parameters, Adam
# slots and loss, bit for bit, after K steps from the same weights).  main() refuses to time anything else.
COVERED_FORMS = ('plain_eager', 'plain_graph', 'overlap_eager', 'overlap_graph')


class StepForms:
    """The forms of one train step bench.py chooses between (model_helper.py:403-417: ONE train op per step in all of them).

    plain_*:   part_a (zero grads, forward, loss, backward, L2 term + per-tensor norms; single replica: clip + Adam of every
               tensor above the bottom listener layer beside that layer's weight-gradient products) -> all-reduce (several
               ranks) -> part_b (the rest of clip + Adam).
    overlap_*: the gradient exchange in two buckets, the first one beside the lower layers' backward: part_a1 -> all-reduce
               of bucket 0 (asynchronous) -> part_a2 -> all-reduce of bucket 1 -> part_b_dp.
    *_eager launches every kernel from the host, *_graph replays HIP graphs captured once (torch.cuda.CUDAGraph) with the
    all-reduces between them.  `multi`: the data-parallel form (clip, exchange, Adam as three passes)."""

    def __init__(self, model, feats, labels, num_steps, multi):
        self.model, self.feats, self.labels, self.U, self.multi = model, feats, labels, num_steps, multi
        dev = feats['encoder_inputs'].device
        self.loss_buf = torch.zeros(1, device=dev)
        self.audio_buf = torch.zeros(1, device=dev)
        self.graphs = []

    # -- the plain step ------------------------------------------------------------------------------------------------------
    def part_a(self):     # zero grads, forward, loss, backward, L2 term + per-tensor norms (+ clip before the all-reduce)
        model = self.model
        model.vars.grad.zero_()
        audio, _, dlogits = model.forward_train(self.feats, self.labels, num_steps=self.U)
        if self.multi:
            model.backward(dlogits)
            model.collect_status(zero_norms=True)   # timeout flag of the persistent kernels: travels with the gradients, gates Adam
            model.clip_gradients()
        else:
            # single replica: norms + clip + Adam of everything above the bottom listener layer run beside that layer's
            # weight-gradient products (LasModel.apply_gradients); the bottom layer's own update is part_b
            model.backward(dlogits, join=False)
            model.apply_gradients(joined=False, update_tail=False)
        model.total_loss(audio, out=self.loss_buf)

    def part_b(self):     # (clip +) Adam + refresh of the bf16 weight images
        model = self.model
        if self.multi:
            model.adam_update()
        else:
            model.apply_tail()
        model.global_step += 1        # eager steps draw fresh dropout / sampling streams; a captured graph keeps its seed
        # (the bf16 weight images are rebuilt at the start of the next step's forward, beside the bottom layer's recurrence)

    def reduce(self):
        if self.multi:
            self.model.all_reduce_gradients()

    def plain_eager(self):
        self.part_a(); self.reduce(); self.part_b()

    # -- the exchange in two buckets, the first one (top listener layer + speller) handed to RCCL while the lower layers'
    #    backward is still running: three parts with the asynchronous all-reduces between them ------------------------------
    def part_a1(self):
        model = self.model
        model.vars.grad.zero_()
        audio, _, dlogits = model.forward_train(self.feats, self.labels, num_steps=self.U)
        self.audio_buf.copy_(audio)
        model.backward_exchange_begin(dlogits, exchange=False)

    def part_a2(self):
        self.model.backward_exchange_end([], exchange=False)
        self.model.total_loss(self.audio_buf, out=self.loss_buf)

    def part_b_dp(self):
        self.model.adam_update()
        self.model.global_step += 1

    def _overlapped(self, run_a1, run_a2, run_b):
        model = self.model
        b0, b1 = model.vars.buckets

        def step():
            run_a1()
            w0 = model.all_reduce_gradients(b0, async_op=True)
            run_a2()
            w1 = model.all_reduce_gradients(b1, async_op=True)
            w0.wait(); w1.wait()
            run_b()
        return step

    def overlap_eager(self):
        self._overlapped(self.part_a1, self.part_a2, self.part_b_dp)()

    # -- construction --------------------------------------------------------------------------------------------------------
    _warm_stream = {}       # device -> the side stream every StepForms of this process warms up on (streams share a handful of
                            # hardware queues: a process should not create one per model)

    def _eagerly(self, fn, n=2):
        """n eager runs on a side stream (allocations, job tables, workspaces: what graph capture needs to find in place)."""
        dev = torch.cuda.current_device()
        s = StepForms._warm_stream.get(dev)
        if s is None:
            s = StepForms._warm_stream[dev] = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(n):        # (the second step starts from stale weight images, as every later one: its job tables)
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

    def warm_up(self):
        self._eagerly(self.plain_eager)

    def build(self, want_eager=True, want_graph=True, overlap_exchange=False, plain_too=True):
        """name -> (step function, uses graphs, overlapped exchange).  warm_up() first.  Every eager run that prepares a
        capture is a REAL optimiser step (the parity test counts them)."""
        candidates = {}
        if overlap_exchange:
            self._eagerly(self.overlap_eager)          # the three parts eagerly (allocations, job tables) before capture
            if want_eager:
                candidates['overlap_eager'] = (self.overlap_eager, False, True)
            if want_graph:
                g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1):
                    self.part_a1()
                with torch.cuda.graph(g2, pool=g1.pool()):
                    self.part_a2()
                with torch.cuda.graph(g3, pool=g1.pool()):
                    self.part_b_dp()
                self.graphs += [g1, g2, g3]
                candidates['overlap_graph'] = (self._overlapped(g1.replay, g2.replay, g3.replay), True, True)
        if plain_too:
            # the plain step: one all-reduce of the flat gradient buffer between the backward pass and Adam
            if want_eager:
                candidates['plain_eager'] = (self.plain_eager, False, False)
            if want_graph:
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga):
                    self.part_a()
                with torch.cuda.graph(gb):
                    self.part_b()
                self.graphs += [ga, gb]

                def step_graph():
                    ga.replay(); self.reduce(); gb.replay()
                candidates['plain_graph'] = (step_graph, True, False)
        return candidates


def launcher_command(argv, gpus, port):
    """The command `bench.py --gpus N` runs when it has to start its own ranks (no WORLD_SIZE in the environment)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus),
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch(argv, gpus):
    """Start the ranks as a child torch.distributed.run and relay its stdout; rank 0's JSON line is printed last.  This
    process never initialises a GPU (the children own them); a failing child makes the exit code non-zero."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: what RCCL needs on this host driver
    proc = subprocess.Popen(launcher_command(argv, gpus, free_port()), stdout=subprocess.PIPE, env=env, text=True)
    last_json = None
    for line in proc.stdout:
        t = line.strip()
        if t.startswith('{') and '"metric"' in t:
            last_json = t                       # held back: printed as the very last line
        else:
            sys.stdout.write(line)
    rc = proc.wait()
    sys.stdout.flush()
    if last_json is not None:
        print(last_json, flush=True)
    return rc if rc else (0 if last_json is not None else 1)


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
    ap.add_argument('--dp-overlap', nargs='?', const='on', default='auto', choices=['auto', 'on', 'off'],
                    help="gradient exchange in two buckets, the first one beside the lower layers' backward.  auto (default): "
                         "with more than one rank an untimed probe picks the faster of the overlapped and the plain exchange; "
                         "off with one rank.  on with --gpus 1: the all-reduces run on a 1-rank RCCL group (plumbing check)")
    ap.add_argument('--dp-test-group', action='store_true',
                    help='--gpus 1 only: run the multi-rank code path (clip, all-reduce, Adam; the overlap probe) on a 1-rank RCCL group')
    ap.add_argument('--cpu-sample', type=int, default=16)
    args = ap.parse_args()
    c = CONFIGS[args.config]

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # started bare (`python bench.py --gpus N`): become the launcher.  Nothing in this process has touched a GPU.
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if rank != 0:
        os.dup2(2, 1)          # only rank 0 owns stdout (one JSON line, last); library banners of the others go to stderr
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d inside a job of WORLD_SIZE %d' % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    group = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    elif args.dp_overlap == 'on' or args.dp_test_group:     # plumbing check on one GPU: a 1-rank RCCL group
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(free_port()))
        torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        group = torch.distributed.group.WORLD
    multi = world > 1 or args.dp_test_group       # the data-parallel form of the step: clip, exchange, Adam
    rccl_ranks = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1

    from phones_las_amd import model_helper as mh
    model = mh.LasModel(build_params(c), world_size=world, process_group=group, rank=rank,
                        binf2phone=binf_matrix(c['binf']) if c.get('binf') else None)
    want_overlap = args.dp_overlap == 'on' or (args.dp_overlap == 'auto' and multi)
    overlap_exchange = want_overlap and len(model.enable_exchange_overlap()) == 2
    feats, labels = synthetic_batch(c, 1234 + rank, dev)
    feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])   # resident bf16 [B,T,F'] batch
    U = c['U']
    forms = StepForms(model, feats, labels, U, multi)
    loss_buf = forms.loss_buf
    forms.warm_up()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def probe(fn, n=8):
        """Untimed (not part of `value`) wall time of n steps; every rank gets the slowest rank's time, so all agree."""
        fn(); barrier()
        t_ = time.perf_counter()
        for _ in range(n):
            fn()
        barrier()
        t_ = time.perf_counter() - t_
        if world > 1:
            tt = torch.tensor([t_], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            t_ = float(tt[0])
        return t_

    candidates = forms.build(want_eager=args.no_graph or args.launch == 'auto', want_graph=not args.no_graph,
                             overlap_exchange=overlap_exchange, plain_too=(not overlap_exchange or args.dp_overlap == 'auto'))
    uncovered = sorted(set(candidates) - set(COVERED_FORMS))
    if uncovered:        # (a form is only timed when tests/test_gpu_step_forms.py pins it to LasModel.train_step)
        raise SystemExit('bench.py: step form(s) %s have no parity test (COVERED_FORMS)' % uncovered)

    # The step has about 130 launches.  Replaying them as HIP graphs takes the host out of the picture; launching them
    # eagerly lets the host run ahead of the GPU, which is a little faster when the host is quick and idle (graph nodes
    # carry a fixed cost).  With several ranks the exchange beside the backward pass may or may not pay (RCCL's kernels
    # share the chip with the persistent recurrent kernels).  An untimed probe of every candidate form picks the fastest
    # for the timed steps; the ranks agree on it (max over ranks).
    def any_rank(flag):
        """True on every rank when `flag` is true on any of them."""
        if world > 1:
            tt = torch.tensor([1.0 if flag else 0.0], device=dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            return bool(tt.item() > 0)
        return bool(flag)

    # A probe candidate whose persistent kernels hit a bounded-wait timeout (conceivable for the overlapped forms, where
    # RCCL's kernels share the chip with groups that need co-residency) must not poison the run: the status words are
    # sticky and gate Adam, so they are read AND CLEARED after every candidate, on every rank, and a form that timed out on
    # any rank is dropped.  The plain eager form is the fall-back.
    chosen, probed, dropped = choose_step_form(candidates, probe, model.read_and_clear_status, any_rank, rank)
    step, used_graph, used_overlap = candidates[chosen]

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

    eager = candidates.get('overlap_eager' if used_overlap else 'plain_eager')
    if eager is None:
        eager = (forms.overlap_eager if used_overlap else forms.plain_eager, False, used_overlap)
    config = {'workload': '%s: %d-layer %s-%d + %s attention + %dx%d LSTM decoder, V=%d, U=%d, %s '
                          'T=%d, F=%d, full train step' % (args.config, c['L'], 'pBiLSTM' if c.get('pyramidal', True) else 'stacked BiLSTM',
                                                           c['H'], c['att'], c.get('dec_layers', 1), c['Hd'],
                                                           c['V'], c['U'], 'ragged (len 600..800, U 64..80)' if c.get('ragged') else 'dense',
                                                           c['T'], c['F']),
              'global_batch': c['B'] * world, 'parallelism': 'dp%d' % world, 'rccl_ranks': rccl_ranks,
              'hip_graph': used_graph, 'exchange': 'two buckets, overlapped' if used_overlap else 'one all-reduce',
              'step_form': chosen, 'probe_s': {k: round(v, 4) for k, v in probed.items()},
              'probe_dropped': dropped,
              'stochastic_draws': 'fresh every step' if not (c.get('dropout') or c.get('sampling')) or not used_graph
                                  else 'frozen at capture (graph replay)'}
    out, rc = epilogue(rank, world, dt, final_loss, model.read_and_clear_status(), eager[0], c, args, config, any_rank, barrier)
    if torch.distributed.is_initialized():
        torch.cuda.synchronize()
        torch.distributed.destroy_process_group()
    if rank == 0:
        # the JSON line must be the LAST line on stdout: RCCL writes its NCCL_DEBUG=VERSION banner to the C stdout
        # buffer, which would otherwise be flushed after this print at process exit
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if rc:
        print('bench.py: ' + out['error'] if out else 'bench.py: invalid run', file=sys.stderr)
        raise SystemExit(rc)


def epilogue(rank, world, dt, final_loss, bad_status, eager_step, c, args, config, any_rank, barrier, table_fn=None):
    """What every rank does after the timed region; returns (JSON record on rank 0 else None, exit code).
    Anything that invalidates the number -- a bounded-wait timeout of a persistent kernel during the timed steps (Adam was
    withheld on them), a non-finite loss -- is reported IN the record (field `error`) and through the exit code; the record
    is produced either way.  The per-kernel timing pass (the same step launched eagerly with HIP events around every launch,
    after the timed region so that `value` is not perturbed) contains the gradient all-reduce when there are several ranks,
    so EVERY rank runs it (rank 0 alone waited for peers that had already left); only rank 0 builds the record."""
    import math
    errors = []
    if any_rank(bool(bad_status)):
        errors.append('a persistent kernel reported a bounded-wait timeout during the timed steps (status %s on rank %d): '
                      'Adam was withheld on those steps' % (bad_status, rank))
    if any_rank(not math.isfinite(final_loss)):
        errors.append('final_loss is not finite (%r on rank %d)' % (final_loss, rank))
    kernels = (table_fn or kernel_table)(eager_step, c)
    barrier()
    rc = 1 if errors else 0
    if rank != 0:
        return None, rc
    ms = dt / args.steps * 1e3
    utt_s = c['B'] * world * args.steps / dt
    f_in, f_rec = lstm_gemm_flops_per_utt(c)
    dom = kernels[0]         # the dominant kernel = the family with the largest time per step
    # the committed PMC passes are of the default configuration's step: another configuration's kernels (other unit
    # counts, other shapes) have no counter evidence and say so
    traffic, traffic_src, traffic_stale = pmc_traffic(dom['family']) if args.config == 'metric-M' else (None, None, None)
    step_tflops = 3 * (f_in + f_rec) * utt_s / world / 1e12
    config = dict(config, final_loss=round(final_loss, 4) if math.isfinite(final_loss) else repr(final_loss))
    out = {
        'metric': 'utterances/s LAS train step (B=64 per GPU, T=800, F=40)', 'value': round(utt_s, 2),
        'unit': 'utterances/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16', 'data': 'synthetic', 'config': config,
        'roofline': {'bound': 'mfma', 'kernel': dom['name'],
                     'achieved': dom['tflops'], 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': dom['frac'], 'traffic': traffic, 'traffic_source': traffic_src,
                     'traffic_stale': traffic_stale, 'csrc_digest': csrc_digest(),
                     'kernel_ms_per_step': dom['ms_per_step'], 'launches_per_step': dom['launches_per_step'],
                     'algorithmic_flops_per_step': dom['algorithmic_flops'],
                     'kernels': kernels,
                     'whole_step_lstm_gemm_tflops_per_gpu': round(step_tflops, 3),
                     'whole_step_frac': round(step_tflops / PEAK_BF16_TFLOPS, 6)},
    }
    if errors:
        out['error'] = '; '.join(errors)
    if world == 1 and not args.no_cpu_baseline and c.get('dec_layers', 1) == 1 and c.get('pyramidal', True):
        out['cpu_baseline'] = cpu_baseline(c, args.cpu_sample)       # (the stand-ins are built for the headline architecture family)
    return out, rc


if __name__ == '__main__':
    main()
