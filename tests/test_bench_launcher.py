"""bench.py's own multi-GPU launcher (`python bench.py --gpus N` without a surrounding torch.distributed.run): the
command it builds, the relay of rank 0's JSON line as the LAST stdout line, the exit code of a failing child -- all
without touching a GPU in the launching process."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_launcher_command_is_one_rank_per_gpu_on_localhost():
    import torch
    bench = _bench()
    cmd = bench.launcher_command(['--gpus', '8', '--steps', '20', '--warmup', '5'], 8, 29511)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29511'
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']
    assert not torch.cuda.is_initialized()          # building the command touches no GPU


def test_self_launch_relays_the_json_line_last_and_the_exit_code(tmp_path, monkeypatch):
    bench = _bench()
    fake = tmp_path / 'fake_ranks.py'
    fake.write_text('import sys, json\n'
                    'print("NCCL banner")\n'
                    'print(json.dumps({"metric": "m", "value": 1.5, "n_gpus": 2}))\n'
                    'print("late library output")\n'
                    'sys.exit(int(sys.argv[1]))\n')
    for code in (0, 3):
        monkeypatch.setattr(bench, 'launcher_command', lambda argv, gpus, port, c=code: [sys.executable, str(fake), str(c)])
        script = ('import sys; sys.path.insert(0, %r); import bench; bench.launcher_command = lambda a, g, p: [sys.executable, %r, %r];'
                  'sys.exit(bench.self_launch([], 2))' % (ROOT, str(fake), str(code)))
        r = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, timeout=120)
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert r.returncode == code
        assert json.loads(lines[-1]) == {'metric': 'm', 'value': 1.5, 'n_gpus': 2}       # JSON last, after the late output
        assert 'late library output' in lines[:-1] and 'NCCL banner' in lines[:-1]


def test_bare_multi_gpu_invocation_becomes_the_launcher(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE: the process must hand over to its launcher before any CUDA call.
    Here (no GPU) the child ranks fail, which must surface as a non-zero exit code -- not as the old SystemExit message."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--no-cpu-baseline', '--config', 'tiny'], capture_output=True, text=True, timeout=600, env=env)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
        assert 'must be launched with torch.distributed.run' not in (r.stdout + r.stderr)
        assert 'torch.distributed' in r.stderr or 'ChildFailedError' in r.stderr or 'HIP' in r.stderr or 'cuda' in r.stderr.lower()


def _epilogue_worker(rank, world, port, ret):
    """bench.epilogue on CPU under gloo: the per-kernel timing pass re-runs a step that contains the gradient all-reduce, so
    EVERY rank has to run it (rank 0 alone used to wait there for peers that had already left -- this worker
    would hang and the test time out).  One rank reports a non-finite loss: every rank must leave with exit code 1 and
    rank 0's record must carry the `error` field next to the measured value."""
    import argparse
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    bench = _bench()
    calls = []

    def eager_step():                      # forward/backward stand-in + the exchange every rank must join
        t = torch.ones(4) * (rank + 1)
        dist.all_reduce(t)
        calls.append(float(t[0]))

    def table_fn(step, c):
        for _ in range(4):                 # kernel_table: one warm step + three timed ones
            step()
        return [{'name': 'lstm_bwd_kernel<256, 4>', 'family': 'lstm_bwd', 'launches_per_step': 3, 'ms_per_step': 2.2,
                 'algorithmic_flops': 1.342e11, 'tflops': 61.0, 'frac': 0.0244}]

    def any_rank(flag):
        t = torch.tensor([1.0 if flag else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    args = argparse.Namespace(steps=10, warmup=2, no_cpu_baseline=True, cpu_sample=16, config='metric-M')
    loss = float('nan') if rank == 1 else 1.25
    out, rc = bench.epilogue(rank, world, 0.0626, loss, [], eager_step, bench.CONFIGS['metric-M'], args,
                             {'workload': 'metric-M'}, any_rank, dist.barrier, table_fn=table_fn)
    ret[rank] = (out, rc, list(calls))
    dist.destroy_process_group()


def test_epilogue_runs_the_timing_pass_on_every_rank_and_reports_errors_in_the_json_line():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_epilogue_worker, args=(2, port, ret), nprocs=2, join=True)
    (out0, rc0, calls0), (out1, rc1, calls1) = ret[0], ret[1]
    assert calls0 == calls1 == [3.0] * 4                       # both ranks ran the four steps of the timing pass
    assert out1 is None and rc0 == rc1 == 1                    # rank 1's NaN loss fails the run on every rank
    assert 'not finite' in out0['error'] and out0['value'] == round(64 * 2 * 10 / 0.0626, 2)
    assert out0['n_gpus'] == 2 and out0['roofline']['kernel'].startswith('lstm_bwd') and 'traffic_stale' in out0['roofline']
    json.dumps(out0)                                           # the record serialises (NaN loss is kept as a string)


def _probe_worker(rank, world, port, ret):
    """bench.choose_step_form under gloo with two ranks and the four candidate forms of a multi-rank run (plain / overlapped
    exchange x eager / graph): every form's step contains a collective, the probe's time is the SLOWEST rank's, one form
    reports a persistent-kernel timeout on ONE rank only (it must be dropped on both), and the status words are read -- and
    cleared -- after every candidate on every rank."""
    import time
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    bench = _bench()
    cost = {'plain_eager': 0.004, 'plain_graph': 0.00398, 'overlap_eager': 0.002, 'overlap_graph': 0.0035}   # seconds per step on rank 0
    state = {'current': None, 'status_reads': 0, 'steps': {k: 0 for k in cost}}

    def make(name):
        def step():
            state['current'] = name
            state['steps'][name] += 1
            time.sleep(cost[name] * (1.5 if rank == 1 else 1.0))         # rank 1 is the slower one
            t = torch.ones(2)
            dist.all_reduce(t)                                           # the gradient exchange every form contains
        return step

    def barrier():
        dist.barrier()

    def probe(fn, n=4):
        fn(); barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        barrier()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt[0])

    def read_and_clear_status():
        state['status_reads'] += 1
        # the overlapped eager form times out on rank 1 only (RCCL kernels beside the persistent groups, say)
        return [1] if (state['current'] == 'overlap_eager' and rank == 1) else []

    def any_rank(flag):
        t = torch.tensor([1.0 if flag else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    candidates = {k: (make(k), k.endswith('_graph'), k.startswith('overlap')) for k in cost}
    chosen, probed, dropped = bench.choose_step_form(candidates, probe, read_and_clear_status, any_rank, rank)
    # every form dropped: the plain eager form is the fall-back
    all_bad = bench.choose_step_form(candidates, probe, lambda: [2], any_rank, rank)
    ret[rank] = (chosen, probed, dropped, state['status_reads'], dict(state['steps']), all_bad[0], sorted(all_bad[2]))
    dist.destroy_process_group()


def test_probe_picks_one_form_on_every_rank_and_drops_a_form_that_timed_out_anywhere():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_probe_worker, args=(2, port, ret), nprocs=2, join=True)
    (c0, p0, d0, reads0, steps0, fb0, fbd0), (c1, p1, d1, reads1, steps1, fb1, fbd1) = ret[0], ret[1]
    assert c0 == c1 == 'overlap_graph'                                  # the fastest of the forms that survived, on both ranks
    assert sorted(d0) == sorted(d1) == ['overlap_eager']               # the timeout on rank 1 dropped the form on rank 0 too
    assert sorted(p0) == sorted(p1) == ['overlap_graph', 'plain_eager', 'plain_graph']
    for k in p0:
        assert abs(p0[k] - p1[k]) < 1e-9                                # the slowest rank's time, identical on both
    assert p0['plain_eager'] > 4 * 0.004 * 1.4                          # ... i.e. rank 1's (1.5 x slower)
    assert reads0 == reads1 == 4 and steps0 == steps1                   # status read after every candidate, same steps everywhere
    assert fb0 == fb1 == 'plain_eager' and fbd0 == fbd1 == sorted(['overlap_eager', 'overlap_graph', 'plain_eager', 'plain_graph'])


def test_probe_prefers_eager_launches_within_one_percent():
    bench = _bench()
    times = {'plain_eager': 1.004, 'plain_graph': 1.0}
    cands = {k: ((lambda: None), k.endswith('_graph'), False) for k in times}
    order = []
    chosen, probed, dropped = bench.choose_step_form(cands, lambda fn: times[order.append(1) or sorted(times)[len(order) - 1]],
                                                     lambda: [], lambda f: f)
    assert chosen == 'plain_eager' and not dropped
    times['plain_eager'] = 1.02
    order.clear()
    assert bench.choose_step_form(cands, lambda fn: times[order.append(1) or sorted(times)[len(order) - 1]], lambda: [], lambda f: f)[0] == 'plain_graph'
    assert bench.choose_step_form({'plain_eager': cands['plain_eager']}, None, None, None)[0] == 'plain_eager'      # one candidate: no probe


def _exchange_model(params_bytes, hidden_bytes, step_s, n=8, link_gbs=153.0, links=7):
    """Per-step seconds of the four forms of a data-parallel step under a simple model of a ring all-reduce over xGMI
    (2 (n-1)/n of the bytes over the slowest link pair, MI355X: 7 links x 153 GB/s per GPU, a ring uses one link each way;
    +40 us of latency per collective): `plain` pays the whole exchange behind the backward pass, `overlap` only the bucket that
    is not hidden beside the lower layers' recurrences.  Graph replay saves 0.5 % of the host-side launch distance.  FAKE
    timings: what is under test is the decision, not the model."""
    t = lambda b: 2.0 * (n - 1) / n * b / (link_gbs * 1e9) + 40e-6
    plain = step_s + t(params_bytes)
    overlap = step_s + t(params_bytes - hidden_bytes) + 40e-6
    return {'plain_eager': plain, 'plain_graph': plain * 0.995, 'overlap_eager': overlap, 'overlap_graph': overlap * 0.995}


def test_overlap_decision_for_small_and_large_gradient_buckets():
    """VERDICT r4 #7a: `--dp-overlap auto` with several ranks probes the plain and the two-bucket exchange and picks by time.
    With metric-M's 26 MB of gradients the exchange is a few per cent of the step whichever way, with cfg3's 146 MB (29.7 M
    listener + 6.8 M speller parameters, fp32: BASELINE.json configs[2]) the bucket hidden beside the lower layers decides; a
    form that times out is dropped whatever its time.  Bucket sizes come from the parameter tables (no GPU)."""
    sys.path.insert(0, ROOT)
    bench = _bench()
    from phones_las_amd import model_helper as mh

    def buckets(cfg):
        p = mh.physical_params(bench.build_params(bench.CONFIGS[cfg]))
        table = mh.param_table(p)
        names = [n for n, _, _ in table]
        size = lambda shape: int(__import__('numpy').prod(shape)) * 4
        lead = next(i for i, n in enumerate(names) if n.startswith('listener/'))
        cut = lead + (p.encoder.num_layers - 1) * 4              # LasModel.enable_exchange_overlap: [top layer + speller (+ CTC)] | [lower layers]
        first = sum(size(s) for n, s, _ in table[cut:]) + sum(size(s) for n, s, _ in table[:lead])
        return sum(size(s) for _, s, _ in table), first

    total_m, first_m = buckets('metric-M')
    total_l, first_l = buckets('cfg4')
    assert 24e6 < total_m < 28e6 and 140e6 < total_l < 150e6                         # 26 MB / 146 MB
    assert 0.5 < first_m / total_m < 0.8 and 0.3 < first_l / total_l < 0.7           # the bucket that leaves first is a real share

    def decide(times, bad=()):
        state = {'cur': None}
        cands = {k: ((lambda k=k: state.__setitem__('cur', k)), k.endswith('_graph'), k.startswith('overlap')) for k in times}

        def probe(fn):
            fn()
            return times[state['cur']] * 8
        return bench.choose_step_form(cands, probe, lambda: ([1] if state['cur'] in bad else []), lambda f: f)

    # metric-M at 6 ms: the exchange is 0.34 ms, all but 0.13 ms of it hidden by the overlapped form: > 1 % -> overlapped (and,
    # the graph form being only 0.5 % ahead of the eager one, eager launches: the tie rule)
    chosen, probed, dropped = decide(_exchange_model(total_m, first_m, 6.0e-3))
    assert chosen == 'overlap_eager' and not dropped and len(probed) == 4
    # ... and once the whole exchange is within 1 % of the step (a step of 60 ms, say) every form ties: the first eager one
    chosen, _, _ = decide(_exchange_model(total_m, first_m, 60e-3))
    assert chosen in ('overlap_eager', 'plain_eager')
    # cfg3 / cfg4 at 18 ms: 1.7 ms of exchange, 0.8 ms of it hidden: the overlapped form by a wide margin
    t = _exchange_model(total_l, first_l, 18e-3)
    assert t['plain_graph'] - t['overlap_graph'] > 0.5e-3
    chosen, probed, dropped = decide(t)
    assert chosen == 'overlap_eager'
    # the overlapped forms time out beside the persistent kernels (RCCL's workgroups took CUs a chain needed): dropped, the
    # plain form runs the timed steps
    chosen, probed, dropped = decide(t, bad=('overlap_eager', 'overlap_graph'))
    assert chosen == 'plain_eager' and sorted(dropped) == ['overlap_eager', 'overlap_graph'] and sorted(probed) == ['plain_eager', 'plain_graph']
