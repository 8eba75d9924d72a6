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
