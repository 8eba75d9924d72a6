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
