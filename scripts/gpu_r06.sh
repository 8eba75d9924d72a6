#!/bin/bash
# Round-6 GPU recipes, one per `gpurun` call:  bash scripts/gpu_r06.sh <recipe> [args]   (output: gpurun_out/r06_<recipe>.log)
# ab_base/ = a built git worktree of the round-5 tree (git worktree add ab_base <commit>; python phones-las_amd/build.py).
set -u
R=${1:-help}; shift || true
mkdir -p gpurun_out
LOG=gpurun_out/r06_$R.log
bench_line() {   # dir, env..., then bench args after --
  local d=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  (cd $d && env "${envs[@]}" python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 |
     python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', '${envs[*]}', d['config']['workload'], d['value'], d['ms_per_step'], d['config'].get('step_form'), d.get('final_loss'))")
}
case $R in
  lstm_ab)      # recurrent kernels alone: round-5 tree, this tree with and without the helper waves; then parity
    { for rep in 1 2; do
        (cd ab_base && python scripts/gpu_lstm_time.py 2>&1 | sed 's/^/r05      /')
        LAS_LSTM_BWD_HW=0 python scripts/gpu_lstm_time.py 2>&1 | sed 's/^/r06 hw=0 /'
        LAS_LSTM_BWD_HW=1 python scripts/gpu_lstm_time.py 2>&1 | sed 's/^/r06 hw=1 /'
      done
      timeout 900 python -m pytest tests/test_gpu_lstm.py -q -x --timeout 300 2>&1 | grep -v '^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl' | tail -6
    } > $LOG 2>&1 ;;
  bench_ab)     # whole step, interleaved: round-5 tree / this tree (args: bench.py arguments, e.g. --config metric-L)
    { for rep in 1 2 3; do
        bench_line ab_base X=0 -- "$@"
        bench_line . LAS_LSTM_BWD_HW=0 -- "$@"
        bench_line . LAS_LSTM_BWD_HW=1 -- "$@"
      done } > $LOG 2>&1 ;;
  fwd_early)    # forward recurrence with the member's own K chunks multiplied in the shadow of the exchange (EARLY) against the build without
    NOE=LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_noearly.so
    { for H in 256 512; do for rep in 1 2; do
        (cd ab_base && H=$H python scripts/gpu_lstm_time.py 2>&1 | grep '^H' | sed 's/^/r05        /')
        env $NOE H=$H python scripts/gpu_lstm_time.py 2>&1 | grep '^H' | sed 's/^/r06 noearly /'
        H=$H python scripts/gpu_lstm_time.py 2>&1 | grep '^H' | sed 's/^/r06 early   /'
      done; done
      timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_golden_shapes.py -q -x --timeout 300 2>&1 | grep -v '^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl' | tail -6
      for cfg in metric-M metric-L; do for rep in 1 2; do
        bench_line ab_base X=0 -- --config $cfg
        bench_line . $NOE -- --config $cfg
        bench_line . X=0 -- --config $cfg
      done; done
    } > $LOG 2>&1 ;;
  adj)          # backward: adjacent-sender polls + toggled offsets + per-flavour loops, against the round-5 tree (256 and 512 units)
    { for H in 256 512; do for rep in 1 2; do
        (cd ab_base && H=$H python scripts/gpu_lstm_time.py 2>&1 | grep '^H' | sed 's/^/r05 /')
        H=$H python scripts/gpu_lstm_time.py 2>&1 | grep '^H' | sed 's/^/r06 /'
      done; done
      timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_golden_shapes.py tests/test_gpu_dp_equivalence.py -q -x --timeout 300 2>&1 | grep -v '^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl' | tail -6
      for cfg in metric-M metric-L cfg4; do for rep in 1 2; do
        bench_line ab_base X=0 -- --config $cfg
        bench_line . X=0 -- --config $cfg
      done; done
    } > $LOG 2>&1 ;;
  hw128)        # helper waves in the single-workgroup backward chains (64 / 128 units): alone, parity, the configs that use them
    { for H in 128 64; do for rep in 1 2; do
        (cd ab_base && H=$H python scripts/gpu_lstm_time.py 2>&1 | grep '^H' | sed 's/^/r05      /')
        LAS_LSTM_BWD_HW=0 H=$H python scripts/gpu_lstm_time.py 2>&1 | grep '^H' | sed 's/^/r06 hw=0 /'
        H=$H python scripts/gpu_lstm_time.py 2>&1 | grep '^H' | sed 's/^/r06 hw=1 /'
      done; done
      timeout 1200 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_golden_shapes.py tests/test_gpu_dp_equivalence.py tests/test_gpu_trajectory.py -q -x --timeout 300 2>&1 | grep -v '^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl' | tail -6
      for cfg in default-arch cfg1 metric-M; do for rep in 1 2; do
        bench_line ab_base X=0 -- --config $cfg
        bench_line . X=0 -- --config $cfg
      done; done
    } > $LOG 2>&1 ;;
  tests)        # the GPU suite
    timeout 2400 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | grep -v '^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl' > $LOG ;;
  *) echo "recipes: lstm_ab bench_ab fwd_early adj hw128 tests"; exit 2 ;;
esac
tail -40 $LOG
