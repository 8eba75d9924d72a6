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
  prof)         # kernel stats + dispatch timeline of one config (args: tag, then bench.py arguments)
    T=${1:-r06_p}; shift || true
    bash scripts/gpu_prof.sh $T --steps 3 --warmup 1 --no-graph "$@" > gpurun_out/$T.out 2>&1
    python scripts/trace_timeline.py gpurun_out/$T > gpurun_out/${T}_timeline.txt 2>&1
    { head -30 gpurun_out/$T.out; tail -60 gpurun_out/${T}_timeline.txt; } > $LOG ;;
  final)        # the round's measurement set (args: tag [suite|nosuite]) -> gpurun_out/, copied to profiles/r06_*
    T=${1:-r06_z}
    { if [ "${2:-suite}" = suite ]; then
        (time timeout 2400 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|error" | tail -5) 2>&1 | tail -8
      fi
      bash scripts/gpu_prof.sh ${T}_metricM --steps 3 --warmup 1 --no-graph > gpurun_out/${T}_metricM.out 2>&1
      python scripts/trace_timeline.py gpurun_out/${T}_metricM > gpurun_out/${T}_timeline_metricM.txt 2>&1
      bash scripts/gpu_prof.sh ${T}_metricL --config metric-L --steps 3 --warmup 1 --no-graph > gpurun_out/${T}_metricL.out 2>&1
      bash scripts/gpu_prof.sh ${T}_cfg5 --config cfg5 --steps 3 --warmup 1 --no-graph > gpurun_out/${T}_cfg5.out 2>&1
      bash scripts/gpu_prof.sh ${T}_default-arch --config default-arch --steps 3 --warmup 1 --no-graph > gpurun_out/${T}_default-arch.out 2>&1
      if [ -f phones-las_amd/liblas_hip_stamps.so ]; then     # diagnostics build: LAS_CXXFLAGS=-DLAS_STAMPS LAS_HIP_LIB=.../liblas_hip_stamps.so python phones-las_amd/build.py
        { for H in 256 512; do LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so H=$H python scripts/gpu_lstm_stamps.py 2>&1 | grep -v amdgpu.ids; done
          echo "--- 256 units, four-wave backward (LAS_LSTM_BWD_HW=0) for comparison"
          LAS_LSTM_BWD_HW=0 LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so H=256 python scripts/gpu_lstm_stamps.py 2>&1 | grep -A8 '^backward'
        } > gpurun_out/r06_lstm_stamps.txt 2>&1
        LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_dec_stamps.py > gpurun_out/r06_dec_stamps_metricM.txt 2>&1
      fi
      bash scripts/gpu_pmc.sh r06 > gpurun_out/r06_pmc.out 2>&1
      timeout 1200 python bench.py --steps 30 --warmup 5 2>&1 | tail -1 > gpurun_out/r06_bench_metricM.json
      for c in metric-M-ragged metric-M-stochastic metric-L cfg4 cfg5 cfg1 default-arch two-cell-bottom-only; do
        timeout 400 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r06_bench_$c.json
      done
      python - <<PY
import json
for c in ["metricM","metric-M-ragged","metric-M-stochastic","metric-L","cfg4","cfg5","cfg1","default-arch","two-cell-bottom-only"]:
    try:
        d=json.load(open("gpurun_out/r06_bench_%s.json"%c))
        print(c, d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"].get("step_form"), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic_stale"), d.get("cpu_baseline",{}).get("value"), d.get("error"))
    except Exception as e:
        print(c, "FAILED", e)
PY
    } > $LOG 2>&1 ;;
  fwd_hw)       # helper waves in the 256-unit forward chains: alone, parity, whole step
    { for rep in 1 2; do
        (cd ab_base && python scripts/gpu_lstm_time.py 2>&1 | grep '^H' | sed 's/^/r05       /')
        LAS_LSTM_FWD_HW=0 python scripts/gpu_lstm_time.py 2>&1 | grep '^H' | sed 's/^/r06 fhw=0 /'
        python scripts/gpu_lstm_time.py 2>&1 | grep '^H' | sed 's/^/r06 fhw=1 /'
      done
      timeout 1200 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_golden_shapes.py tests/test_gpu_step_forms.py -q -x --timeout 300 2>&1 | grep -v '^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl' | tail -6
      for cfg in metric-M metric-M-ragged cfg5; do for rep in 1 2; do
        bench_line . LAS_LSTM_FWD_HW=0 -- --config $cfg
        bench_line . LAS_LSTM_FWD_HW=1 -- --config $cfg
      done; done
    } > $LOG 2>&1 ;;
  dec512)       # the 512-unit Bahdanau decoder alone and metric-L / cfg4 / cfg5 against the round-5 tree (flat loads of the query path removed)
    { for rep in 1 2; do
        (cd ab_base && python scripts/gpu_dec512_time.py 2>&1 | tail -3 | sed 's/^/r05 /')
        python scripts/gpu_dec512_time.py 2>&1 | tail -3 | sed 's/^/r06 /'
      done
      timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_model.py tests/test_gpu_golden_shapes.py -q -x --timeout 300 2>&1 | grep -v '^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl' | tail -6
      for cfg in metric-L cfg5; do for rep in 1 2; do
        bench_line ab_base X=0 -- --config $cfg
        bench_line . X=0 -- --config $cfg
      done; done
    } > $LOG 2>&1 ;;
  tn_ablate)    # what the TN ring's K loop waits for: the product build against -DTNR_NO_MFMA / _NO_DSREAD / _NO_GLDS builds (metric-L shapes, alone)
    { for rep in 1 2; do
        TAG=product python scripts/gpu_tn_ablate.py 2>&1 | grep TN
        for f in NO_MFMA NO_DSREAD NO_GLDS; do
          TAG=$f LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_tnr_$f.so python scripts/gpu_tn_ablate.py 2>&1 | grep TN
        done
      done } > $LOG 2>&1 ;;
  tests)        # the GPU suite
    timeout 2400 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | grep -v '^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl' > $LOG ;;
  *) echo "recipes: lstm_ab bench_ab fwd_early adj hw128 fwd_hw dec512 tn_ablate prof final tests"; exit 2 ;;
esac
tail -40 $LOG
