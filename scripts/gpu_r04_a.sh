# round 4, call A: the new recurrent-kernel tests first (launch epochs, fused input projection), then the whole GPU suite,
# then A/B bench lines (new defaults vs the round-3 forms through the env switches) and a kernel trace of the new step
set -x
(timeout 900 python -m pytest tests/test_gpu_lstm.py -m gpu -x -q --timeout 600 2>&1 | tail -15)
(time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_lstm.py 2>&1 | grep -E "passed|failed|Error|error" | tail -12) 2>&1 | tail -16
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_a_bench_new.json
LAS_LSTM_FUSED_X=0 LAS_TAIL_OVERLAP=0 LAS_DMEM_MERGED=0 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_a_bench_old.json
LAS_LSTM_FUSED_X=0 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_a_bench_nofused.json
LAS_TAIL_OVERLAP=0 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_a_bench_notail.json
LAS_DMEM_MERGED=0 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_a_bench_nomerge.json
timeout 600 python bench.py --config metric-M-ragged --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_a_bench_ragged.json
python - <<PY
import json
for c in ["new","old","nofused","notail","nomerge","ragged"]:
    try:
        d=json.load(open("gpurun_out/r04_a_bench_%s.json"%c))
        print(c, d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d["config"]["probe_s"], d.get("error"))
    except Exception as e:
        print(c, "FAILED", e, open("gpurun_out/r04_a_bench_%s.json"%c).read()[-600:])
PY
bash scripts/gpu_prof.sh r04_a_metricM --steps 3 --warmup 1 --no-graph > gpurun_out/r04_a_metricM.out 2>&1
python scripts/trace_timeline.py gpurun_out/r04_a_metricM > gpurun_out/r04_a_timeline.txt 2>&1
tail -5 gpurun_out/r04_a_timeline.txt
