# round 4, call E: fused projection + loss launch (tests, A/B), the convergence twin, the whole GPU suite, metric-L trace
set -x
(timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q --timeout 600 2>&1 | tail -n 30)
(timeout 900 python -m pytest tests/test_gpu_convergence.py -m gpu -x -q -s --timeout 800 2>&1 | tail -n 12)
for v in "" "LAS_PROJ_CE=0"; do
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_e_bench.json
  python - "$v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_e_bench.json"))
    print("BENCH", sys.argv[1] or "default", d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d["config"]["probe_s"], d.get("error"))
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_e_bench.json").read()[-800:])
PY
done
for v in "" "LAS_LSTM_FUSED_X=0" "LAS_LSTM_STREAM=0"; do
  env $v timeout 600 python bench.py --config metric-L --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_e_bench.json
  python - "$v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_e_bench.json"))
    print("BENCH metric-L", sys.argv[1] or "default", d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d.get("error"))
    print("   ", [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_e_bench.json").read()[-800:])
PY
done
bash scripts/gpu_prof.sh r04_e_metricL --config metric-L --steps 3 --warmup 1 --no-graph > gpurun_out/r04_e_metricL.out 2>&1
python scripts/trace_timeline.py gpurun_out/r04_e_metricL > gpurun_out/r04_e_timeline_L.txt 2>&1
(time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --ignore tests/test_gpu_convergence.py --deselect tests/test_gpu_model.py 2>&1 | grep -E "passed|failed|Error|error" | tail -n 12) 2>&1 | tail -n 16
