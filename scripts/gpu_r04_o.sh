# round 4, call O: backward recurrence in time windows (tests, A/B on metric-M / metric-L)
set -x
(timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_lstm.py -m gpu -q --timeout 600 -k "window" 2>&1 | tail -n 25)
for v in "LAS_BWD_WINDOWS=1" "LAS_BWD_WINDOWS=2" "LAS_BWD_WINDOWS=3" "LAS_BWD_WINDOWS=4"; do
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_o_bench.json
  python - "metric-M $v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_o_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d["config"]["probe_s"], d.get("error"))
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_o_bench.json").read()[-800:])
PY
done
for v in "LAS_BWD_WINDOWS=1" "LAS_BWD_WINDOWS=3" "LAS_BWD_WINDOWS=3 LAS_BWD_WINDOWS_ALL=1"; do
  env $v timeout 600 python bench.py --config metric-L --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_o_bench.json
  python - "metric-L $v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_o_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d.get("error"))
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_o_bench.json").read()[-800:])
PY
done
