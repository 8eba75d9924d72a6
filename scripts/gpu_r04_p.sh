# round 4, call P: streamed input products with one operand per direction (dropout / split inputs): tests, stochastic bench A/B
set -x
(timeout 1500 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_stochastic.py -m gpu -q --timeout 600 -k "streamed or dropout or stochastic or replayed" 2>&1 | tail -n 12)
for v in "" "LAS_LSTM_STREAM=0"; do
 for cfg in metric-M-stochastic two-cell-bottom-only; do
  env $v timeout 600 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_p_bench.json
  python - "$cfg $v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_p_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d.get("error"))
    print("   ", [(k["family"], k["launches_per_step"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_p_bench.json").read()[-800:])
PY
 done
done
