# round 4, call F: input products inside the poll wait (A/B), convergence twin over six seeds, the whole GPU suite with names
set -x
(timeout 900 python -m pytest tests/test_gpu_lstm.py -m gpu -x -q --timeout 600 2>&1 | tail -n 8)
(timeout 1200 python -m pytest tests/test_gpu_convergence.py -m gpu -x -q --timeout 1000 2>&1 | tail -n 8)
for v in "" "LAS_LSTM_FUSED_X=0" "LAS_LSTM_STREAM=0" "LAS_LSTM_STREAM=0 LAS_LSTM_FUSED_X=0"; do
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_f_bench.json
  python - "$v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_f_bench.json"))
    print("BENCH", sys.argv[1] or "default", d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d["config"]["probe_s"], d.get("error"))
    print("   ", [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_f_bench.json").read()[-800:])
PY
done
(time timeout 1800 python -m pytest tests -m gpu -q -rf --timeout 600 --ignore tests/test_gpu_convergence.py --deselect tests/test_gpu_lstm.py 2>&1 | grep -E "passed|failed|FAILED" | tail -n 12) 2>&1 | tail -n 16
