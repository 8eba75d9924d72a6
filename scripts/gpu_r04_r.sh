# round 4, call R: streamed input products at 512 units once more (recurrence first, smaller producer grid): A/B on metric-L
set -x
for v in "" "LAS_LSTM_STREAM_512=1" "LAS_LSTM_STREAM_512=1 LAS_STREAM_GRID=128" "LAS_LSTM_STREAM_512=1 LAS_STREAM_GRID=96"; do
  env $v timeout 600 python bench.py --config metric-L --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_r_bench.json
  python - "metric-L $v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_r_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"]["step_form"], d.get("error"), [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"][:5]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_r_bench.json").read()[-800:])
PY
done
