# round 4, call Q: size of the streamed producer's grid (A/B on metric-M)
set -x
for v in "LAS_STREAM_GRID=512" "LAS_STREAM_GRID=96" "LAS_STREAM_GRID=128" "LAS_STREAM_GRID=192" "LAS_STREAM_GRID=256" "LAS_STREAM_GRID=512 LAS_STREAM_HOG_KB=0"; do
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_q_bench.json
  python - "metric-M $v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_q_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"]["step_form"], d["config"]["probe_s"], d.get("error"), [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"][:4]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_q_bench.json").read()[-800:])
PY
done
