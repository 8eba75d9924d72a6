# round 4, call I: two decoder cells in one forward launch (tests, bench A/B)
set -x
(timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q --timeout 600 -k "two_cell or general_decoder" 2>&1 | tail -n 25)
for v in "" "LAS_DEC_PERSIST2=0"; do
  env $v timeout 900 python bench.py --config two-cell-bottom-only --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_i_bench.json
  python - "two-cell-bottom-only $v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_i_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d.get("error"))
    print("   ", [(k["family"], k["launches_per_step"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_i_bench.json").read()[-1500:])
PY
done
