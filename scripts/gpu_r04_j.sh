# round 4, call J: two decoder cells in one forward AND one backward launch, input dropout + scheduled sampling inside (tests, bench A/B)
set -x
(timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_stochastic.py -m gpu -x -q --timeout 600 -k "two_cell or general_decoder or one_launch or persistent_decoder or padding_rows" 2>&1 | tail -n 25)
for cfg in two-cell-bottom-only default-arch; do
for v in "" "LAS_DEC_PERSIST2_BWD=0" "LAS_DEC_PERSIST2=0"; do
  env $v timeout 900 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_j_bench.json
  python - "$cfg $v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_j_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d.get("error"))
    print("   ", [(k["family"], k["launches_per_step"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_j_bench.json").read()[-1500:])
PY
done
done
