# round 4, call K: two-cell one-launch kernels after the register work (tests, benches)
set -x
(timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_stochastic.py -m gpu -q --timeout 600 -k "two_cell or general_decoder or one_launch or persistent_decoder" 2>&1 | tail -n 15)
for cfg in two-cell-bottom-only default-arch; do
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_k_bench.json
  python - "$cfg" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_k_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d.get("error"))
    print("   ", [(k["family"], k["launches_per_step"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_k_bench.json").read()[-1500:])
PY
done
