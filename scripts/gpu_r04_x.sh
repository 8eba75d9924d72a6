# round 4, call X: padded key rows + compile-time-width Luong scores in the general forward body (tests, benches)
set -x
(timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_stochastic.py tests/test_gpu_golden_shapes.py -m gpu -q --timeout 600 2>&1 | tail -n 30)
for cfg in default-arch two-cell-bottom-only cfg5; do
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_x_bench.json
  python - "$cfg" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_x_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d.get("error"), [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"] if k["family"].startswith("dec")])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_x_bench.json").read()[-1500:])
PY
done
