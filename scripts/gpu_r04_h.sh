# round 4, call H: the NT ring kernel with four waves of 128 x 128 (LAS_GEMM_RING=6) against the eight-wave forms
set -x
for r in 5 6; do LAS_GEMM_RING=$r timeout 300 python scripts/gpu_nt_ab.py 2>&1 | tail -n 8; done
LAS_GEMM_RING=6 timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -x -q 2>&1 | tail -n 4
for r in 5 6; do
  LAS_GEMM_RING=$r timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_h_bench.json
  python - "ring=$r" <<PY
import json, sys
d=json.load(open("gpurun_out/r04_h_bench.json"))
print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d.get("error"))
print("   ", [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
PY
  LAS_GEMM_RING=$r timeout 600 python bench.py --config metric-L --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_h_bench.json
  python - "metric-L ring=$r" <<PY
import json, sys
d=json.load(open("gpurun_out/r04_h_bench.json"))
print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d.get("error"))
print("   ", [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
PY
done
