# round 4, call AA: compile-time-width Luong scores with q in registers (general forward body): benches, stamps, tests
set -x
for cfg in default-arch two-cell-bottom-only; do
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_aa_bench.json
  python - "$cfg" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_aa_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d.get("error"), [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"] if k["family"].startswith("dec")])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_aa_bench.json").read()[-1500:])
PY
done
(timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_stochastic.py -m gpu -q --timeout 600 -k "two_cell or general_decoder or one_launch or persistent or padding or embedding" 2>&1 | tail -n 4)
LAS_CXXFLAGS=-DLAS_STAMPS LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python phones-las_amd/build.py --force 2>&1 | tail -1
CFG=default-arch LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_dec2_stamps.py 2>&1 | grep -A17 "forward: step"
