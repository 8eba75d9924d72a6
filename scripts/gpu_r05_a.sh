# round 5, call a: parity tests of the timed step forms + today's baseline of the headline (bench line, kernel table)
set -x
timeout 900 python -m pytest tests/test_gpu_step_forms.py -q -x --timeout 600 2>&1 | tail -15
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_a_bench_metricM.json
timeout 600 python bench.py --config metric-L --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_a_bench_metric-L.json
python - <<PY
import json
for c in ["metricM","metric-L"]:
    d=json.load(open("gpurun_out/r05_a_bench_%s.json"%c))
    print(c, d["ms_per_step"], d["value"], d["config"].get("step_form"), d["config"].get("probe_s"))
    for k in d["roofline"]["kernels"]: print("   ", k["family"], k["launches_per_step"], k["ms_per_step"], k["tflops"])
PY
