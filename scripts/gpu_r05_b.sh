# round 5, call b: follower products (kernel alone, listener with / without), the touched suites, bench A/B through LAS_FOLLOW
set -x
timeout 900 python -m pytest tests/test_gpu_follow.py -q -x --timeout 600 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_step_forms.py tests/test_gpu_lstm.py tests/test_gpu_gemm.py tests/test_abi.py -q --timeout 600 2>&1 | tail -8
for f in 1 0; do
LAS_FOLLOW=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_b_bench_metricM_follow$f.json
LAS_FOLLOW=$f timeout 600 python bench.py --config metric-L --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_b_bench_metric-L_follow$f.json
done
LAS_FOLLOW_BWD=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_b_bench_metricM_followfwd.json
python - <<PY
import json
for c in ["metricM_follow1","metricM_follow0","metricM_followfwd","metric-L_follow1","metric-L_follow0"]:
    try:
        d=json.load(open("gpurun_out/r05_b_bench_%s.json"%c))
        print(c, d["ms_per_step"], d["value"], d["config"].get("step_form"), d["config"].get("probe_s"), d["config"].get("final_loss"), d.get("error"))
        for k in d["roofline"]["kernels"]: print("   ", k["family"], k["launches_per_step"], k["ms_per_step"], k["tflops"])
    except Exception as e:
        print(c, "FAILED", e)
PY
