# round 3, final measurements: GPU suite, rocprofv3 kernel stats (metric-M, metric-L, cfg5), PMC passes of metric-M, bench lines of every config
set -x
(time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed" | tail -3) 2>&1 | tail -6
bash scripts/gpu_prof.sh r03_d_metricM --steps 3 --warmup 1 --no-graph > gpurun_out/r03_d_metricM.out 2>&1
bash scripts/gpu_prof.sh r03_d_metricL --config metric-L --steps 3 --warmup 1 --no-graph > gpurun_out/r03_d_metricL.out 2>&1
bash scripts/gpu_prof.sh r03_d_cfg5 --config cfg5 --steps 3 --warmup 1 --no-graph > gpurun_out/r03_d_cfg5.out 2>&1
bash scripts/gpu_pmc.sh r03 > gpurun_out/r03_pmc.out 2>&1
mkdir -p profiles && cp gpurun_out/r03_pmc_traffic.json profiles/r03_pmc_traffic.json
timeout 900 python bench.py --steps 30 --warmup 5 2>&1 | tail -1 > gpurun_out/r03_bench_final.json
for c in metric-M-stochastic metric-L cfg4 cfg5 cfg1; do timeout 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_$c.json; done
python - <<PY
import json
for c in ["final","metric-M-stochastic","metric-L","cfg4","cfg5","cfg1"]:
    d=json.load(open("gpurun_out/r03_bench_%s.json"%c))
    print(c, d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic_stale"), d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("all_core_run"))
PY
