# round 5, measurement set: GPU suite, rocprofv3 kernel stats (metric-M, metric-L, cfg5, two-cell), PMC passes of metric-M, bench lines of every config
set -x
T=${1:-r05_z}
if [ "${2:-suite}" = suite ]; then
(time timeout 1800 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|error" | tail -5) 2>&1 | tail -8
fi
bash scripts/gpu_prof.sh ${T}_metricM --steps 3 --warmup 1 --no-graph > gpurun_out/${T}_metricM.out 2>&1
bash scripts/gpu_prof.sh ${T}_metricL --config metric-L --steps 3 --warmup 1 --no-graph > gpurun_out/${T}_metricL.out 2>&1
bash scripts/gpu_prof.sh ${T}_cfg5 --config cfg5 --steps 3 --warmup 1 --no-graph > gpurun_out/${T}_cfg5.out 2>&1
bash scripts/gpu_prof.sh ${T}_twocell --config two-cell-bottom-only --steps 3 --warmup 1 --no-graph > gpurun_out/${T}_twocell.out 2>&1
bash scripts/gpu_pmc.sh r05 > gpurun_out/r05_pmc.out 2>&1
mkdir -p profiles && cp gpurun_out/r05_pmc_traffic.json profiles/r05_pmc_traffic.json
timeout 1200 python bench.py --steps 30 --warmup 5 2>&1 | tail -1 > gpurun_out/r05_bench_metricM.json
for c in metric-M-ragged metric-M-stochastic metric-L cfg4 cfg5 cfg1 default-arch two-cell-bottom-only; do timeout 400 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_bench_$c.json; done
python - <<PY
import json
for c in ["metricM","metric-M-ragged","metric-M-stochastic","metric-L","cfg4","cfg5","cfg1","default-arch","two-cell-bottom-only"]:
    try:
        d=json.load(open("gpurun_out/r05_bench_%s.json"%c))
        print(c, d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"].get("step_form"), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic_stale"), d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("thread_sweep"), d.get("error"))
    except Exception as e:
        print(c, "FAILED", e)
PY
