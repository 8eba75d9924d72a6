# round 3 profiles: rocprofv3 kernel stats of metric-M and metric-L (eager launches), PMC passes of metric-M, the headline bench line
set -x
bash scripts/gpu_prof.sh r03_a_metricM --steps 3 --warmup 1 --no-graph > gpurun_out/r03_a_metricM.out 2>&1
bash scripts/gpu_prof.sh r03_a_metricL --config metric-L --steps 3 --warmup 1 --no-graph > gpurun_out/r03_a_metricL.out 2>&1
bash scripts/gpu_pmc.sh r03 > gpurun_out/r03_pmc.out 2>&1
timeout 600 python bench.py --steps 30 --warmup 5 2>&1 | tail -1 > gpurun_out/r03_bench_a.json
tail -5 gpurun_out/r03_pmc.out; cut -c1-300 gpurun_out/r03_bench_a.json
