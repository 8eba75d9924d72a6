# round 4, call AD: workgroups of the bottom layer's (exposed) weight-gradient products: A/B on metric-M
set -x
for v in "LAS_TN_WGS_EXPOSED=704" "LAS_TN_WGS_EXPOSED=128" "LAS_TN_WGS_EXPOSED=256" "LAS_TN_WGS_EXPOSED=384" "LAS_TN_WGS_EXPOSED=512" "LAS_TN_WGS_EXPOSED=704"; do
  env $v timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_ad_bench.json
  python - "metric-M $v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_ad_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"]["step_form"], d.get("error"))
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_ad_bench.json").read()[-800:])
PY
done
