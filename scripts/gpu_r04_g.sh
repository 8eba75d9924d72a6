# round 4, call G: input products behind the first polling round (single site), A/B; default-arch and two-cell configs; ragged
set -x
(timeout 900 python -m pytest tests/test_gpu_lstm.py -m gpu -x -q --timeout 600 2>&1 | tail -n 5)
for v in "" "LAS_LSTM_FUSED_X=0" "LAS_LSTM_STREAM=0 LAS_LSTM_FUSED_X=0"; do
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_g_bench.json
  python - "$v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_g_bench.json"))
    print("BENCH", sys.argv[1] or "default", d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d["config"]["probe_s"], d.get("error"))
    print("   ", [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_g_bench.json").read()[-800:])
PY
done
for c in metric-M-ragged default-arch two-cell-bottom-only cfg5 cfg1 cfg4; do
  timeout 900 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_g_bench_$c.json
  python - $c <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_g_bench_%s.json" % sys.argv[1]))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d["config"]["probe_s"], d.get("error"))
    print("   ", [(k["family"], k["launches_per_step"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_g_bench_%s.json" % sys.argv[1]).read()[-1500:])
PY
done
bash scripts/gpu_prof.sh r04_g_default --config default-arch --steps 2 --warmup 1 --no-graph > gpurun_out/r04_g_default.out 2>&1
head -n 30 gpurun_out/r04_g_default.out | cut -c1-200
