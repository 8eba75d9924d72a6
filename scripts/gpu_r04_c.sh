# round 4, call C: streamed input products with write-through stores (no L2 write-back fence per tile): A/B against the fence,
# against sharing the chain's CUs, against no streaming; the convergence twin; the whole GPU suite
set -x
for v in "" "LAS_STREAM_FENCE=1" "LAS_STREAM_HOG_KB=0" "LAS_LSTM_STREAM=0" "LAS_LSTM_PREFETCH=0" ; do
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_c_bench.json
  python - "$v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_c_bench.json"))
    print("BENCH", sys.argv[1] or "default", d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d["config"]["probe_s"], d.get("error"))
    print("   ", [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_c_bench.json").read()[-800:])
PY
done
(time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --ignore tests/test_gpu_convergence.py 2>&1 | grep -E "passed|failed|Error|error" | tail -12) 2>&1 | tail -16
bash scripts/gpu_prof.sh r04_c_metricM --steps 3 --warmup 1 --no-graph > gpurun_out/r04_c_metricM.out 2>&1
python scripts/trace_timeline.py gpurun_out/r04_c_metricM > gpurun_out/r04_c_timeline.txt 2>&1
head -20 gpurun_out/r04_c_timeline.txt
