# round 4, call B: streamed input products (tests, A/B), the input products moved behind the granule stores, the two tests that
# failed in call A with their details, the row-poison probe of the one-launch decoders
set -x
(timeout 900 python -m pytest tests/test_gpu_lstm.py -m gpu -x -q --timeout 600 2>&1 | tail -15)
(timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 600 -k "edge_shapes or reproducible" 2>&1 | tail -40)
(timeout 900 python -m pytest tests/test_gpu_golden_shapes.py -m gpu -q --timeout 600 2>&1 | tail -8)
for v in "" "LAS_LSTM_STREAM=0" "LAS_LSTM_FUSED_X=0" "LAS_LSTM_STREAM=0 LAS_LSTM_FUSED_X=0 LAS_TAIL_OVERLAP=0 LAS_DMEM_MERGED=0"; do
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_b_bench.json
  python - "$v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_b_bench.json"))
    print("BENCH", sys.argv[1] or "default", d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d["config"]["probe_s"], d.get("error"))
    print("   ", [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_b_bench.json").read()[-800:])
PY
done
for c in metric-M-ragged metric-L cfg4 cfg5 cfg1; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_b_bench_$c.json
  python - $c <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_b_bench_%s.json" % sys.argv[1]))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d.get("error"))
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_b_bench_%s.json" % sys.argv[1]).read()[-800:])
PY
done
timeout 600 python scripts/gpu_row_poison.py 2>&1 | tail -70
bash scripts/gpu_prof.sh r04_b_metricM --steps 3 --warmup 1 --no-graph > gpurun_out/r04_b_metricM.out 2>&1
python scripts/trace_timeline.py gpurun_out/r04_b_metricM > gpurun_out/r04_b_timeline.txt 2>&1
tail -3 gpurun_out/r04_b_timeline.txt
