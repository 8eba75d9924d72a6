# round 4, call S: the one-launch backward chain as its own instantiation again (128-unit regression): tests + benches
set -x
(timeout 1500 python -m pytest tests/test_gpu_lstm.py -m gpu -q --timeout 600 2>&1 | tail -n 5)
for cfg in cfg1 default-arch metric-M; do
  timeout 600 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04_s_bench.json
  python - "$cfg" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_s_bench.json"))
    print("BENCH", sys.argv[1], d["ms_per_step"], d["value"], d["config"]["step_form"], d.get("error"), [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"][:5]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_s_bench.json").read()[-800:])
PY
done
