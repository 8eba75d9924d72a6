# round 4, call D: XCD-local streamed input products (tests, A/B), phase stamps of the one-launch decoders and of the recurrent
# kernels (diagnostics build)
set -x
(timeout 900 python -m pytest tests/test_gpu_lstm.py -m gpu -x -q --timeout 600 2>&1 | tail -n 12)
for v in "" "LAS_LSTM_STREAM=0"; do
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/r04_d_bench.json
  python - "$v" <<PY
import json, sys
try:
    d=json.load(open("gpurun_out/r04_d_bench.json"))
    print("BENCH", sys.argv[1] or "default", d["ms_per_step"], d["value"], d["config"].get("final_loss"), d["config"]["step_form"], d["config"]["probe_s"], d.get("error"))
    print("   ", [(k["family"], k["ms_per_step"]) for k in d["roofline"]["kernels"]])
except Exception as e:
    print("BENCH", sys.argv[1], "FAILED", e, open("gpurun_out/r04_d_bench.json").read()[-800:])
PY
done
bash scripts/gpu_prof.sh r04_d_metricM --steps 3 --warmup 1 --no-graph > gpurun_out/r04_d_metricM.out 2>&1
python scripts/trace_timeline.py gpurun_out/r04_d_metricM > gpurun_out/r04_d_timeline.txt 2>&1
head -n 20 gpurun_out/r04_d_timeline.txt
(timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden_shapes.py -m gpu -q --timeout 600 2>&1 | tail -n 6)
export LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so
LAS_CXXFLAGS=-DLAS_STAMPS python phones-las_amd/build.py --force 2>&1 | tail -n 2
LAS_LSTM_STREAM=0 timeout 300 python scripts/gpu_dec_stamps.py 2>&1 | tail -n 40
LAS_LSTM_STREAM=0 H=256 timeout 300 python scripts/gpu_lstm_stamps.py 2>&1 | tail -n 40
