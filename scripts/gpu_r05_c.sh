# round 5, call c: follower throttle sweep (persistent workgroups), forward-only / backward-only, against streaming and neither
set -x
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_c_$n.json
}
run base LAS_FOLLOW=0
run nostream LAS_FOLLOW=0 LAS_LSTM_STREAM=0
for w in 16 32 48 96; do run fwd$w LAS_FOLLOW_BWD=0 LAS_FOLLOW_WGS=$w; done
for w in 16 32 96; do run both$w LAS_FOLLOW_WGS=$w; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_c_*.json")):
    try:
        d=json.load(open(f))
        ks={k["family"]:k["ms_per_step"] for k in d["roofline"]["kernels"]}
        print(f.split("r05_c_")[1][:-5].ljust(10), d["ms_per_step"], d["config"].get("step_form"), " ".join("%s=%.3f"%(k,v) for k,v in ks.items()))
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 600 python -m pytest tests/test_gpu_step_forms.py -q -x --timeout 600 -k data_parallel 2>&1 | tail -40 > gpurun_out/r05_c_dp.txt
