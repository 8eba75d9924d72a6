# round 5, call f: ping-pong schedules of the NT / TN rings: correctness (GEMM suite under both), products alone, in the train step
set -x
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_follow.py tests/test_gpu_lstm.py -q --timeout 600 2>&1 | tail -5
LAS_TN_PP=0 timeout 300 python scripts/gpu_gemm_time.py 2>&1 | grep TN
LAS_TN_PP=1 timeout 300 python scripts/gpu_gemm_time.py 2>&1 | grep TN
run() { n=$1; c=$2; shift; shift; env "$@" timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_f_$n.json; }
for c in metric-M metric-L; do
run ${c}_pp00 $c LAS_TN_PP=0 LAS_GEMM_PP=0
run ${c}_pp10 $c LAS_TN_PP=1 LAS_GEMM_PP=0
run ${c}_pp01 $c LAS_TN_PP=0 LAS_GEMM_PP=1
run ${c}_pp11 $c LAS_TN_PP=1 LAS_GEMM_PP=1
run ${c}_pp00b $c LAS_TN_PP=0 LAS_GEMM_PP=0
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_f_*.json")):
    try:
        d=json.load(open(f))
        ks={k["family"]:k["ms_per_step"] for k in d["roofline"]["kernels"]}
        print(f.split("r05_f_")[1][:-5].ljust(16), d["ms_per_step"], d["config"].get("step_form"), d["config"].get("final_loss"), " ".join("%s=%.3f"%(k,v) for k,v in ks.items()))
    except Exception as e:
        print(f, "FAILED", e)
PY
