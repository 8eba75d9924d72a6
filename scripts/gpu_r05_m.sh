set -x
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_lstm.py -q --timeout 600 2>&1 | tail -3 > gpurun_out/r05_m_tests.txt
cat gpurun_out/r05_m_tests.txt
for pp in 0 1; do echo "wide pp=$pp"; LAS_TN_PP=$pp SPLIT=32,32,24 timeout 300 python scripts/gpu_gemm_time.py 2>&1 | grep " TN "; done
run() { n=$1; c=$2; shift; shift; env "$@" timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_m_$n.json; }
for rep in 1 2; do for c in metric-M metric-L; do for w in 0 1; do run ${c}_pp${w}_$rep $c LAS_TN_PP=$w; done; done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_m_*.json")):
    try:
        d=json.load(open(f)); ks={k["family"]:k["ms_per_step"] for k in d["roofline"]["kernels"]}
        print(f.split("r05_m_")[1][:-5].ljust(22), d["ms_per_step"], d["config"].get("final_loss"), " ".join("%s=%.3f"%(k,v) for k,v in ks.items()))
    except Exception as e: print(f,"FAILED",e)
PY
