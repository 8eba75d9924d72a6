# round 5, call j: the step-form tests with their failure output; 512-unit backward decoder with one resident chunk fewer (no spills) A/B
set -x
timeout 1200 python -m pytest tests/test_gpu_step_forms.py -q --timeout 900 -x 2>&1 | tail -60 > gpurun_out/r05_j_forms.txt
timeout 1200 python -m pytest tests/test_gpu_step_forms.py -q --timeout 900 -k "cu_pressure" 2>&1 | tail -60 > gpurun_out/r05_j_cu.txt
for rep in 1 2; do for k in 5 4; do
LAS_DEC_BWD512_KRES=$k timeout 600 python bench.py --config metric-L --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_j_metric-L_kres${k}_$rep.json
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_j_*.json")):
    d=json.load(open(f)); ks={k["family"]:k["ms_per_step"] for k in d["roofline"]["kernels"]}
    print(f.split("r05_j_")[1][:-5].ljust(24), d["ms_per_step"], "dec_bwd=%.3f dec_fwd=%.3f" % (ks.get("dec_persist_bwd",0), ks.get("dec_persist_fwd",0)))
PY
