set -x
timeout 600 python -m pytest tests -m gpu -q -x --timeout 120 2>&1 | tail -6
timeout 300 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -2
bash scripts/gpu_prof.sh ${1:-prof_x} --steps 2 --warmup 1 --no-graph > gpurun_out/${1:-prof_x}.out 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/${1:-prof_x}/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:80].ljust(80), r["Calls"], round(int(r["TotalDurationNs"])/1e6,2), round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
