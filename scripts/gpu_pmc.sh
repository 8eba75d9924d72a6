# HBM traffic and MFMA-busy cycles of the dominant kernel from PMC counters: one counter per pass (FETCH_SIZE, WRITE_SIZE,
# SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)
# (MI355X_MICROARCH.md §HBM / rocprofv3 PMC slots).  Results -> gpurun_out/pmc_*/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-graph > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob
for c in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'):
    fs = glob.glob('gpurun_out/pmc_%s/*/*counter_collection.csv' % c)
    if not fs:
        print(c, 'no counter file', glob.glob('gpurun_out/pmc_%s/*/*' % c)); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = {}
    for r in rows:
        k = r['Kernel_Name'][:60]
        agg.setdefault(k, []).append(float(r['Counter_Value']))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(c, k.ljust(60), 'n=%d' % len(v), 'avg=%.1f' % (sum(v) / len(v)), 'max=%.1f' % max(v))
PY
