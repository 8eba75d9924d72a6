# HBM traffic and MFMA-busy cycles of the step's kernels from PMC counters: one counter per pass (FETCH_SIZE, WRITE_SIZE,
# SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do
# not fit one pass).  Results -> gpurun_out/pmc_*/ ; scripts/pmc_summary.py turns them into profiles/<round>_pmc_traffic.json.
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${1:-r03}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-graph > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python3 scripts/pmc_summary.py $ROUND
