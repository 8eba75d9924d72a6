# round 4, call M: dispatch timeline of one train step of the reference's default architecture (launch count), and of two-cell-bottom-only
set -x
for cfg in default-arch two-cell-bottom-only; do
bash scripts/gpu_prof.sh r04_m_$cfg --config $cfg --steps 3 --warmup 1 --no-graph > gpurun_out/r04_m_$cfg.out 2>&1
python scripts/trace_timeline.py gpurun_out/r04_m_$cfg > gpurun_out/r04_m_${cfg}_timeline.txt
wc -l gpurun_out/r04_m_${cfg}_timeline.txt; tail -1 gpurun_out/r04_m_${cfg}_timeline.txt
done
