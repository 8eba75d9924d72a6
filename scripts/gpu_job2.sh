set -x
nproc
( time timeout 300 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline ) 2>&1 | tail -8
( time timeout 300 python -u -c "
import bench
print(bench.cpu_baseline(bench.CONFIGS['metric-M'], 4))
" ) 2>&1 | tail -8
