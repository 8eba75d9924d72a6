set -x
timeout 300 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-420
timeout 600 python -u bench.py --config metric-L --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700
