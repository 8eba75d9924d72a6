set -x
timeout 600 python -m pytest tests -m gpu -q -x --timeout 120 2>&1 | tail -15
timeout 300 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -3
