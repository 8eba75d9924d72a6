set -x
python __graft_entry__.py smoke 2>&1 | tail -3
python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -5
python bench.py --config tiny --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -5
python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline 2>&1 | tail -5
python bench.py --steps 5 --warmup 2 --cpu-sample 1 2>&1 | tail -5
