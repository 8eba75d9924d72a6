for rep in 1 2; do
for v in "1 1" "0 1" "1 0" "0 0"; do set -- $v
  LAS_REFRESH_BESIDE=$1 LAS_TN_EXPOSED=$2 python bench.py --no-cpu-baseline --steps 30 --warmup 5 --no-graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('refresh_beside=$1 tn_exposed=$2', d['value'], d['ms_per_step'])"
done; done
