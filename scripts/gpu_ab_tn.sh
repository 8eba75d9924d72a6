for rep in 1 2; do
for w in 704 352 512 1024; do
  LAS_TN_WGS=$w python bench.py --no-cpu-baseline --steps 30 --warmup 5 --no-graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('metric-M tn_wgs=$w', d['value'], d['ms_per_step'])"
done; done
for w in 704 352 1024; do
  LAS_TN_WGS=$w python bench.py --no-cpu-baseline --config metric-L --steps 10 --warmup 3 --no-graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('metric-L tn_wgs=$w', d['value'], d['ms_per_step'])"
done
