for rep in 1 2 3; do
  for us in 0 12 25; do
    LAS_SIDE_DELAY_US=$us python bench.py --no-cpu-baseline --steps 30 --warmup 5 --no-graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('delay_us=$us', d['value'], d['ms_per_step'])"
  done
  (cd ab_base && python bench.py --no-cpu-baseline --steps 30 --warmup 5 --no-graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['value'], d['ms_per_step'])")
done
