# round 5 against the end of round 4 (ab_base/ = a worktree of commit e446358, built), interleaved inside one call
for c in metric-M metric-L cfg4; do for rep in 1 2; do for d in ab_base .; do
  (cd $d && python bench.py --config $c --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', '$d' == '.' and 'round 5' or 'round 4', d['ms_per_step'], d['value'], d['config']['step_form'])")
done; done; done
