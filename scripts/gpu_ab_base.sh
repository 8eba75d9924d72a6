# A/B inside one gpurun call: ab_base/ (a git worktree of the previous commit, built) against the working tree
ARGS=${@:---no-graph}
for rep in 1 2 3; do
  for d in ab_base .; do
    (cd $d && python bench.py --no-cpu-baseline --steps 30 --warmup 5 $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', d['value'], d['ms_per_step'], d['config']['step_form'])")
  done
done
