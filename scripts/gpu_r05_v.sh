# round 5: which of the round's GEMM switches pays at metric-M?  (interleaved, one call)
for rep in 1 2 3; do for e in "X=1" "LAS_TN_WIDE=0" "LAS_GEMM_PP=0" "LAS_TN_WIDE=0 LAS_GEMM_PP=0"; do
  echo -n "metric-M $e: "; env $e python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['step_form'], [round(k['ms_per_step'],3) for k in d['roofline']['kernels'][:4]])"
done; done
