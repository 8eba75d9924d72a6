# round 5: workgroups of the weight-gradient products beside the backward chains (LAS_TN_WGS) and of the exposed pair (LAS_TN_WGS_EXPOSED), metric-M, interleaved
for rep in 1 2; do for w in 704 352 1056 1408; do
  echo -n "LAS_TN_WGS=$w: "; LAS_TN_WGS=$w python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [round(k['ms_per_step'],3) for k in d['roofline']['kernels'][:4]])"
done; done
for rep in 1 2; do for w in 704 352 1408; do
  echo -n "LAS_TN_WGS_EXPOSED=$w: "; LAS_TN_WGS_EXPOSED=$w python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [round(k['ms_per_step'],3) for k in d['roofline']['kernels'][:4]])"
done; done
