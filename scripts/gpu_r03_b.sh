# round 3, second pass: deterministic weight-gradient sums (A/B against the atomics), the whole GPU suite, bench lines
set -x
(time timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -15) 2>&1
for i in 1 2; do
LAS_TN_ATOMIC=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
done
for c in cfg5 metric-L cfg1 cfg4; do timeout 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700; done
