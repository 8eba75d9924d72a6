# round 5: 128 x 512 TN tiles only where the product is exposed (256-unit layers) against everywhere, interleaved
for rep in 1 2 3; do for e in "LAS_TN_WIDE_BESIDE_256=0" "LAS_TN_WIDE_BESIDE_256=1"; do for c in metric-M; do
  echo -n "$c $e: "; env $e python bench.py --config $c --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['step_form'], d['config'].get('final_loss'), [round(k['ms_per_step'],3) for k in d['roofline']['kernels'][:4]])"
done; done; done
for e in "LAS_TN_WIDE_BESIDE_256=0" "LAS_TN_WIDE_BESIDE_256=1"; do for c in cfg5 metric-M-stochastic; do
  echo -n "$c $e: "; env $e python bench.py --config $c --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['step_form'])"
done; done
timeout 600 python -m pytest tests/test_gpu_step_forms.py tests/test_gpu_lstm.py -q -m gpu -k "plain_forms or tail or bilstm_forward" 2>&1 | grep -E "^E  |passed|failed"
