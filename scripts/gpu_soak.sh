for cfg in metric-M metric-M-ragged metric-M-stochastic metric-L cfg4 cfg5 default-arch; do
  python bench.py --no-cpu-baseline --config $cfg --steps 3000 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', 'steps', d['steps'], 'ms_per_step', d['ms_per_step'], 'utt/s', d['value'], d['config'].get('step_form'), 'final_loss', d['config'].get('final_loss'), 'error', d.get('error'))"
done
