# round 5: how much do the products beside the chains cost the chains?  (LAS_SERIAL=1: no side streams; LAS_LSTM_STREAM=0: input products in front of their chains)
for e in "X=1" "LAS_SERIAL=1" "LAS_SERIAL=1 LAS_LSTM_STREAM=0" "LAS_LSTM_STREAM=0" "X=1"; do
  for c in metric-M metric-L; do
    echo "== $e $c"; env $e timeout 400 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('step_form'), [(k['name'][:28], round(k['ms_per_step'],3)) for k in d['roofline'].get('kernels', [])][:8])"
  done
done
