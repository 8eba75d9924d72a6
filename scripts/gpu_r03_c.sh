# round 3: 512-unit chains as 8 members (default) against 16 members, alone and in the metric-L / cfg4 step
set -x
timeout 900 python -m pytest tests/test_gpu_lstm.py -q -x --timeout 600 2>&1 | tail -3
for g in 16 8; do
  LAS_LSTM_G512=$g H=512 python scripts/gpu_lstm_time.py 2>&1 | grep "^H"
  for r in 4 8 16; do LAS_LSTM_ROWS=$r LAS_LSTM_G512=$g H=512 python scripts/gpu_lstm_time.py 2>&1 | grep "T 800" | sed "s/^/rows $r: /"; done
done
for g in 16 8 16 8; do LAS_LSTM_G512=$g timeout 300 python bench.py --config metric-L --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200; done
timeout 600 python -m pytest tests/test_gpu_golden_shapes.py -q -x --timeout 600 -k "metricL or dec512" 2>&1 | tail -3
