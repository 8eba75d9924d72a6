set -x
timeout 900 python -m pytest tests/test_gpu_lstm.py -q -x --timeout 600 2>&1 | tail -3
python scripts/gpu_lstm_time.py 2>&1 | grep "^H"
H=512 python scripts/gpu_lstm_time.py 2>&1 | grep "^H"
H=128 python scripts/gpu_lstm_time.py 2>&1 | grep "T 800"
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160; done
timeout 300 python bench.py --config metric-L --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --config cfg4 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
