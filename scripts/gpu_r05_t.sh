# round 5: forward recurrence polls two adjacent granules per 16-byte load (default) against one per load (liblas_hip_narrow.so, built with
# LAS_CXXFLAGS=-DLAS_FWD_NARROW_POLL LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_narrow.so python phones-las_amd/build.py)
N=$PWD/phones-las_amd/liblas_hip_narrow.so
timeout 900 python -m pytest tests/test_gpu_lstm.py -q -m gpu 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head
for i in 1 2; do
echo "wide";   for h in 256 512; do H=$h timeout 300 python scripts/gpu_lstm_time.py 2>&1 | grep "^H" | head -2; done
echo "narrow"; for h in 256 512; do LAS_HIP_LIB=$N H=$h timeout 300 python scripts/gpu_lstm_time.py 2>&1 | grep "^H" | head -2; done
done
for i in 1 2 3; do
echo -n "metric-M wide: ";   timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('final_loss'))"
echo -n "metric-M narrow: "; LAS_HIP_LIB=$N timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('final_loss'))"
done
for i in 1 2; do
echo -n "metric-L wide: ";   timeout 400 python bench.py --config metric-L --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('final_loss'))"
echo -n "metric-L narrow: "; LAS_HIP_LIB=$N timeout 400 python bench.py --config metric-L --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('final_loss'))"
done
