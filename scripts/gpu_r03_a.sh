# round 3, first pass: the cfg5 NaN fix, the full-size fixtures, the 30-step runs, bench lines of every config
set -x
STEPS=3 HOOK=0 python scripts/gpu_cfg5_nan.py 2>&1 | cut -c1-400 | tail -5
timeout 900 python -m pytest tests/test_gpu_golden_shapes.py tests/test_gpu_bench_shapes.py -q -x --timeout 600 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_gemm.py -q -x --timeout 300 2>&1 | tail -5
for c in cfg5 metric-M; do timeout 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900; done
