#!/bin/bash
# forward decoder changes: decoder tests, cfg5 forward stamps, cfg5 / metric-L bench
timeout 900 python -m pytest tests/test_gpu_golden_shapes.py tests/test_gpu_model.py tests/test_gpu_speller_seam.py tests/test_gpu_binary_decoders.py -q 2>&1 | grep -E "passed|failed|^E  |^FAILED" | tail -8
CFG=cfg5 LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so timeout 300 python scripts/gpu_dec_stamps.py 2>&1 | grep -v amdgpu | head -11
timeout 300 python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5:', d['ms_per_step'], d['config'].get('final_loss'), d.get('error'))"
timeout 300 python bench.py --config metric-L --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('metric-L:', d['ms_per_step'], d['config'].get('final_loss'), d.get('error'))"
