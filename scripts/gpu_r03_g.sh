#!/bin/bash
# forward decoder stamps (cfg5), cfg5 bench, decoder test files
CFG=cfg5 LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_dec_stamps.py 2>&1 | grep -v amdgpu | head -11
python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
python -m pytest tests/test_gpu_golden_shapes.py tests/test_gpu_model.py tests/test_gpu_binary_decoders.py -x -q 2>&1 | tail -4
