#!/bin/bash
# monotonic chain in registers: decoder tests, stamps of both directions, cfg5 bench
python -m pytest tests/test_gpu_golden_shapes.py tests/test_gpu_model.py tests/test_gpu_binary_decoders.py tests/test_gpu_speller_seam.py tests/test_oracle_speller.py -q 2>&1 | grep -E "passed|failed|^E  " | tail -5
CFG=cfg5 LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_dec_stamps.py 2>&1 | grep -v amdgpu | head -11
LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_seq_bwd_stamps.py 2>&1 | grep -v amdgpu.ids | head -9
python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
