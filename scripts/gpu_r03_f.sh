#!/bin/bash
# phase stamps of the sequential backward and the cfg5 step, register d(keys) on and off
LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_seq_bwd_stamps.py 2>&1 | grep -v amdgpu.ids
LAS_DEC_SEQ_REGK=0 LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_seq_bwd_stamps.py 2>&1 | grep -v amdgpu.ids
python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
LAS_DEC_SEQ_REGK=0 python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
python -m pytest tests/test_gpu_golden_shapes.py -x -q -k "cfg5 or one_launch or dec512" 2>&1 | tail -2
LAS_DEC_SEQ_REGK=0 python -m pytest tests/test_gpu_golden_shapes.py -x -q -k "cfg5 or one_launch" 2>&1 | tail -2
