#!/bin/bash
# cfg5 after a decoder-backward change: parity tests, phase stamps, bench A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_golden_shapes.py -x -q -k "cfg5 or one_launch or dec512" > gpurun_out/e_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/e_tests.log
LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_seq_bwd_stamps.py > gpurun_out/e_stamps.log 2>&1
python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/e_bench_cfg5.log 2>&1
tail -5 gpurun_out/e_tests.log; cat gpurun_out/e_stamps.log; tail -1 gpurun_out/e_bench_cfg5.log | cut -c1-400
LAS_DEC_SEQ_REGK=0 python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
