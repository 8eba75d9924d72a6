#!/bin/bash
# four workgroups per utterance in the sequential backward: tests, stamps, cfg5 bench A/B
timeout 900 python -m pytest tests/test_gpu_golden_shapes.py tests/test_gpu_model.py -q -k "cfg5 or one_launch or bit_reproducible or long_memory" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | tail -8
LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so timeout 300 python scripts/gpu_seq_bwd_stamps.py 2>&1 | grep -v amdgpu.ids | head -9
timeout 300 python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('parts 4:', d['ms_per_step'], d['config'].get('final_loss'), d.get('error'))"
LAS_DEC_SEQ_PARTS=1 timeout 300 python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('parts 1:', d['ms_per_step'], d['config'].get('final_loss'), d.get('error'))"
