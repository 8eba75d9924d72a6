set -x
timeout 1500 python -m pytest tests/test_gpu_step_forms.py -q --timeout 900 -k "plain_forms or tail_split or data_parallel" -x > gpurun_out/r05_k_dp.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_step_forms.py -q --timeout 900 -k "cu_pressure" > gpurun_out/r05_k_cu.txt 2>&1
grep -n "passed\|failed" gpurun_out/r05_k_dp.txt gpurun_out/r05_k_cu.txt
