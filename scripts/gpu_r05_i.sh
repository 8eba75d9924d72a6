# round 5, call i: the reworked tests; phase stamps of the recurrent kernels and of the one-launch decoder from a diagnostics build
set -x
timeout 1200 python -m pytest tests/test_gpu_step_forms.py tests/test_gpu_model.py tests/test_fast_input.py tests/test_abi.py -q --timeout 900 -k "cu_pressure or binf or fast or abi or data_parallel or plain_forms" 2>&1 | tail -8
LAS_CXXFLAGS=-DLAS_STAMPS LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python phones-las_amd/build.py --force 2>&1 | tail -1
for H in 256 512; do H=$H LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so timeout 300 python scripts/gpu_lstm_stamps.py; done > gpurun_out/r05_lstm_stamps.txt 2>&1
LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so timeout 300 python scripts/gpu_dec_stamps.py > gpurun_out/r05_dec_stamps_metricM.txt 2>&1
cat gpurun_out/r05_lstm_stamps.txt gpurun_out/r05_dec_stamps_metricM.txt | tail -70
timeout 300 python scripts/gpu_lstm_time.py
