# round 4, call T: window tests after the instantiation split; embedded tokens in the two-cell launches (tests)
set -x
(timeout 1200 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_gemm.py -m gpu -q --timeout 600 -k "window" 2>&1 | tail -n 6)
(timeout 1500 python -m pytest tests/test_gpu_stochastic.py tests/test_gpu_model.py -m gpu -q --timeout 600 -k "embedding or two_cell or general_decoder" 2>&1 | tail -n 25)
