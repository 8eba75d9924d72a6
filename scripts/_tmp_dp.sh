for e in 1 2 3; do
timeout 600 python -m pytest tests/test_gpu_step_forms.py -q -m gpu 2>&1 | grep "AssertionError\|passed\|failed" | head -5
done
