# round 5: l2_norm_kernel's last-arriver reduce reads only the rows that can hold a tensor's partials: tests, kernel time at metric-L / metric-M
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_step_forms.py tests/test_gpu_dp.py -q -m gpu -k "train_op or adam or reproducible or forms or tail or dp or exchange" 2>&1 | tail -3
for c in metric-L metric-M; do
bash scripts/gpu_prof.sh r05_q_$c --config $c --steps 3 --warmup 1 --no-graph > gpurun_out/r05_q_$c.out 2>&1
f=$(find gpurun_out/r05_q_$c -name '*kernel_stats.csv' | head -1)
grep -E "l2_norm_kernel|clip_adam_kernel" $f | cut -c1-60,200-400
python scripts/trace_timeline.py gpurun_out/r05_q_$c | tail -12
done
for c in metric-L cfg4 metric-M; do timeout 400 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], d['config'].get('step_form'))"; done
