# round 5: the two directions' exposed bottom-layer products on streams that really run side by side (probed third stream): interleaved A/B
for i in 1 2 3; do for p in 0 1; do
echo -n "LAS_SIDE2_PROBE=$p: "; LAS_SIDE2_PROBE=$p timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('step_form'))"
done; done
for p in 0 1; do echo -n "metric-L LAS_SIDE2_PROBE=$p: "; LAS_SIDE2_PROBE=$p timeout 400 python bench.py --config metric-L --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('step_form'))"; done
