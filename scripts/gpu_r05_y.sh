# round 5, end: re-check of the knobs around the streamed products after the forward kernels' polls changed (metric-M, interleaved)
for rep in 1 2; do for e in "X=1" "LAS_STREAM_GRID=128" "LAS_STREAM_GRID=256" "LAS_SIDE_DELAY_US=4" "LAS_SIDE_DELAY_US=24" "LAS_GEMM_PP=2" "LAS_LSTM_STREAM=0"; do
  echo -n "$e: "; env $e python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [round(k['ms_per_step'],3) for k in d['roofline']['kernels'][:4]])"
done; done
