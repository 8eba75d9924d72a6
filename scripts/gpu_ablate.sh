for a in 0 1 2 3 4 7; do
  LAS_ABLATE=$a timeout 120 python -c "
import bench, torch
c=bench.CONFIGS['metric-M']
import sys
try:
    ms,fl=bench.time_dominant_kernel(c)
    print('ablate $a: fwd kernel ms', round(ms,3), 'us/step', round(ms*1e3/800,3))
except Exception as e:
    print('ablate $a failed', e)
" 2>&1 | grep ablate
done
