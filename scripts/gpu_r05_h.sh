# round 5, call h: whole GPU suite on the current tree; streamed input products against the product in front of the recurrence,
# three interleaved repetitions per configuration
set -x
(time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15) 2>&1 | tail -20
for rep in 1 2 3; do
for c in metric-M metric-M-stochastic metric-M-ragged default-arch cfg5; do
for st in 1 0; do
LAS_LSTM_STREAM=$st timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05_h_${c}_stream${st}_$rep.json
done; done; done
python - <<PY
import json,glob,collections
res=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05_h_*.json")):
    try:
        d=json.load(open(f)); k=f.split("r05_h_")[1].rsplit("_",1)[0]; res[k].append(d["ms_per_step"])
    except Exception as e:
        print(f,"FAILED",e)
for k,v in sorted(res.items()): print(k.ljust(36), " ".join("%.3f"%x for x in v), " median %.3f"%sorted(v)[len(v)//2])
PY
