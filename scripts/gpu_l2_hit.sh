#!/bin/bash
# L2 hit / miss counts of a metric-L step's kernels (one --pmc pass; question: do the streamed weight chunks of the 512-unit
# one-launch forward decoder hit the L2?)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l2 -- python3 $R/bench.py --config metric-L --no-cpu-baseline --steps 1 --warmup 0 --no-graph > $R/gpurun_out/pmc_l2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/pmc_l2/*/*counter_collection.csv'))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:70]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('TCC_REQ_sum', 0))[:12]:
    h, m, q = v.get('TCC_HIT_sum', 0), v.get('TCC_MISS_sum', 0), v.get('TCC_REQ_sum', 0)
    print('%-70s req %.3g hit %.3g miss %.3g hit rate %.3f' % (k, q, h, m, h / max(h + m, 1)))
PY
tail -2 gpurun_out/pmc_l2.log | cut -c1-200
