# kernel traces of the working tree and of ab_base/ in one call -> gpurun_out/ab_new, gpurun_out/ab_old
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  d=$R; [ $v = old ] && d=$R/ab_base
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$v -- python3 $d/bench.py --no-cpu-baseline --steps 6 --warmup 3 --no-graph > $R/gpurun_out/ab_$v.log 2>&1
done
cd $R
for v in new old; do echo == $v; python3 scripts/trace_timeline.py gpurun_out/ab_$v | tail -1; done
