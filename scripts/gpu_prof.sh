# rocprofv3 kernel trace of a short bench run; summaries land in gpurun_out/<name>/
name=${1:-prof}
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$name.log 2>&1
cd $R
f=$(find gpurun_out/$name -name '*kernel_stats.csv' | head -1)
echo "stats file: $f"
head -40 "$f"
tail -3 gpurun_out/$name.log
