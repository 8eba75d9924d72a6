set -x
( time timeout 600 python -u bench.py ) 2>&1 | tail -6
bash scripts/gpu_pmc.sh 2>&1 | tail -24
