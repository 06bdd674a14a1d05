R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
EVOGP_DEBUG_MARKS=1 timeout 300 python bench.py --steps 1 --warmup 0 --headline-only 2>&1 | grep "evogp\]" | sort | uniq -c | head -12 | cut -c1-900
