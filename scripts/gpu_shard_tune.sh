#!/bin/bash
# sweep of the interpreter's work-distribution knobs on the shard sizes of an 8- / 4-rank run:  bash scripts/gpu_shard_tune.sh TAG [batch sizes]
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-tune}; mkdir -p $OUT; cd $R
{
python scripts/dbg/shard_tune.py 125000 250000
for b in 2 4 8 16; do for ds in 0 1 2; do for st in 30 60; do
  EVOGP_TC_BATCH=$b EVOGP_TC_DYNSHIFT=$ds EVOGP_TC_STATIC=$st python scripts/dbg/shard_tune.py 125000 250000
done; done; done
} 2>&1 | grep trees > $OUT/${TAG}_shard_tune.log
cat $OUT/${TAG}_shard_tune.log
