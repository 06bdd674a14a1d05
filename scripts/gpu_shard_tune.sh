#!/bin/bash
# sweep of the interpreter's work-distribution knobs on the shard sizes of an 8- / 4-rank run
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-tune}; mkdir -p $OUT; cd $R
{
python scripts/dbg/shard_tune.py 125000 250000
for ds in 0 1 2 3; do for st in 10 30 50 70; do
  EVOGP_TC_DYNSHIFT=$ds EVOGP_TC_STATIC=$st python scripts/dbg/shard_tune.py 125000 250000
done; done
} 2>&1 | grep trees > $OUT/${TAG}_shard_tune.log
cat $OUT/${TAG}_shard_tune.log
