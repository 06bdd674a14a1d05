#!/bin/bash
# shard-size timings (scripts/dbg/shard_tune.py) of the default library and of every evogp_amd/lib/libevogp_hip_<variant>.so
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-vs}; mkdir -p $OUT; cd $R
{
echo "== default"; python scripts/dbg/shard_tune.py 125000 250000 500000 1000000
cp evogp_amd/lib/libevogp_hip.so /tmp/keep.so
for alt in $(ls evogp_amd/lib/libevogp_hip_*.so 2>/dev/null); do
  v=$(basename $alt .so); v=${v#libevogp_hip_}
  cp $alt evogp_amd/lib/libevogp_hip.so
  echo "== $v"; python scripts/dbg/shard_tune.py 125000 250000 500000 1000000
  cp /tmp/keep.so evogp_amd/lib/libevogp_hip.so
done
echo "== default again"; python scripts/dbg/shard_tune.py 125000 250000 500000 1000000
} 2>&1 | grep "trees\|==" > $OUT/${TAG}_variant_shards.log
cat $OUT/${TAG}_variant_shards.log
