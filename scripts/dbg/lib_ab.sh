#!/bin/bash
# A/B of engine builds on small calls: evogp_amd/lib/libevogp_hip.so against every libevogp_hip_<variant>.so (scripts/build_variant.sh), two
# rounds, tree_SR_fitness in a loop of 200 calls on SIZES trees (scripts/dbg/shard_timeline.py).   gpurun -- 'bash scripts/dbg/lib_ab.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
cp evogp_amd/lib/libevogp_hip.so /tmp/keep.so
{
for round in 1 2; do
  for lib in default $(ls evogp_amd/lib/libevogp_hip_*.so 2>/dev/null); do
    v=$(basename $lib .so); v=${v#libevogp_hip_}
    [ "$lib" = default ] && cp /tmp/keep.so evogp_amd/lib/libevogp_hip.so || cp $lib evogp_amd/lib/libevogp_hip.so
    for n in ${SIZES:-4096 16384 100000}; do echo "$v: $(python scripts/dbg/shard_timeline.py $n 2>&1 | grep 'ms per call')"; done
  done
done
cp /tmp/keep.so evogp_amd/lib/libevogp_hip.so
} > $OUT/lib_ab.log 2>&1
sort -s -k1,1 $OUT/lib_ab.log
