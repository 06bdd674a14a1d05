#!/bin/bash
# What a fitness call costs when there is almost nothing to do: kernel durations (rocprofv3 --kernel-trace) of calls on 4 k ... 64 k trees
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
: > $OUT/launch_floor.txt
for n in ${SIZES:-4096 16384 65536}; do
  timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_fl_$n -o tl -- python $R/scripts/dbg/shard_timeline.py $n > $OUT/fl_$n.log 2>&1
  echo "== $n trees: $(grep 'ms per call' $OUT/fl_$n.log)" >> $OUT/launch_floor.txt
  python $R/scripts/rocpd_timeline.py $(find $OUT/prof_fl_$n -name "*.db" | head -1) 6 >> $OUT/launch_floor.txt
  rm -rf $OUT/prof_fl_$n
done
cat $OUT/launch_floor.txt
