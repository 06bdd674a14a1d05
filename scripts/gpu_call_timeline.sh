#!/bin/bash
# Kernel timeline of back-to-back fitness calls on 125 k and on 1 M trees (scripts/dbg/shard_timeline.py under rocprofv3 --kernel-trace)
# -> gpurun_out/TAG_call_timeline.txt.   gpurun -- 'bash scripts/gpu_call_timeline.sh TAG'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-tl}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
: > $OUT/${TAG}_call_timeline.txt
for n in 125000 1000000; do
  timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_tl_$n -o tl -- python $R/scripts/dbg/shard_timeline.py $n > $OUT/${TAG}_tl_$n.log 2>&1
  echo "== $n trees: $(grep 'ms per call' $OUT/${TAG}_tl_$n.log)" >> $OUT/${TAG}_call_timeline.txt
  python $R/scripts/rocpd_timeline.py $(find $OUT/prof_tl_$n -name "*.db" | head -1) 18 >> $OUT/${TAG}_call_timeline.txt
  rm -rf $OUT/prof_tl_$n
done
tail -20 $OUT/${TAG}_call_timeline.txt
