#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_tl -o tl -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/scripts/rocpd_timeline.py $(find $OUT/prof_tl -name "*.db" | head -1) 400 > $OUT/50_timeline.txt
rm -rf $OUT/prof_tl
grep -n "sr_tc_kernel<8, false, 2" $OUT/50_timeline.txt | head -3
