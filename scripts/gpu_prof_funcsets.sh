#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_fs -o fs -- python $R/scripts/sr_funcsets.py > $OUT/60_funcsets.log 2>&1
python $R/scripts/rocpd_timeline.py $(find $OUT/prof_fs -name "*.db" | head -1) 4000 > $OUT/60_funcsets_timeline.txt
rm -rf $OUT/prof_fs
grep "^|" $OUT/60_funcsets.log
