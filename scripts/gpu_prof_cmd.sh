#!/bin/bash
# kernel statistics (rocprofv3 --kernel-trace --stats) of an arbitrary command:  gpurun -- 'bash scripts/gpu_prof_cmd.sh TAG python scripts/x.py args'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=$1; shift; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o tr -- "$@" > $OUT/${TAG}_prof_cmd.log 2>&1
python $R/scripts/rocpd_summary.py $(find $OUT/prof_$TAG -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats.md 2>&1
rm -rf $OUT/prof_$TAG; cd $R
head -12 $OUT/${TAG}_kernel_stats.md | cut -c1-160
