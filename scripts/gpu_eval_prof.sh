#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for t in 16 64; do
EVOGP_EVAL_TPW=$t timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_ev$t -o ev -- python $R/scripts/eval_only.py 2>&1 | grep "len mean\|us per"
python $R/scripts/rocpd_summary.py $(find $OUT/prof_ev$t -name "*.db" | head -1) 2>&1 | head -6 | cut -c1-220
rm -rf $OUT/prof_ev$t
done
