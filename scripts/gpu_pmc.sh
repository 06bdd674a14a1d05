#!/bin/bash
# PMC counters for the fitness kernel (separate passes, kernel-trace only — never combined with sys/hip traces)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/30_counters.txt 2>&1
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/31_pmc$i.log 2>&1
  echo "pmc$i rc=$?" >> $OUT/31_pmc$i.log
  python $R/scripts/rocpd_summary.py $(find $OUT/pmc$i -name "*.db" | head -1) > $OUT/32_pmc$i.md 2>&1
  rm -rf $OUT/pmc$i
  tail -3 $OUT/31_pmc$i.log | cut -c1-300
  grep -i "sr_tc_kernel\|tc_compile\|sr_fast\|sr_general" $OUT/32_pmc$i.md | cut -c1-250
done
ls $OUT
