#!/bin/bash
# HBM traffic of the breeding pass and of the other pieces of a generation step (scripts/dbg/gen_step_parts.py) from the PMC counters,
# one counter per pass as MI355X_MICROARCH.md prescribes.   gpurun -- 'bash scripts/gpu_breed_pmc.sh TAG'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-breedpmc}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmcb_$TAG$i -o pmc -- python $R/scripts/dbg/gen_step_parts.py > $OUT/${TAG}_breed_pmc$i.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/pmcb_$TAG$i -name "*.db" | head -1) 2>&1 | grep -A400 "counter" | grep -i "counter\|---\|breed\|generate\|select\|replace" > $OUT/${TAG}_breed_pmc$i.md
  rm -rf $OUT/pmcb_$TAG$i
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_b$TAG -o tr -- python $R/scripts/dbg/gen_step_parts.py > $OUT/${TAG}_breed_prof.log 2>&1
python $R/scripts/rocpd_summary.py $(find $OUT/prof_b$TAG -name "*.db" | head -1) > $OUT/${TAG}_breed_kernel_stats.md 2>&1
rm -rf $OUT/prof_b$TAG
cd $R
cat $OUT/${TAG}_breed_pmc1.md $OUT/${TAG}_breed_pmc2.md | cut -c1-200; head -14 $OUT/${TAG}_breed_kernel_stats.md | cut -c1-170
