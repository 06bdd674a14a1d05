#!/bin/bash
# one more PMC pass over the headline (scalar unit, LDS, branches, transcendentals):  gpurun -- 'bash scripts/gpu_pmc_extra.sh TAG'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-pmcx}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=3
for set in "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32" "SQ_BUSY_CYCLES SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmcx_$TAG$i -o pmc -- python $R/bench.py --steps 4 --warmup 1 --headline-only > $OUT/${TAG}_05_pmc$i.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/pmcx_$TAG$i -name "*.db" | head -1) 2>&1 | grep -A400 "counter" | grep -i "counter\|---\|sr_tc_kernel\|tc_compile_kernel" > $OUT/${TAG}_05_pmc$i.md
  rm -rf $OUT/pmcx_$TAG$i
  cat $OUT/${TAG}_05_pmc$i.md | cut -c1-170
done
