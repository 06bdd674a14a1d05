R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
export OPS_JSON=$OUT/sq_ops.json ROCPD_BY_GRID=1
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM"; do
  n=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/sqpmc_$n -o pmc -- python $R/scripts/ops_pmc.py run > $OUT/sq_$n.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/sqpmc_$n -name "*.db" | head -1) > $OUT/sq_$n.md 2>&1
  rm -rf $OUT/sqpmc_$n
  grep -i "eval_direct\|generate_staged\|kernel |" $OUT/sq_$n.md | cut -c1-400
done
