cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
for m in 1 0; do
  EVOGP_TC_FUNC_MASK=$m timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_mask$m -o tr -- python $R/bench.py --steps 20 --warmup 3 --headline-only > $OUT/r03s_rocprof_mask$m.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/prof_mask$m -name "*.db" | head -1) > $OUT/r03s_kernel_stats_mask$m.md 2>&1
  rm -rf $OUT/prof_mask$m
  echo "== mask $m"; head -9 $OUT/r03s_kernel_stats_mask$m.md | cut -c1-140
  grep -o '"ms_per_step": [0-9.]*' $OUT/r03s_rocprof_mask$m.log
done
