R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
EVOGP_TC_HINTS=0 timeout 600 python tests/tools/hints_check.py > $OUT/r03l_hints0.log 2>&1; tail -3 $OUT/r03l_hints0.log | cut -c1-400
EVOGP_TC_HINTS=1 timeout 600 python tests/tools/hints_check.py > $OUT/r03l_hints1.log 2>&1; tail -3 $OUT/r03l_hints1.log | cut -c1-400
