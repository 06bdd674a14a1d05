R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 900 python scripts/shard_model.py > $OUT/r03j_shard_model.log 2>&1; grep trees $OUT/r03j_shard_model.log
EVOGP_TC_HINTS=0 timeout 900 python scripts/shard_model.py 2>&1 | grep trees | sed 's/^/hints off: /' > $OUT/r03j_shard_model_nohints.log; cat $OUT/r03j_shard_model_nohints.log
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tc_wide.py tests/test_gpu_fuzz.py -m gpu -q -x > $OUT/r03j_pytest.log 2>&1; tail -4 $OUT/r03j_pytest.log | cut -c1-250
