R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_gpu_breed.py tests/test_gpu_rollout.py -m gpu -q -x > $OUT/r03p_pytest.log 2>&1; tail -3 $OUT/r03p_pytest.log | cut -c1-250
timeout 300 python scripts/dbg/select_time.py 2>&1 | grep us > $OUT/r03p_select_time.log; cat $OUT/r03p_select_time.log
