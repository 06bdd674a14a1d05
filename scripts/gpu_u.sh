R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tc_wide.py tests/test_gpu_fuzz.py tests/test_gpu_ulp.py tests/test_gpu_mutation_parity.py -m gpu -q -x > $OUT/r03u_pytest.log 2>&1; tail -3 $OUT/r03u_pytest.log | cut -c1-250
EVOGP_DEBUG_MARKS=1 timeout 300 python bench.py --steps 1 --warmup 0 --headline-only 2>&1 | grep "evogp\]" | sort | uniq -c | head -5 | cut -c1-300
for m in 0 1; do EVOGP_TC_FUNC_MASK=$m timeout 900 python scripts/shard_model.py 2>&1 | grep trees | sed "s/^/mask $m: /"; done > $OUT/r03u_shard_model_mask.log; cat $OUT/r03u_shard_model_mask.log
