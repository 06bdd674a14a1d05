R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
for u in 1 4; do EVOGP_BREED_UNIT=$u timeout 600 python scripts/dbg/gen_step_parts.py 250000 500000 1000000 2>&1 | grep "^pop" | sed "s/^/unit $u: /"; done > $OUT/r03k_breed_unit.log 2>&1; cut -c1-200 $OUT/r03k_breed_unit.log
timeout 900 python scripts/shard_model.py 2>&1 | grep trees > $OUT/r03k_shard_model.log; cat $OUT/r03k_shard_model.log
EVOGP_BREED_UNIT=4 timeout 1800 python -m pytest tests/test_gpu_breed.py -m gpu -q -x > $OUT/r03k_pytest_breed_unit4.log 2>&1; tail -3 $OUT/r03k_pytest_breed_unit4.log | cut -c1-250
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "north_star or hints or full_size" > $OUT/r03k_pytest.log 2>&1; tail -4 $OUT/r03k_pytest.log | cut -c1-250
