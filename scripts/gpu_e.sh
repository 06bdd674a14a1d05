R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_breed.py -m gpu -q -x -s -k "compiles_ahead or sharded_native or genetic_programming" > $OUT/r03e_pytest_breed.log 2>&1; tail -8 $OUT/r03e_pytest_breed.log | cut -c1-250
timeout 600 python scripts/dbg/gen_step_parts.py 100000 125000 1000000 > $OUT/r03e_gen_step_parts.log 2>&1; cat $OUT/r03e_gen_step_parts.log | grep -v amdgpu.ids
timeout 600 python scripts/c3_profile.py > $OUT/r03e_c3_digits.log 2>&1; grep generation $OUT/r03e_c3_digits.log | head -3
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "argmax_count" > $OUT/r03e_pytest_cls.log 2>&1; tail -2 $OUT/r03e_pytest_cls.log
