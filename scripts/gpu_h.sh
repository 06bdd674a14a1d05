R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
EVOGP_DEBUG_CLS=1 timeout 600 python scripts/c3_profile.py > $OUT/r03h_c3_digits.log 2>&1; grep -E "generation|marked" $OUT/r03h_c3_digits.log | head -12
EVOGP_DEBUG_CLS=1 timeout 600 python scripts/c3_profile.py synthetic > $OUT/r03h_c3_synth.log 2>&1; grep -E "generation|marked" $OUT/r03h_c3_synth.log | head -6
timeout 900 python -m pytest tests/test_gpu_breed.py -m gpu -q -x -k "compiles_ahead" > $OUT/r03h_pytest_breed.log 2>&1; tail -3 $OUT/r03h_pytest_breed.log | cut -c1-250
