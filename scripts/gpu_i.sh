R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 600 python scripts/c3_profile.py > $OUT/r03i_c3_digits.log 2>&1; grep -E "generation" $OUT/r03i_c3_digits.log | head -8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tc_wide.py -m gpu -q -x -k "argmax or classifier or wide_kernel_deep or malformed" > $OUT/r03i_pytest_cls.log 2>&1; tail -3 $OUT/r03i_pytest_cls.log | cut -c1-250
