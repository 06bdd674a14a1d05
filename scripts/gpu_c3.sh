cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
python $R/scripts/c3_profile.py > $OUT/r03d_c3_digits.log 2>&1
python $R/scripts/c3_profile.py synthetic > $OUT/r03d_c3_synth.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_c3 -o tr -- python $R/scripts/c3_profile.py > $OUT/r03d_c3_rocprof.log 2>&1
python $R/scripts/rocpd_summary.py $(find $OUT/prof_c3 -name "*.db" | head -1) > $OUT/r03d_c3_kernel_stats.md 2>&1
rm -rf $OUT/prof_c3
cd $R; timeout 600 python -m pytest "tests/test_gpu_parity.py" -k argmax_count -m gpu -q > $OUT/r03d_pytest.log 2>&1
tail -3 $OUT/r03d_pytest.log; cat $OUT/r03d_c3_digits.log $OUT/r03d_c3_synth.log; head -14 $OUT/r03d_c3_kernel_stats.md | cut -c1-200
