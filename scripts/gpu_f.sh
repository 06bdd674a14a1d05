R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
for w in 5 6 8; do
  EVOGP_BREED_COMPILE_WAVES=$w timeout 600 python scripts/dbg/gen_step_parts.py 125000 1000000 2>&1 | grep -v amdgpu.ids | grep "breed + compile" | sed "s/^/waves $w: /"
done > $OUT/r03f_breed_compile_waves.log 2>&1
cat $OUT/r03f_breed_compile_waves.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_breed.py -m gpu -q -x > $OUT/r03f_pytest_breed.log 2>&1; tail -4 $OUT/r03f_pytest_breed.log | cut -c1-250
