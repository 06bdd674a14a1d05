R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
for m in "fused" "pipe 2" "pipe 4" "pipe 8"; do
  set -- $m
  EVOGP_BREED_COMPILE_MODE=$1 EVOGP_BREED_COMPILE_PIECES=${2:-4} timeout 600 python scripts/dbg/gen_step_parts.py 250000 1000000 2>&1 | grep -v amdgpu.ids | grep "breed + compile" | sed "s/^/mode $m: /"
done > $OUT/r03g_breed_compile_modes.log 2>&1
cat $OUT/r03g_breed_compile_modes.log | cut -c1-260
EVOGP_BREED_COMPILE_MODE=pipe timeout 900 python -m pytest tests/test_gpu_breed.py -m gpu -q -x -k "compiles_ahead" > $OUT/r03g_pytest_breed_pipe.log 2>&1; tail -3 $OUT/r03g_pytest_breed_pipe.log | cut -c1-250
