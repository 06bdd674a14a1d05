R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
EVOGP_BREED_COMPILE=1 EVOGP_TC_HINTS=1 timeout 1800 python -m pytest tests -m gpu -q > $OUT/r03q_pytest_experiments_on.log 2>&1; tail -6 $OUT/r03q_pytest_experiments_on.log | cut -c1-250
EVOGP_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --global-pop 40000 --pop-per-gpu 20000 > $OUT/r03q_two_ranks.log 2>&1; tail -1 $OUT/r03q_two_ranks.log | cut -c1-300
