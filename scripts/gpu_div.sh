#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 600 python scripts/div_modes.py > $OUT/40_div_modes.log 2>&1; tail -3 $OUT/40_div_modes.log | cut -c1-700
EVOGP_SR_DIV=short timeout 900 python -m pytest tests -m gpu -q -k "sr_fitness or full_size or golden or api or gp_loop or breed" > $OUT/41_pytest_fastdiv.log 2>&1; tail -15 $OUT/41_pytest_fastdiv.log | cut -c1-300
EVOGP_SR_DIV=short timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-300
