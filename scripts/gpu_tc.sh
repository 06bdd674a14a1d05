#!/bin/bash
# One gpurun call for the threaded-code path: first-contact smoke (both row widths) under short timeouts, then the
# GPU test suite, bench and A/B against the older paths.  Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_tc.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for K in 8 4; do timeout 240 python scripts/tc_smoke.py $K > $OUT/10_tc_smoke_$K.log 2>&1; echo "tc_smoke $K rc=$?" >> $OUT/10_tc_smoke_$K.log; done
if ! grep -q TC_SMOKE_OK $OUT/10_tc_smoke_8.log; then echo "TC K8 FAILED" >> $OUT/10_tc_smoke_8.log; ls -la $OUT; exit 0; fi
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/12_bench.log 2>&1; echo "bench rc=$?" >> $OUT/12_bench.log
{ for A in 3 10 0; do echo "== EVOGP_SR_ASM=$A"; EVOGP_SR_ASM=$A timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline; done
  for B in 1 2 4 8 16; do echo "== tc batch=$B"; EVOGP_TC_BATCH=$B timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline; done
  for WG in 256 512 1024; do echo "== tc wg=$WG"; EVOGP_TC_WG=$WG timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline; done
  echo "== tc K=4"; EVOGP_TC_K=4 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline
} > $OUT/12b_ab.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/11_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/11_pytest_gpu.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_tc -o tc -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/14_rocprof.log 2>&1
echo "rocprof rc=$?" >> $OUT/14_rocprof.log
python $R/scripts/rocpd_summary.py $(find $OUT/prof_tc -name "*.db" | head -1) > $OUT/14_kernel_stats.md 2>&1
ls -la $OUT
