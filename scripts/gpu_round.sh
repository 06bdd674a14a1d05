#!/bin/bash
# One gpurun call that produces everything a round needs: smoke(), first-contact checks of the threaded-code path, the GPU
# test suite, the bench line (with CPU baseline), A/B lines, a rocprofv3 kernel trace summary, cycle accounting and
# HBM-traffic counters.  Usage (from the dev container):  gpurun --timeout 2400 -- 'bash scripts/gpu_round.sh [tag]'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-round}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
{
echo "== device"; rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -8; nproc
python -c "import torch;print('torch', torch.__version__, 'gpus', torch.cuda.device_count())"
echo "== smoke"; timeout 900 python __graft_entry__.py smoke; echo "smoke rc=$?"
} > $OUT/${TAG}_00_smoke.log 2>&1
for K in 8 4; do timeout 240 python tests/tools/tc_smoke.py $K > $OUT/${TAG}_01_tc_smoke_$K.log 2>&1; echo "tc_smoke $K rc=$?" >> $OUT/${TAG}_01_tc_smoke_$K.log; done
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_02_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_02_pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_03_bench.json 2> $OUT/${TAG}_03_bench.err; echo "bench rc=$?" >> $OUT/${TAG}_03_bench.err
{ for A in "EVOGP_SR_ASM=3" "EVOGP_SR_ASM=0" "EVOGP_NATIVE_STEP=0"; do echo "== $A"; env $A timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline; done; } > $OUT/${TAG}_04_ab.log 2>&1
# functional check of the N>1 code path of bench.py on this 1-GPU box: two ranks share the GPU over gloo (not a measurement)
EVOGP_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 5 --warmup 1 --pop-per-gpu 20000 > $OUT/${TAG}_04b_two_ranks_shared_gpu.log 2>&1
timeout 300 python scripts/bench_ops.py > $OUT/${TAG}_09_ops.md 2>&1
timeout 300 python scripts/div_modes.py > $OUT/${TAG}_10_div_modes.log 2>&1
{ timeout 100 scripts/ubench/div_faithful; timeout 100 scripts/ubench/valu_rates; } > $OUT/${TAG}_11_ubench.log 2>&1
timeout 200 python scripts/tc_cycles.py > $OUT/${TAG}_05_cycles.json 2>/dev/null
timeout 300 python tests/tools/tc_mix.py > $OUT/${TAG}_06_mix.log 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o tr -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_07_rocprof.log 2>&1
python $R/scripts/rocpd_summary.py $(find $OUT/prof_$TAG -name "*.db" | head -1) > $OUT/${TAG}_07_kernel_stats.md 2>&1
rm -rf $OUT/prof_$TAG
i=0
for set in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$TAG$i -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/pmc_$TAG$i -name "*.db" | head -1) 2>&1 | grep -A400 "counter" | grep -i "counter\|---\|sr_tc\|tc_compile\|breed\|generate" > $OUT/${TAG}_08_pmc$i.md
  rm -rf $OUT/pmc_$TAG$i
done
tail -2 $OUT/${TAG}_00_smoke.log; tail -1 $OUT/${TAG}_01_tc_smoke_8.log; tail -2 $OUT/${TAG}_02_pytest_gpu.log; cut -c1-400 $OUT/${TAG}_03_bench.json; head -6 $OUT/${TAG}_07_kernel_stats.md | cut -c1-160
