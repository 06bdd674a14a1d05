#!/bin/bash
# A GPU session in steps: usage  gpurun --timeout N -- 'bash scripts/gpu_session.sh TAG step [step ...]'
# steps: smoke tests bench prof pmc calib extra:<script.py> ...   Everything lands in gpurun_out/TAG_*.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-session}; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for step in "$@"; do
case $step in
smoke)
  { rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -6; nproc
    python -c "import torch;print('torch', torch.__version__, 'gpus', torch.cuda.device_count())"
    timeout 900 python __graft_entry__.py smoke; echo "smoke rc=$?"; } > $OUT/${TAG}_00_smoke.log 2>&1
  tail -2 $OUT/${TAG}_00_smoke.log ;;
tests)
  timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_01_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_01_pytest_gpu.log
  tail -15 $OUT/${TAG}_01_pytest_gpu.log ;;
testsall)   # do not stop at the first failure
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/${TAG}_01_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_01_pytest_gpu.log
  tail -40 $OUT/${TAG}_01_pytest_gpu.log ;;
bench)
  timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_02_bench.json 2> $OUT/${TAG}_02_bench.err; echo "bench rc=$?" >> $OUT/${TAG}_02_bench.err
  tail -3 $OUT/${TAG}_02_bench.err; cut -c1-1500 $OUT/${TAG}_02_bench.json ;;
bench2)     # the N > 1 code path on this 1-GPU box: two ranks share the GPU over gloo (functional check, never a measurement)
  EVOGP_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --global-pop 40000 --pop-per-gpu 20000 > $OUT/${TAG}_03_two_ranks_shared_gpu.log 2>&1
  tail -2 $OUT/${TAG}_03_two_ranks_shared_gpu.log | cut -c1-600 ;;
prof)
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o tr -- python $R/bench.py --steps 20 --warmup 3 --headline-only > $OUT/${TAG}_04_rocprof.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/prof_$TAG -name "*.db" | head -1) > $OUT/${TAG}_04_kernel_stats.md 2>&1
  rm -rf $OUT/prof_$TAG; cd $R
  tail -1 $OUT/${TAG}_04_rocprof.log | cut -c1-400; head -8 $OUT/${TAG}_04_kernel_stats.md | cut -c1-170 ;;
pmc)
  cd /tmp
  i=0
  for set in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$TAG$i -o pmc -- python $R/bench.py --steps 4 --warmup 1 --headline-only > $OUT/${TAG}_05_pmc$i.log 2>&1
    python $R/scripts/rocpd_summary.py $(find $OUT/pmc_$TAG$i -name "*.db" | head -1) 2>&1 | grep -A400 "counter" | grep -i "counter\|---\|evogp" > $OUT/${TAG}_05_pmc$i.md
    rm -rf $OUT/pmc_$TAG$i
  done
  cd $R
  python scripts/pmc_json.py $OUT/${TAG}_05_pmc1.md $OUT/${TAG}_05_pmc2.md $OUT/${TAG}_05_pmc3.md $OUT/${TAG}_05_pmc1.log > $OUT/${TAG}_05_pmc_latest.json 2> $OUT/${TAG}_05_pmc_json.err
  cat $OUT/${TAG}_05_pmc3.md | cut -c1-200 ;;
calib)      # known-byte-count reads: what does FETCH_SIZE report for 2-, 4- and 16-byte-per-lane loads?
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/calib_$TAG -o pmc -- python $R/scripts/pmc_calibrate.py > $OUT/${TAG}_06_calib.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/calib_$TAG -name "*.db" | head -1) 2>&1 | grep -i "counter\|---\|calib_read" > $OUT/${TAG}_06_calib.md
  rm -rf $OUT/calib_$TAG; cd $R
  cat $OUT/${TAG}_06_calib.md | cut -c1-200 ;;
opspmc)     # the HBM-bound operators under FETCH_SIZE / WRITE_SIZE (scripts/ops_pmc.py) -> ${TAG}_07_ops_pmc.md
  cd /tmp
  export OPS_JSON=$OUT/${TAG}_07_ops.json ROCPD_BY_GRID=1
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/opspmc_$TAG$set -o pmc -- python $R/scripts/ops_pmc.py run > $OUT/${TAG}_07_ops_$set.log 2>&1
    python $R/scripts/rocpd_summary.py $(find $OUT/opspmc_$TAG$set -name "*.db" | head -1) > $OUT/${TAG}_07_ops_$set.md 2>&1
    rm -rf $OUT/opspmc_$TAG$set
  done
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/opspmc_${TAG}st -o tr -- python $R/scripts/ops_pmc.py run > $OUT/${TAG}_07_ops_stats.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/opspmc_${TAG}st -name "*.db" | head -1) > $OUT/${TAG}_07_ops_stats.md 2>&1
  rm -rf $OUT/opspmc_${TAG}st
  unset ROCPD_BY_GRID
  cd $R
  python scripts/ops_pmc.py table $OUT/${TAG}_07_ops_FETCH_SIZE.md $OUT/${TAG}_07_ops_WRITE_SIZE.md $OUT/${TAG}_07_ops_stats.md $OPS_JSON > $OUT/${TAG}_07_ops_pmc.md 2>&1
  cat $OUT/${TAG}_07_ops_pmc.md | cut -c1-260 ;;
pytest:*)    # pytest:<file or node id> [-k expr]
  s=${step#pytest:}; n=$(basename ${s%% *} .py)
  timeout 1800 python -m pytest $s -m gpu -q -x > $OUT/${TAG}_01_pytest_$n.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_01_pytest_$n.log
  tail -12 $OUT/${TAG}_01_pytest_$n.log | cut -c1-300 ;;
extra:*)
  s=${step#extra:}; n=$(basename ${s%% *} .py)
  timeout 900 python $s > $OUT/${TAG}_10_$n.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_10_$n.log; tail -25 $OUT/${TAG}_10_$n.log | cut -c1-300 ;;
esac
done
