#!/bin/bash
# One gpurun call: smoke, GPU parity tests, bench, variant sweep, rocprofv3 kernel trace.
# Usage (from the dev container):  gpurun --timeout 2400 -- 'bash scripts/gpu_round.sh [quick]'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
MODE=${1:-full}
{
echo "== device"; rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -8; nproc
python -c "import torch;print('torch', torch.__version__, 'gpus', torch.cuda.device_count())"
echo "== smoke"; timeout 900 python __graft_entry__.py smoke; echo "smoke rc=$?"
} > $OUT/00_smoke.log 2>&1
# threaded-code core: first contact under a short timeout; fall back to the C++ interpreter for the rest of the run if it misbehaves
for DEP in 10 16; do timeout 150 python scripts/asm_smoke.py $DEP > $OUT/00_asm_smoke_$DEP.log 2>&1; echo "asm_smoke $DEP rc=$?" >> $OUT/00_asm_smoke_$DEP.log; done
if ! grep -q ASM_SMOKE_OK $OUT/00_asm_smoke_10.log; then export EVOGP_SR_ASM=0; echo "ASM DISABLED" >> $OUT/00_asm_smoke_10.log; fi
if [ "$MODE" != "benchonly" ]; then
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/01_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/01_pytest_gpu.log
fi
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/02_bench.log 2>&1; echo "bench rc=$?" >> $OUT/02_bench.log
{ for A in 0 10 16; do echo "== ASM=$A"; EVOGP_SR_ASM=$A timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline; done; } > $OUT/02b_asm_ab.log 2>&1
{ for A in 10 16; do EVOGP_SR_ASM=$A timeout 200 python scripts/asm_cycles.py; done; } > $OUT/05_cycles.log 2>&1
if [ "$MODE" = "quick" ]; then ls -la $OUT; exit 0; fi
{
for K in 1 2 4; do for DEP in 16 32; do
  echo "== K=$K DEPTH=$DEP"; EVOGP_SR_K=$K EVOGP_SR_DEPTH=$DEP timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline
done; done
echo "== structured build (no skip-uniform-regions), default K/DEPTH"
EVOGP_HIP_LIB=$R/evogp_amd/lib/libevogp_hip_structured.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline
for B in 4 8 16 32 64; do echo "== batch=$B"; EVOGP_SR_BATCH=$B timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline; done
} > $OUT/03_sweep.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r01 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/04_rocprof.log 2>&1
echo "rocprof rc=$?" >> $OUT/04_rocprof.log
find $OUT/prof -name "*stats*" | head >> $OUT/04_rocprof.log
ls -la $OUT
