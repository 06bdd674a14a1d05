#!/bin/bash
# Variant libraries of the fused fitness kernel (scripts/build_variant.sh): every evogp_amd/lib/libevogp_hip_<v>.so whose name does not
# end in "stats" is timed on the headline call and compared word for word with the default build (scripts/gpu_div_ab.sh); the
# *stats builds print where the waves' clocks go (scripts/fused_cycles.py).   gpurun -- 'bash scripts/gpu_fused_variants.sh TAG'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r04b}
mkdir -p $OUT /tmp/statslibs
cd $R
mv evogp_amd/lib/libevogp_hip_*stats.so /tmp/statslibs/ 2>/dev/null
AB_ENV="twokernel:EVOGP_TC_FUSED=0,EVOGP_TC_PACK=0 $AB_ENV" bash scripts/gpu_div_ab.sh $TAG | grep "headline call\|IDENTICAL\|DIFFERENT\|cmp new"
cp evogp_amd/lib/libevogp_hip.so /tmp/keep.so
for s in /tmp/statslibs/*.so; do
  [ -f "$s" ] || continue
  cp $s evogp_amd/lib/libevogp_hip.so
  echo "== $(basename $s)"
  POP=1000000 timeout 300 python scripts/fused_cycles.py 2>&1 | grep -v amdgpu.ids
  POP=125000 timeout 300 python scripts/fused_cycles.py 2>&1 | grep -v amdgpu.ids
  cp /tmp/keep.so evogp_amd/lib/libevogp_hip.so
done 2>&1 | tee $OUT/${TAG}_fused_cycles.log
if [ "$2" == "tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1
  tail -15 $OUT/${TAG}_pytest_gpu.log
fi
