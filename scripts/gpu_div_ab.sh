#!/bin/bash
# A/B of interpreter builds: the library in evogp_amd/lib against every evogp_amd/lib/libevogp_hip_<variant>.so
# (scripts/build_variant.sh), fitness words compared bit for bit.   gpurun -- 'bash scripts/gpu_div_ab.sh TAG'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-divab}
mkdir -p $OUT
cd $R
{
  timeout 600 python ${AB_SCRIPT:-scripts/dbg/div_range_ab.py} run new
  # environment variants of the same library:  AB_ENV="name:VAR=value name2:VAR=value" bash scripts/gpu_div_ab.sh TAG
  for ev in $AB_ENV; do
    v=${ev%%:*}; kv=${ev#*:}; kv=${kv//,/ }
    env $kv timeout 600 python ${AB_SCRIPT:-scripts/dbg/div_range_ab.py} run $v
    python ${AB_SCRIPT:-scripts/dbg/div_range_ab.py} cmp new $v
    echo "cmp new $v rc=$?"
  done
  cp evogp_amd/lib/libevogp_hip.so /tmp/keep.so
  for alt in $(ls evogp_amd/lib/libevogp_hip_*.so 2>/dev/null); do
    v=$(basename $alt .so); v=${v#libevogp_hip_}
    cp $alt evogp_amd/lib/libevogp_hip.so
    timeout 600 python ${AB_SCRIPT:-scripts/dbg/div_range_ab.py} run $v
    cp /tmp/keep.so evogp_amd/lib/libevogp_hip.so
    python ${AB_SCRIPT:-scripts/dbg/div_range_ab.py} cmp new $v
    echo "cmp new $v rc=$?"
  done
} > $OUT/${TAG}_divab.log 2>&1
rm -f $OUT/divab_*.npz
grep -v "amdgpu.ids\|RuntimeWarning\|Xw = " $OUT/${TAG}_divab.log | tail -60 | cut -c1-250
