#!/bin/bash
# An A/B library of the engine: the same sources with the interpreter generated under other switches.
#   scripts/build_variant.sh <suffix> VAR=value [VAR=value ...]   -> evogp_amd/lib/libevogp_hip_<suffix>.so
# (only sr_tc.hip is recompiled; the other objects come from build/obj, so run `make -C evogp_amd/csrc` first)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
suffix=$1; shift
D=$R/build/variant_$suffix
rm -rf $D; mkdir -p $D/src $D/lib; ln -sfn $R/include $R/build/include
cp $R/evogp_amd/csrc/*.hip $R/evogp_amd/csrc/*.hpp $D/src/
(cd $D/src && env "$@" python3 $R/evogp_amd/csrc/gen/gen_tc_asm.py . > /dev/null)
(cd $D/src && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-inline-asm -mllvm -amdgpu-atomic-optimizer-strategy=None $EXTRA_HIPFLAGS -I$R/include -c sr_tc.hip -o sr_tc.o)
objs=$(ls $R/build/obj/*.o | grep -v /sr_tc.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/evogp_amd/lib/libevogp_hip_$suffix.so $objs $D/src/sr_tc.o
echo "built evogp_amd/lib/libevogp_hip_$suffix.so ($*)"
