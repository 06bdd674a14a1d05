#!/bin/bash
# Device-only compile of sr_tc.hip to assembly + the resource lines of the kernels matching $1 (default: fused|pack); the assembly of
# sr_fused_kernel<8,2> goes to /tmp/dis/fused2.s.   bash scripts/dev_isa.sh [pattern]
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/dis
make -C $R/evogp_amd/csrc tc_interp_k8.inc > /dev/null 2>&1
(cd $R/evogp_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-inline-asm \
   -mllvm -amdgpu-atomic-optimizer-strategy=None $EXTRA_HIPFLAGS --cuda-device-only -S sr_tc.hip -o /tmp/dis/sr_tc.s 2>&1 | grep -v "hip-link" | head -20)
awk '/^  - \.agpr_count|\.name: |\.private_segment_fixed_size|\.sgpr_count|\.sgpr_spill_count|\.vgpr_count|\.vgpr_spill_count|\.agpr_count/' /tmp/dis/sr_tc.s \
  | paste - - - - - - - | grep -i "${1:-fused\|pack}" | sed 's/  */ /g; s/\.agpr_count: 0//; s/- //'
awk '/^_ZN5evogp15sr_fused_kernelILi8ELi2EEEvNS_11FusedParamsE:/{p=1} p{print} /^\.Lfunc_end[0-9]*:/{if(p)exit}' /tmp/dis/sr_tc.s > /tmp/dis/fused2.s
echo "fused2.s: $(wc -l < /tmp/dis/fused2.s) lines, scratch ops: $(grep -c scratch_ /tmp/dis/fused2.s), asm copies: $(grep -c 'Ltc_exit_here_[0-9]*:' /tmp/dis/fused2.s)"
