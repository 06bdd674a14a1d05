#!/bin/bash
# The one-kernel fitness call (sr_fused_kernel) against the two-kernel path: fitness words of the A/B forests bit for bit, the
# headline call's time, then the GPU test suite and the bench line.   gpurun -- 'bash scripts/gpu_fused_ab.sh TAG [quick]'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r04a}
mkdir -p $OUT
cd $R
AB_ENV="twokernel:EVOGP_TC_FUSED=0 $AB_ENV" bash scripts/gpu_div_ab.sh $TAG
if [ "$2" != "quick" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1
  tail -15 $OUT/${TAG}_pytest_gpu.log
  timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
  python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "roofline") if k in d})
PY
fi
