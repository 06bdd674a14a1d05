#!/bin/bash
# The round's standard GPU pass: the GPU test suite, smoke(), the bench line (all extras).   gpurun -- 'bash scripts/gpu_round.sh TAG'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-round}
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/${TAG}_pytest_gpu.log 2>&1
tail -22 $OUT/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step") if k in d})
for k in ("uci_sr_shape", "c5_rollout", "configs3", "shard_model"):
    print(k, json.dumps(d.get(k))[:1500])
PY
