#!/bin/bash
# quick check of a change set: targeted GPU tests given as a -k expression ($1), operator table, short bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "$1" > $OUT/30_pytest_k.log 2>&1; tail -3 $OUT/30_pytest_k.log
timeout 300 python scripts/bench_ops.py > $OUT/31_ops.md 2>&1; grep "^|" $OUT/31_ops.md | cut -c1-160
EVOGP_GEN_STAGED=0 timeout 300 python scripts/bench_ops.py 2>&1 | grep "tree_generate" | cut -c1-160
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/32_bench.json 2> $OUT/32_bench.err; python3 - <<PY
import json
d=json.loads(open("$OUT/32_bench.json").read().strip().splitlines()[-1])
print("value %.3e launch_ms %.4f gen %s" % (d["value"], d["roofline"]["launch_ms"], d["generation_ms"]))
PY
