#!/bin/bash
# quick loop for the threaded-code path: smoke (K=8), bench sweeps given as "VAR=val,VAR=val" words, kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 240 python tests/tools/tc_smoke.py 8 > $OUT/20_tc_smoke_8.log 2>&1; echo "tc_smoke 8 rc=$?" >> $OUT/20_tc_smoke_8.log
if ! grep -q TC_SMOKE_OK $OUT/20_tc_smoke_8.log; then echo "TC K8 FAILED"; tail -5 $OUT/20_tc_smoke_8.log; exit 0; fi
{ for cfg in "$@"; do echo "== $cfg"; env $(echo $cfg | tr ',' ' ') timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline; done; } > $OUT/21_sweep.log 2>&1
timeout 200 python scripts/tc_cycles.py > $OUT/22_cycles.log 2>&1; tail -1 $OUT/22_cycles.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_tc -o tc -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/24_rocprof.log 2>&1
python $R/scripts/rocpd_summary.py $(find $OUT/prof_tc -name "*.db" | head -1) > $OUT/24_kernel_stats.md 2>&1
cd $OUT && python3 - <<'PY'
import json
for line in open("21_sweep.log"):
    line=line.strip()
    if line.startswith("=="): print(line, end="  ")
    elif line.startswith("{"):
        d=json.loads(line); print("value %.3e  launch_ms %.4f gen_ms %.3f" % (d["value"], d["roofline"]["launch_ms"], d["generation_ms"]["median"]))
PY
head -8 $OUT/24_kernel_stats.md | cut -c1-200
