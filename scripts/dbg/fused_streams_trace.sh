#!/bin/bash
# kernel timeline (with queue ids) of scripts/dbg/fused_streams.py under EVOGP_TC_FUSED=1
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp
EVOGP_TC_FUSED=1 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_fz -o tl -- python $R/scripts/dbg/fused_streams.py ${1:-two} > $R/gpurun_out/fz.log 2>&1
grep call $R/gpurun_out/fz.log | grep -v "777s 0"
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$R/gpurun_out/prof_fz/**/*.db", recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print(cols)
q = [c for c in cols if "queue" in c or "stream" in c]
rows = db.execute("select name, start, end, %s, grid_x, lds_size from kernels order by start" % (q[0] if q else "0")).fetchall()[-40:]
t0 = rows[0][1]
for name, s, e, qq, gx, lds in rows:
    short = name.split("(")[0].replace("void ", "").replace("evogp::", "")[:48]
    print("%10.1f us dur %8.1f q %s grid %s lds %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qq, gx, lds, short))
PY
rm -rf $R/gpurun_out/prof_fz
