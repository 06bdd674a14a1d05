#!/usr/bin/env python3
"""Print the kernel timeline of a rocprofv3 rocpd result: start offset, duration and the gap to the previous kernel's end,
for the last N dispatches.  Usage: python scripts/rocpd_timeline.py <results.db> [N]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()[-n:]
t0 = rows[0][1]
prev_end = None
for name, s, e in rows:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    short = name.split("(")[0].replace("void ", "").replace("evogp::", "")[:60]
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {short}")
    prev_end = e
