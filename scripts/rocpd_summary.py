#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (SQLite) result into a per-kernel stats table (like --stats CSV)
and, when PMC counters were collected, per-kernel counter sums.  Usage:
    python scripts/rocpd_summary.py <results.db> [out.md]
ROCPD_BY_GRID=1: launches of one kernel with different grids are separate rows (name + " [grid N]"): scripts/ops_pmc.py tells the
measured launches of an operator from the set-up launches of the same kernel that way."""
import os
import sqlite3
import sys

BY_GRID = os.environ.get("ROCPD_BY_GRID") == "1"


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    if BY_GRID:
        name_col = f"({name_col} || ' [grid ' || grid_x || ' lds ' || lds_size || ']')"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                       "max(workgroup_x), max(grid_x) from kernels group by 1 order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = ["| kernel | calls | total ms | avg us | min us | max us | % | VGPR | AGPR | SGPR | LDS B | scratch B | wg | grid |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        n = r[0] if len(r[0]) < 90 else r[0][:60] + "..." + r[0][-27:]
        out.append(f"| `{n}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | {r[5]/1e3:.2f} | {100*r[2]/total:.1f} | "
                   f"{r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} | {r[12]} |")
    try:
        if BY_GRID:
            pm = cur.execute("select (p.name || ' [grid ' || k.grid_x || ' lds ' || k.lds_size || ']'), p.counter_name, count(*), sum(p.counter_value) from pmc_events p "
                             "join kernels k on k.dispatch_id = p.dispatch_id and k.guid = p.guid group by 1,2 order by 1,2").fetchall()
        else:
            pm = cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by 1,2 order by 1,2").fetchall()
    except Exception as e:  # layout differs between versions: report, do not fail
        pm = []
        try:
            cols2 = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
            out.append(f"\n(pmc_events columns: {cols2}; query failed: {e})")
        except Exception:
            pass
    if pm:
        out += ["", "| kernel | counter | dispatches | sum | per dispatch |", "|---|---|---|---|---|"]
        for k, c, n, v in pm:
            kk = k if len(k) < 70 else k[:45] + "..." + k[-22:]
            out.append(f"| `{kk}` | {c} | {n} | {v:.6g} | {v/max(n,1):.6g} |")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
