#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (rocpd sqlite, the default output of `rocprofv3 --kernel-trace --stats`) into the
per-kernel stats summary committed under profiles/ (name, calls, total ms, average ms, min, max, %)."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                     "from kernels group by name, grid_x order by sum(duration) desc").fetchall()  # per (kernel, grid): the round launch of k_control (1 + helpers workgroups) apart from its 1-workgroup command launches
    tot = sum(r[2] for r in rows) or 1
    lines = ["# rocprofv3 --kernel-trace --stats summary (durations in ms)",
             f"# source: {db}", "name,calls,total_ms,avg_ms,min_ms,max_ms,pct,vgpr,sgpr,lds_bytes,grid_x,wg_x"]
    for r in rows:
        lines.append(f"\"{r[0]}\",{r[1]},{r[2]/1e6:.4f},{r[3]/1e6:.4f},{r[4]/1e6:.4f},{r[5]/1e6:.4f},{100*r[2]/tot:.2f},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
