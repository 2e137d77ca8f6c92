#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of `rocprofv3 --pmc <counter> --kernel-trace` (rocpd sqlite) into the JSON
committed under profiles/ and read by bench.py for `roofline.traffic`.

usage: pmc_summary.py <fetch.db> <write.db> <out.json>
Values are KB as rocprofv3 reports them; bench.py applies the gfx950 correction (FETCH_SIZE x2, MI355X_MICROARCH.md HBM section)."""
import json
import sqlite3
import sys


def main(fetch_db, write_db, out_path):
    out = {}
    for name, f in (("FETCH_SIZE", fetch_db), ("WRITE_SIZE", write_db)):
        c = sqlite3.connect(f)
        rows = c.execute("select kernel_name, counter_name, count(*), sum(value), max(value) from counters_collection "
                         "group by kernel_name, counter_name order by sum(value) desc").fetchall()
        out[name] = [dict(kernel=r[0][:60], counter=r[1], dispatches=r[2], sum_kb=r[3], max_kb=r[4]) for r in rows[:6]]
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 1 --warmup 0 --cpu-budget 0` at BASELINE "
            "configs[2]; values in KB as reported; FETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md HBM section)")
    import os, re, subprocess
    try:   # the round kernel these counters were taken from (bench.py prints the same hash of the library it times)
        txt = subprocess.run(["bash", os.path.join(os.path.dirname(os.path.abspath(__file__)), "kcontrol_isa_hash.sh")], capture_output=True, text=True, timeout=120).stdout
        isa = (re.search(r"sha256 ([0-9a-f]{16})", txt) or [None, None])[1]
    except Exception:
        isa = None
    json.dump({"note": note, "kernel_isa_hash": isa, "counters": out}, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
