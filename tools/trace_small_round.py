"""the launch timeline of ONE small steady-state round from a rocprofv3 --kernel-trace .db: every kernel of the last schedule_round with start offset, duration and the gap before it
   usage: python tools/trace_small_round.py <results.db> [n_kernels_per_round_guess]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
# the last round = the kernels after the last k_control launch's predecessor run: take the trailing 60 launches
tail = rows[-70:]
t0 = tail[0][1]
prev = None
for name, s, e, g in tail:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:7.1f} gap  {(e - s) / 1e3:8.1f} us  grid {g:6d}  {name[:60]}")
    prev = e
