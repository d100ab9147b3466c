#!/usr/bin/env python3
"""Timeline of the last bench step from a rocprofv3 rocpd database: kernel start/end and the idle gap before each.
usage: timeline.py results.db [n_last_kernels]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t]
view = "kernels" if "kernels" in tabs else None
if view:
    cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
    rows = db.execute(f"select name, start, end from {view} order by start").fetchall()
else:
    print(tabs); sys.exit(1)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-n:]
t0 = rows[0][1]
prev_end = None
busy = 0
for name, st, en in rows:
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    name = name.replace("(anonymous namespace)::", "").split("(")[0][-60:]
    print(f"{(st - t0) / 1e3:9.1f} us  +{(en - st) / 1e3:8.1f}  gap {gap:7.1f}  {name}")
    busy += en - st
    prev_end = max(prev_end or en, en)
print(f"span {(rows[-1][2] - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
