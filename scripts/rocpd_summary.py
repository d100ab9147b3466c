#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel calls / total / average duration (us).
usage: rocpd_summary.py results.db > profiles/<name>.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("kernel,calls,total_us,avg_us,percent")
for name, calls, tot, avg, pct in rows:
    name = name.replace("(anonymous namespace)::", "").replace('"', "'")
    if len(name) > 110:
        name = name[:107] + "..."
    print(f'"{name}",{calls},{tot:.1f},{avg:.2f},{pct:.2f}')
