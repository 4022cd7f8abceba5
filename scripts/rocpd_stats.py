#!/usr/bin/env python3
"""Kernel-stats summary (name, calls, total ms, avg us, %) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats` writes <name>_results.db on this ROCm).  Usage: rocpd_stats.py DB [OUT.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                  "group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, c, t, a, mn, mx in rows:
    lines.append(f"| `{n[:110]}` | {c} | {t / 1e6:.2f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * t / tot:.2f} |")
out = "\n".join(lines) + f"\n\ntotal kernel time {tot / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "a").write(out)
print(out)
