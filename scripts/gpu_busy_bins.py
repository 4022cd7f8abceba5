#!/usr/bin/env python3
"""GPU busy share over time from a rocprofv3 kernel trace (rocpd SQLite): the union of all kernel intervals (any stream) per time bin.  With several
streams in flight a per-stream idle figure (gpu_idle.py) says nothing; this one says whether the device ever waits for the host.
Usage: gpu_busy_bins.py DB [bin_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
bin_ns = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 50_000_000
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("stream_id", "queue_id") if c in cols), None)
rows = db.execute(f"select start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
nb = (t1 - t0) // bin_ns + 1
busy = [0] * nb
streams = [set() for _ in range(nb)]
cur_s, cur_e = rows[0][0], rows[0][1]


def add(s, e):
    b = (s - t0) // bin_ns
    while s < e:
        lim = t0 + (b + 1) * bin_ns
        busy[b] += min(e, lim) - s
        s = min(e, lim)
        b += 1


for r in rows[1:]:
    s, e = r[0], r[1]
    if s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        add(cur_s, cur_e)
        cur_s, cur_e = s, e
add(cur_s, cur_e)
if qcol:
    for r in rows:
        streams[(r[0] - t0) // bin_ns].add(r[2])
print(f"{len(rows)} kernels over {(t1 - t0) / 1e9:.2f} s; bins of {bin_ns / 1e6:.0f} ms: busy share of the device (union over streams) | streams with launches")
for i in range(nb):
    print(f"  {i * bin_ns / 1e9:7.2f} s  {100 * busy[i] / bin_ns:6.2f} %  {len(streams[i]) if qcol else ''}")
