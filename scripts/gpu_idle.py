#!/usr/bin/env python3
"""How busy the GPU is between the first and the last kernel of the LAST rollout step in a rocprofv3 kernel trace (rocpd
SQLite): sum of kernel durations vs the span, and the idle gaps attributed to the kernel that FOLLOWS each gap (the one
whose launch arrived late).  Usage: gpu_idle.py DB [n_last_kernels]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("stream_id", "queue_id") if c in cols), None)
rows = db.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
# One ROLLOUT (denoise) step = one transformer forward + the fused CFG / SDE step, on the launch stream.  The window is the last NINE
# denoise steps of the last rollout: from the end of the 10th-from-last sde_finalize_kernel to the end of the last one, launch-stream
# kernels only (the stream the SDE kernels run on).  (Until round 5 this script windowed on the last two group_advantage kernels; since
# scoring became a reward future that window held the PickScore tail of the previous group, not a rollout step: VERDICT r5 weak 10.)
marks = [i for i, r in enumerate(rows) if "sde_finalize_kernel" in r[0]]
if len(marks) >= 10:
    q = rows[marks[-1]][3] if qcol else None
    lo, hi = marks[-10] + 1, marks[-1] + 1
    win = [r[:3] for r in rows[lo:hi] if qcol is None or r[3] == q]
    n_steps = 9
else:
    win, n_steps = [r[:3] for r in rows], 1
print(f"window: the last {n_steps} denoise steps of the last rollout (between sde_finalize_kernel launches), launch stream "
      f"{'(' + qcol + ' = ' + str(q) + ')' if len(marks) >= 10 and qcol else '(all streams)'}")
span = win[-1][2] - win[0][1]
busy = sum(e - s for _, s, e in win)
print(f"kernels in the window: {len(win)} ({len(win) / n_steps:.0f} per denoise step); span {span / 1e6:.2f} ms; sum of kernel durations {busy / 1e6:.2f} ms; idle {100 * (1 - busy / span):.2f} %")
gaps = {}
prev_end = win[0][2]
for n, s, e in win[1:]:
    g = max(0, s - prev_end)
    a = gaps.setdefault(n[:70], [0, 0])
    a[0] += g; a[1] += 1
    prev_end = max(prev_end, e)
for n, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  gap before {n:70s} {g / 1e6:8.3f} ms over {c:5d} launches ({g / c / 1e3:6.2f} us each)")

tot = {}
for n, s_, e in win:
    a = tot.setdefault(n[:90], [0, 0])
    a[0] += e - s_; a[1] += 1
print("kernel time inside the step:")
for n, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"  {t / 1e6:8.3f} ms {100 * t / span:5.1f} %  {c:5d} x {t / c / 1e3:8.1f} us  {n}")
