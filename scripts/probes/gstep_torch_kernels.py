import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# micro-steps end with sde_step_bwd ... use the LAST two occurrences of attn_bwd_delta first-in-step? simpler: window between the last two 'sde_step_bwd' kernels
marks = [i for i, r in enumerate(rows) if "sde_step_bwd_kernel" in r[0]]
lo, hi = marks[-2] + 1, marks[-1] + 1
win = rows[lo:hi]
span = win[-1][2] - win[0][1]
print(f"kernels {len(win)} span {span/1e6:.2f} ms")
tot = {}
for n, s, e in win:
    if n.startswith("void at::") or "rocclr" in n or n.startswith("at::"):
        a = tot.setdefault(n[:150], [0, 0]); a[0] += e - s; a[1] += 1
t_all = sum(v[0] for v in tot.values())
print(f"torch-native kernels: {t_all/1e6:.3f} ms in {sum(v[1] for v in tot.values())} launches")
for n, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {t/1e6:7.3f} ms {c:4d} x {t/c/1e3:7.1f} us  {n}")
