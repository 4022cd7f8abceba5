#!/usr/bin/env python3
"""Mean of each PMC counter per kernel from a rocprofv3 rocpd database.  Usage: pmc_db.py DB [kernel-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = f"%{sys.argv[2]}%" if len(sys.argv) > 2 else "%"
for kn, cn, v, n in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name "
                               "like ? group by kernel_name, counter_name", (pat,)):
    print(f"{kn[:50]:50s} {cn:28s} {v:14.4e} n={n}")
for kn, d, n in db.execute("select name, avg(duration), count(*) from kernels where name like ? group by name", (pat,)):
    print(f"{kn[:50]:50s} avg duration {d / 1e3:.1f} us n={n}")
