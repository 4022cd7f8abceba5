#!/usr/bin/env python3
"""Effective shader clock per kernel = GRBM_GUI_ACTIVE / kernel duration (MI355X_MICROARCH.md "DVFS give-back") from a rocprofv3
--pmc GRBM_GUI_ACTIVE database.  Usage: pmc_clock.py DB [OUT.json]"""
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cnt = {k: (v, n) for k, v, n in db.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = 'GRBM_GUI_ACTIVE' group by kernel_name")}
dur = {k: (d, n) for k, d, n in db.execute("select name, avg(duration), count(*) from kernels group by name")}
out = {}
for k, (c, n) in cnt.items():
    if k in dur and dur[k][0] > 20e3:        # kernels longer than 20 us
        out[k[:90]] = {"launches": n, "avg_us": round(dur[k][0] / 1e3, 1), "gui_active_cycles": round(c), "effective_clock_ghz": round(c / dur[k][0], 3)}
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"])[:16]:
    print(f"{k[:70]:70s} n={v['launches']:5d} {v['avg_us']:9.1f} us  clock {v['effective_clock_ghz']:.3f} GHz")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
