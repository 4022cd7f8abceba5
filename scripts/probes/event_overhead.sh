#!/bin/bash
# What the per-launch HIP events of bench.py's roofline cost: the same run with events around every / every 7th / every 29th launch
for st in 1 7 29 1 7 29; do
  python bench.py --no-epoch --no-cpu-baseline --no-pricing --steps 4 --event-stride $st 2>/dev/null > /tmp/ev_$st.json
  python -c "
import json; r=json.load(open('/tmp/ev_$st.json')); print('stride', $st, r['value'], 'img/s', r['ms_per_step'], 'ms', 'frac', r['roofline']['frac'], 'timed launches', r['roofline']['launches_timed'], 'share', r['roofline']['share_of_step_time'])"
done
