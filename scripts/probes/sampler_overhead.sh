#!/bin/bash
# Does the rocm-smi sampler thread of bench.py perturb the timed steps?  (ADVGRPO_BENCH_NO_SMI=1 disables it)
for v in 0 1 0 1; do
  ADVGRPO_BENCH_NO_SMI=$v python bench.py --no-epoch --no-cpu-baseline --no-pricing --steps 6 2>/dev/null > /tmp/sm_$v.json
  python -c "
import json; r=json.load(open('/tmp/sm_$v.json')); print('no_smi', $v, r['value'], 'img/s', r['ms_per_step'], 'ms', r['clock_and_power'] and r['clock_and_power']['sclk_mhz_median'])"
done
