#!/bin/bash
# round 6, ninth GPU job: rollout streams measured against the scoring stream as well -- the default line three times on one box
set -x
R=$PWD
O=$R/gpurun_out/r6_job9
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
for i in 1 2 3; do timeout 600 python $R/bench.py --steps 12 --warmup 3 --no-epoch --no-pricing --no-cpu-baseline > $O/bench_$i.json 2>$O/bench_$i.err; done
timeout 600 python $R/bench.py --steps 12 --warmup 3 --no-epoch --no-pricing --no-cpu-baseline --sync-scoring > $O/bench_sync_scoring.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], 'serial', d['serial']['ms_per_step'])
PY
grep -h -i 'warn' $O/*.err | head
