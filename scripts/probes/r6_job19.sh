#!/bin/bash
# more schedule knobs on one box, two alternations: decode inside the reward future, 4 hardware queues
set -x
R=$PWD
O=$R/gpurun_out/r6_job19; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
F="--steps 12 --warmup 3 --no-epoch --no-pricing --no-cpu-baseline"
for i in 1 2; do
  timeout 600 python $R/bench.py $F > $O/base_$i.json 2>/dev/null
  timeout 600 python $R/bench.py $F --decode-in-future 1 > $O/decode_in_future_$i.json 2>/dev/null
  GPU_MAX_HW_QUEUES=4 timeout 600 python $R/bench.py $F > $O/hwq4_$i.json 2>/dev/null
  GPU_MAX_HW_QUEUES=16 timeout 600 python $R/bench.py $F > $O/hwq16_$i.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], 'serial', d['serial']['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
