#!/bin/bash
# the G-step's adapter-gradient side stream = a rollout stream: the line with its epoch leg twice, trainer / train tests
set -x
R=$PWD
O=$R/gpurun_out/r6_job12
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
for i in 1 2; do timeout 600 python $R/bench.py --steps 12 --warmup 3 --no-pricing --no-cpu-baseline > $O/bench_with_epoch_$i.json 2>/dev/null; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); e=d.get('epoch') or {}
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'serial', d['serial']['ms_per_step'], e.get('images_per_s_full_epoch'), e.get('phases_s'), e.get('g_step_inside',{}).get('micro_step'), d['clock_and_power']['sclk_mhz_median'])
PY
timeout 1200 python -m pytest $R/tests/test_gpu_trainer.py $R/tests/test_gpu_train.py -m gpu -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
