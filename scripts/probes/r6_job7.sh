#!/bin/bash
# round 6, seventh GPU job: schedule knobs on top of (two groups in flight, one decode stream per group): three groups, scoring on the rollout stream
set -x
R=$PWD
O=$R/gpurun_out/r6_job7
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
F="--steps 12 --warmup 3 --no-epoch --no-pricing --no-cpu-baseline"
for i in 1 2; do
  timeout 600 python $R/bench.py $F > $O/g2_$i.json 2>/dev/null
  timeout 600 python $R/bench.py $F --groups-in-flight 3 > $O/g3_$i.json 2>/dev/null
  timeout 600 python $R/bench.py $F --sync-scoring > $O/g2_sync_scoring_$i.json 2>/dev/null
done
timeout 900 python -m pytest $R/tests/test_gpu_trainer.py -m gpu -x -q > $O/tests_trainer.txt 2>&1
tail -3 $O/tests_trainer.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], 'serial', d['serial']['ms_per_step'])
PY
