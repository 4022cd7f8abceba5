#!/bin/bash
# the decoder's launch-stream side stream = a rollout stream: default line three times on one box, trainer tests, one line with the epoch leg
set -x
R=$PWD
O=$R/gpurun_out/r6_job11
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
F="--steps 12 --warmup 3 --no-epoch --no-pricing --no-cpu-baseline"
for i in 1 2 3; do timeout 600 python $R/bench.py $F > $O/bench_$i.json 2>/dev/null; done
timeout 600 python $R/bench.py --steps 12 --warmup 3 --no-pricing --no-cpu-baseline > $O/bench_with_epoch.json 2>$O/bench_with_epoch.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], 'serial', d['serial']['ms_per_step'], (d.get('epoch') or {}).get('images_per_s_full_epoch'), (d.get('epoch') or {}).get('phases_s'))
PY
timeout 900 python -m pytest $R/tests/test_gpu_trainer.py $R/tests/test_gpu_vae.py -m gpu -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
