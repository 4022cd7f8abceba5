#!/bin/bash
# stream set-up variants of the in-flight leg on ONE box (experiment knobs via environment, removed afterwards)
set -x
R=$PWD
O=$R/gpurun_out/r6_job10
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
F="--steps 12 --warmup 3 --no-epoch --no-pricing --no-cpu-baseline"
for i in 1 2; do
  timeout 600 python $R/bench.py $F > $O/v0_current_$i.json 2>/dev/null
  ADVGRPO_BENCH_NO_VAE_PREP=1 timeout 600 python $R/bench.py $F > $O/v1_noprep_$i.json 2>/dev/null
  ADVGRPO_BENCH_NO_VAE_PREP=1 ADVGRPO_BENCH_OLD_PARTNERS=1 timeout 600 python $R/bench.py $F > $O/v2_noprep_oldpartners_$i.json 2>/dev/null
  ADVGRPO_BENCH_OLD_PARTNERS=1 timeout 600 python $R/bench.py $F > $O/v3_prep_oldpartners_$i.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], 'serial', d['serial']['ms_per_step'])
PY
