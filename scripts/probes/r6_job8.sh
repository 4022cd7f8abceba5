#!/bin/bash
# round 6, eighth GPU job: the serial leg after prepare_streams (twice), then the whole GPU suite + smoke
set -x
R=$PWD
O=$R/gpurun_out/r6_job8
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
for i in 1 2; do timeout 600 python $R/bench.py --steps 12 --warmup 3 --no-epoch --no-pricing --no-cpu-baseline > $O/bench_$i.json 2>$O/bench_$i.err; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], 'serial', d['serial']['ms_per_step'])
PY
timeout 1500 python -m pytest $R/tests -m gpu -x -q > $O/full_gpu_tests.txt 2>&1
tail -4 $O/full_gpu_tests.txt
cd $R && timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
