#!/bin/bash
# end of round 6: the whole GPU suite, smoke(), and the driver's own command line
set -x
R=$PWD
O=$R/gpurun_out/r6_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 1500 python -m pytest $R/tests -m gpu -x -q > $O/full_gpu_tests.txt 2>&1
tail -4 $O/full_gpu_tests.txt
cd $R && timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
cd /tmp && timeout 900 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver_style_20_steps.json 2> $O/bench.err
tail -c 300 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_c2_driver_style_20_steps.json").read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['frac_of_bf16_mfma_peak'], d['serial'], d['roofline']['frac'], d['epoch']['images_per_s_full_epoch'], d['clock_and_power'])
PY
