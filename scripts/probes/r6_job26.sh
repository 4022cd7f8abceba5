#!/bin/bash
# final tree: the secondary lines (c3 / c4 / c5) after the small-M tile choice, and one more default c2 line on this box
set -x
R=$PWD
O=$R/gpurun_out/r6_job26; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout 900 python $R/bench.py --config c3 --steps 5 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 1200 python $R/bench.py --config c4 --steps 3 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 1200 python $R/bench.py --config c5 --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
python - <<PY
import json
for c in ("c2","c3","c4","c5"):
    try:
        d=json.loads(open("$O/bench_%s.json"%c).read().strip().splitlines()[-1])
        print(c, d['value'], d['ms_per_step'], d.get('frac_of_bf16_mfma_peak'), d['serial']['ms_per_step'], d['roofline']['frac'], d.get('epoch',{}).get('images_per_s_full_epoch'), d['clock_and_power']['sclk_mhz_median'])
    except Exception as e:
        print(c, "FAILED", e)
PY
