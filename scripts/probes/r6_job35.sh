#!/bin/bash
# closing tree: the c2 kernel table (serial schedule, rocprofv3 --kernel-trace --stats) and the idle window
set -x
R=$PWD
O=$R/gpurun_out/r6_job35; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAST="--steps 3 --warmup 1 --no-epoch --no-cpu-baseline --no-pricing --schedule serial"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt_c2 -o x -- python $R/bench.py $FAST > $O/bench_c2_prof.json 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_c2/x_results.db $O/kernel_stats_c2.md > /dev/null
python $R/scripts/gpu_idle.py $O/kt_c2/x_results.db > $O/gpu_idle_c2.txt
rm -rf $O/kt_c2
head -14 $O/kernel_stats_c2.md | cut -c1-200
