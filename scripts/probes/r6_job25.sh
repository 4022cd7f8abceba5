#!/bin/bash
# after the small-M tile choice + the resident hd-80 attention: the whole GPU suite, smoke(), the driver's command line, the c2 kernel table
set -x
R=$PWD
bash $R/scripts/probes/r6_final.sh
O=$R/gpurun_out/r6_job25; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAST="--steps 3 --warmup 1 --no-epoch --no-cpu-baseline --no-pricing --schedule serial"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt_c2 -o x -- python $R/bench.py $FAST > $O/bench_c2_prof.json 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_c2/x_results.db $O/kernel_stats_c2.md > /dev/null
python $R/scripts/gpu_idle.py $O/kt_c2/x_results.db > $O/gpu_idle_c2.txt
find $O -name '*.db' -size +30M -delete
head -12 $O/kernel_stats_c2.md | cut -c1-200
