#!/bin/bash
# does the device ever wait for the host with two groups in flight?  kernel trace of the default schedule, busy share per 50 ms
set -x
R=$PWD
O=$R/gpurun_out/r6_job17; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
rocprofv3 --kernel-trace -d $O/kt -o x -- python $R/bench.py --steps 6 --warmup 2 --no-epoch --no-pricing --no-cpu-baseline > $O/bench.json 2>/dev/null
python $R/scripts/gpu_busy_bins.py $O/kt/x_results.db 50 > $O/busy_bins.txt
rm -rf $O/kt
tail -c 600 $O/bench.json | head -c 300; echo; cat $O/busy_bins.txt | tail -130
