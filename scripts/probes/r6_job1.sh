#!/bin/bash
# round 6, first GPU job: targeted tests of the round's host-side changes, the driver-style line on the new default schedule, the yardsticks,
# the tile stamps of the four rollout epilogue classes, and a kernel trace of the serial schedule with the repaired gpu_idle window
set -x
R=$PWD
O=$R/gpurun_out/r6_job1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 900 python -m pytest $R/tests/test_gpu_vit.py $R/tests/test_gpu_trainer.py $R/tests/test_gpu_rollout.py $R/tests/test_hub.py -m gpu -x -q > $O/tests_targeted.txt 2>&1
tail -3 $O/tests_targeted.txt
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/bench_c2_driver_style.json 2> $O/bench_c2_driver_style.err
tail -c 600 $O/bench_c2_driver_style.err
timeout 300 python $R/scripts/yardstick_matmul.py 2>/dev/null | grep -v amdgpu > $O/yardstick_matmul.txt
timeout 300 python $R/scripts/yardstick_attention.py 2>/dev/null | grep -v amdgpu > $O/yardstick_attention.txt
export ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_experiments.so
for m in plain gelu gateres rms; do
  N=6144; [ $m = gateres ] && N=1536; [ $m = rms ] && N=4608; [ $m = plain ] && N=1536
  echo "== class $m 16384 x $N x 1536" >> $O/p8_stamps.txt
  timeout 120 python $R/scripts/p8_stamps.py 16384 $N 1536 $m 2>/dev/null | grep -v amdgpu >> $O/p8_stamps.txt
done
unset ADVGRPO_LIB
rocprofv3 --kernel-trace --stats -d $O/kt_c2 -o x -- python $R/bench.py --steps 3 --warmup 1 --no-epoch --no-cpu-baseline --no-pricing --schedule serial > $O/bench_c2_prof.json 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_c2/x_results.db $O/kernel_stats_c2.md > /dev/null
python $R/scripts/gpu_idle.py $O/kt_c2/x_results.db > $O/gpu_idle_c2.txt
rm -rf $O/kt_c2
ls -la $O
