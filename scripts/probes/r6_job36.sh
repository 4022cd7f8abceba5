#!/bin/bash
# 64-row tiles for the skinny adapter products of the G-step (N <= 64, M >= 1024): micro-step time, alternating, experiments build (the product library still has the old choice)
set -x
R=$PWD
O=$R/gpurun_out/r6_job36; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_experiments.so
for i in 1 2 3; do
  echo "== 128 x 64 tiles (ADVGRPO_GEMM_SKINNY=0)" >> $O/gstep.txt
  ADVGRPO_GEMM_SKINNY=0 timeout 300 python $R/scripts/bench_gstep.py 2>/dev/null | grep 'G-step' >> $O/gstep.txt
  echo "== 64 x 128 tiles" >> $O/gstep.txt
  timeout 300 python $R/scripts/bench_gstep.py 2>/dev/null | grep 'G-step' >> $O/gstep.txt
done
echo "== serial, old" >> $O/gstep.txt
ADVGRPO_GEMM_SKINNY=0 timeout 300 python $R/scripts/bench_gstep.py serial 2>/dev/null | grep 'G-step' >> $O/gstep.txt
echo "== serial, new" >> $O/gstep.txt
timeout 300 python $R/scripts/bench_gstep.py serial 2>/dev/null | grep 'G-step' >> $O/gstep.txt
cd $R && timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3 >> $O/gstep.txt
cat $O/gstep.txt
