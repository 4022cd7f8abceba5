#!/bin/bash
# gemm8p epilogue with buffer stores (masked lanes get an out-of-range offset: no execution-mask branch per pass): libadvgrpo_hip.so = P8_BUF_STORES=1,
# libadvgrpo_base.so = 0.  Bit identity, per-shape times, tile stamps, whole-rollout A/B in one process, G micro-step.
set -x
R=$PWD
O=$R/gpurun_out/r6_job20; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 1200 python -m pytest $R/tests/test_gpu_gemm.py $R/tests/test_gpu_fp8.py $R/tests/test_gpu_mmdit.py $R/tests/test_gpu_edge_cases.py -m gpu -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for i in 1 2; do for lib in base hip; do echo "== $lib" >> $O/pair_shapes_ab.txt; ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_$lib.so timeout 120 python $R/scripts/bench_pair_shapes.py 2>/dev/null | grep -v amdgpu >> $O/pair_shapes_ab.txt; done; done
cat $O/pair_shapes_ab.txt
timeout 600 python $R/scripts/probes/rollout_ab_inprocess.py $R/adv_grpo_amd/libadvgrpo_base.so $R/adv_grpo_amd/libadvgrpo_hip.so 3 5 2>/dev/null | grep -v amdgpu > $O/rollout_ab.txt
cat $O/rollout_ab.txt
export ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_experiments.so
for m in plain gelu gateres rms; do
  N=6144; [ $m = gateres ] && N=1536; [ $m = rms ] && N=4608; [ $m = plain ] && N=1536
  echo "== class $m 16384 x $N x 1536 (buffer stores)" >> $O/p8_stamps_buf.txt
  timeout 120 python $R/scripts/p8_stamps.py 16384 $N 1536 $m 2>/dev/null | grep -v amdgpu >> $O/p8_stamps_buf.txt
done
unset ADVGRPO_LIB
cat $O/p8_stamps_buf.txt
for lib in base hip base hip; do echo "== $lib" >> $O/gstep_ab.txt; ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_$lib.so timeout 200 python $R/scripts/bench_gstep.py 2>/dev/null | grep -v amdgpu | head -1 >> $O/gstep_ab.txt; done
cat $O/gstep_ab.txt
