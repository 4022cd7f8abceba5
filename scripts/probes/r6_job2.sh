#!/bin/bash
# round 6, second GPU job: (A) the register-path ("direct") epilogue of gemm8p rebuilt this round -- libadvgrpo_hip.so here is built with P8_DIRECT=1,
# libadvgrpo_base.so with P8_DIRECT=0 -- parity, per-shape times, tile stamps, in-process step A/B; (B) the 4-wave two-workgroups-per-CU twin (gemm4w,
# experiments build, round-2 epilogue) against gemm8p on this round's box; (C) CFG halves on two streams ON TOP of two groups in flight
set -x
R=$PWD
O=$R/gpurun_out/r6_job2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 900 python -m pytest $R/tests/test_gpu_gemm.py $R/tests/test_gpu_fp8.py $R/tests/test_gpu_mmdit.py -m gpu -x -q > $O/tests_direct.txt 2>&1
tail -3 $O/tests_direct.txt
for i in 1 2; do for lib in base hip; do echo "== $lib" >> $O/pair_shapes_ab.txt; ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_$lib.so timeout 120 python $R/scripts/bench_pair_shapes.py 2>/dev/null | grep -v amdgpu >> $O/pair_shapes_ab.txt; done; done
timeout 600 python $R/scripts/probes/rollout_ab_inprocess.py $R/adv_grpo_amd/libadvgrpo_base.so $R/adv_grpo_amd/libadvgrpo_hip.so 3 5 2>/dev/null | grep -v amdgpu > $O/rollout_ab_direct.txt
export ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_experiments_direct.so
for m in plain gelu gateres rms; do
  N=6144; [ $m = gateres ] && N=1536; [ $m = rms ] && N=4608; [ $m = plain ] && N=1536
  echo "== class $m 16384 x $N x 1536 (direct epilogue)" >> $O/p8_stamps_direct.txt
  timeout 120 python $R/scripts/p8_stamps.py 16384 $N 1536 $m 2>/dev/null | grep -v amdgpu >> $O/p8_stamps_direct.txt
done
export ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_experiments.so
for i in 1 2; do
  echo "== gemm8p (round-5 epilogue)" >> $O/gemm4w_vs_8p.txt; timeout 120 python $R/scripts/bench_pair_shapes.py 2>/dev/null | grep -v amdgpu >> $O/gemm4w_vs_8p.txt
  echo "== gemm4w (ADVGRPO_GEMM_4W=1: 4-wave workgroups, two per CU, tile 128 x 256)" >> $O/gemm4w_vs_8p.txt; ADVGRPO_GEMM_4W=1 timeout 120 python $R/scripts/bench_pair_shapes.py 2>/dev/null | grep -v amdgpu >> $O/gemm4w_vs_8p.txt
done
unset ADVGRPO_LIB
export ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_base.so
timeout 600 python $R/bench.py --steps 10 --warmup 3 --no-epoch --no-pricing --no-cpu-baseline > $O/bench_trainer_schedule.json 2>/dev/null
timeout 600 python $R/bench.py --steps 10 --warmup 3 --no-epoch --no-pricing --no-cpu-baseline --cfg-streams > $O/bench_trainer_schedule_cfg_streams.json 2>/dev/null
unset ADVGRPO_LIB
ls -la $O
