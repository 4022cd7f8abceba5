#!/bin/bash
# s_setprio variants of attention_fwd_pipe_kernel (ATT_PRIO 1 / 2 / 3), the exponential floor (ATT_ABL = 130: v_exp -> v_mul, no fallback; WRONG results)
set -x
R=$PWD
O=$R/gpurun_out/r6_job31; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
for rep in 1 2; do
  echo "== product" >> $O/prio.txt
  timeout 200 python $R/scripts/bench_attention.py 100 5 2>/dev/null | grep 'S=1229\|S=1024' >> $O/prio.txt
  for v in ablp1 ablp2 ablp3 abl130 abl64; do
    echo "== $v" >> $O/prio.txt
    ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_$v.so timeout 200 python $R/scripts/bench_attention.py 100 5 2>/dev/null | grep 'S=1229\|S=1024' >> $O/prio.txt
  done
done
cat $O/prio.txt
