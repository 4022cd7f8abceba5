#!/bin/bash
# what the row sums / the exponentials / the fallback check cost in attention_fwd_pipe_kernel (timing-only ablations; results WRONG with any bit set)
set -x
R=$PWD
O=$R/gpurun_out/r6_job30; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
for rep in 1 2; do
  echo "== product" >> $O/ablate.txt
  timeout 200 python $R/scripts/bench_attention.py 100 5 2>/dev/null | grep 'S=1229\|S=1024' >> $O/ablate.txt
  for v in 64 2 66 128; do
    echo "== ATT_ABL=$v" >> $O/ablate.txt
    ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_abl$v.so timeout 200 python $R/scripts/bench_attention.py 100 5 2>/dev/null | grep 'S=1229\|S=1024' >> $O/ablate.txt
  done
done
cat $O/ablate.txt
