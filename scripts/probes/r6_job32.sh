#!/bin/bash
# what paces attention_fwd_pipe_kernel's main loop: timing-only ablations (all + 128 = never take the fallback): 1 = no wait + barrier, 4 = no LDS fragment
# reads in the slots, 8 = no DMA, 64 = no row sums
set -x
R=$PWD
O=$R/gpurun_out/r6_job32; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
echo "== product" >> $O/abl.txt
timeout 200 python $R/scripts/bench_attention.py 100 5 2>/dev/null | grep 'S=1229\|S=1024' >> $O/abl.txt
for v in 129 132 136 137 140 141 196; do
  echo "== ATT_ABL=$v" >> $O/abl.txt
  ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_abl$v.so timeout 200 python $R/scripts/bench_attention.py 100 5 2>/dev/null | grep 'S=1229\|S=1024' >> $O/abl.txt
done
echo "== product" >> $O/abl.txt
timeout 200 python $R/scripts/bench_attention.py 100 5 2>/dev/null | grep 'S=1229\|S=1024' >> $O/abl.txt
cat $O/abl.txt
