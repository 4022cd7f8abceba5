#!/bin/bash
# the convolution's wide (192 x 256) tile: VAE tests, bit identity and decode time against the 192 x 128 tile at several dispatch thresholds
set -x
R=$PWD
O=$R/gpurun_out/r6_job33; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
X=$R/adv_grpo_amd/libadvgrpo_experiments.so
for rep in 1 2; do
for m in 1000000000 2048 1024 256; do
  ADVGRPO_LIB=$X ADVGRPO_X3_WIDE_MIN=$m timeout 300 python $R/scripts/probes/vae_wide_ab.py 8 64 2>/dev/null | grep WIDE_MIN >> $O/ab.txt
done; done
ADVGRPO_LIB=$X ADVGRPO_X3_WIDE_MIN=1000000000 timeout 300 python $R/scripts/probes/vae_wide_ab.py 4 128 2>/dev/null | grep WIDE_MIN >> $O/ab_1024.txt
ADVGRPO_LIB=$X ADVGRPO_X3_WIDE_MIN=2048 timeout 300 python $R/scripts/probes/vae_wide_ab.py 4 128 2>/dev/null | grep WIDE_MIN >> $O/ab_1024.txt
ADVGRPO_LIB=$X ADVGRPO_X3_WIDE_MIN=256 timeout 300 python $R/scripts/probes/vae_wide_ab.py 4 128 2>/dev/null | grep WIDE_MIN >> $O/ab_1024.txt
echo tests ran in the first pass > $O/tests.txt
cat $O/ab.txt $O/ab_1024.txt $O/tests.txt
