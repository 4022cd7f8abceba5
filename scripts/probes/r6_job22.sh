#!/bin/bash
# the ViT-H layer's four Linears at M = 2056 (and 4112) under each tile variant
set -x
R=$PWD
O=$R/gpurun_out/r6_job22; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_experiments.so
for M in 2056 4112; do
for v in auto 15 26 30 0 1 2 14 27 21; do
  if [ $v = auto ]; then timeout 120 python $R/scripts/probes/vit_gemm_variants.py $M 2>/dev/null | grep FORCE >> $O/variants.txt
  else ADVGRPO_GEMM_FORCE=$v timeout 120 python $R/scripts/probes/vit_gemm_variants.py $M 2>/dev/null | grep FORCE >> $O/variants.txt; fi
done; done
cat $O/variants.txt
