#!/bin/bash
# the PickScore scorer (ViT-H, M = 2056 rows: 170 tiles of 128 x 128 for N = 1280) with every small-M GEMM forced to another tile variant (experiments build)
set -x
R=$PWD
O=$R/gpurun_out/r6_job15; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_experiments.so
for v in auto 1 2 0 14 15; do
  echo "== ADVGRPO_GEMM_FORCE=$v" >> $O/scorer_variants.txt
  if [ $v = auto ]; then timeout 200 python $R/scripts/probes/scorer_gemm_shapes.py 2>/dev/null | grep -v amdgpu | head -3 >> $O/scorer_variants.txt
  else ADVGRPO_GEMM_FORCE=$v timeout 200 python $R/scripts/probes/scorer_gemm_shapes.py 2>/dev/null | grep -v amdgpu | head -3 >> $O/scorer_variants.txt; fi
done
cat $O/scorer_variants.txt
