#!/bin/bash
# head-dim-80 attention with K / V resident in LDS: bit identity with the tiled kernel, time per launch, the ViT-H tower, tests
set -x
R=$PWD
O=$R/gpurun_out/r6_job24; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
X=$R/adv_grpo_amd/libadvgrpo_experiments.so
echo "== tiled kernel (ADVGRPO_ATTN_NO_RESIDENT=1)" > $O/attn.txt
ADVGRPO_LIB=$X ADVGRPO_ATTN_NO_RESIDENT=1 timeout 200 python $R/scripts/probes/attn_d80_ab.py 2>/dev/null | grep sha >> $O/attn.txt
echo "== resident kernel" >> $O/attn.txt
ADVGRPO_LIB=$X timeout 200 python $R/scripts/probes/attn_d80_ab.py 2>/dev/null | grep sha >> $O/attn.txt
for i in 1 2; do
  echo "== tiled" >> $O/attn.txt
  ADVGRPO_LIB=$X ADVGRPO_ATTN_NO_RESIDENT=1 timeout 200 python $R/scripts/probes/vit_tower_time.py 8 30 2>/dev/null | grep tower >> $O/attn.txt
  echo "== resident (product library)" >> $O/attn.txt
  timeout 200 python $R/scripts/probes/vit_tower_time.py 8 30 2>/dev/null | grep tower >> $O/attn.txt
done
cd $R && timeout 1500 python -m pytest tests/test_gpu_attention.py tests/test_gpu_vit.py tests/test_gpu_goldens.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt
cat $O/attn.txt $O/tests.txt
