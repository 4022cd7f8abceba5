#!/bin/bash
# small-M tile choice (gemm_variant: slots-aware 128x64 / 192x128 for 1024 <= M < 8192): the ViT-H tower, the text encoders, tests
set -x
R=$PWD
O=$R/gpurun_out/r6_job23; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
X=$R/adv_grpo_amd/libadvgrpo_experiments.so
for i in 1 2; do
  echo "== old dispatch (ADVGRPO_GEMM_SMALLM=0, experiments build)" >> $O/tower.txt
  ADVGRPO_LIB=$X ADVGRPO_GEMM_SMALLM=0 timeout 200 python $R/scripts/probes/vit_tower_time.py 8 30 2>/dev/null | grep tower >> $O/tower.txt
  ADVGRPO_LIB=$X ADVGRPO_GEMM_SMALLM=0 timeout 200 python $R/scripts/probes/vit_tower_time.py 16 30 2>/dev/null | grep tower >> $O/tower.txt
  echo "== new dispatch (product library)" >> $O/tower.txt
  timeout 200 python $R/scripts/probes/vit_tower_time.py 8 30 2>/dev/null | grep tower >> $O/tower.txt
  timeout 200 python $R/scripts/probes/vit_tower_time.py 16 30 2>/dev/null | grep tower >> $O/tower.txt
done
echo "== old" >> $O/text.txt
ADVGRPO_LIB=$X ADVGRPO_GEMM_SMALLM=0 timeout 300 python $R/scripts/bench_text_encoders.py 2>/dev/null | grep -v amdgpu >> $O/text.txt
echo "== new" >> $O/text.txt
timeout 300 python $R/scripts/bench_text_encoders.py 2>/dev/null | grep -v amdgpu >> $O/text.txt
ADVGRPO_LIB=$X timeout 100 python $R/scripts/probes/vit_gemm_variants.py 2056 2>/dev/null | grep FORCE >> $O/tower.txt
cd $R && timeout 1500 python -m pytest tests/test_gpu_vit.py tests/test_gpu_text_encoders.py tests/test_gpu_gemm.py tests/test_gpu_goldens.py tests/test_gpu_qwen_text.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt
cat $O/tower.txt $O/text.txt $O/tests.txt
