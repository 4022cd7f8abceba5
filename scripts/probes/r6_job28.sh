#!/bin/bash
# resident hd-80 attention with swizzled 256-byte K rows: bit identity, time, PMC conflict share, tests
set -x
R=$PWD
O=$R/gpurun_out/r6_job28; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
X=$R/adv_grpo_amd/libadvgrpo_experiments.so
echo "== tiled kernel (ADVGRPO_ATTN_NO_RESIDENT=1)" > $O/attn.txt
ADVGRPO_LIB=$X ADVGRPO_ATTN_NO_RESIDENT=1 timeout 200 python $R/scripts/probes/attn_d80_ab.py 2>/dev/null | grep sha >> $O/attn.txt
echo "== resident kernel" >> $O/attn.txt
ADVGRPO_LIB=$X timeout 200 python $R/scripts/probes/attn_d80_ab.py 2>/dev/null | grep sha >> $O/attn.txt
timeout 200 python $R/scripts/probes/vit_tower_time.py 8 30 2>/dev/null | grep tower >> $O/attn.txt
C="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY"
timeout 600 rocprofv3 --pmc $C -d $O/pmc_new -o x -- python $R/scripts/probes/vit_tower_time.py 8 3 > /dev/null 2>&1
python $R/scripts/pmc_db.py $O/pmc_new/x_results.db attention > $O/pmc_resident.txt
rm -rf $O/pmc_new
cd $R && timeout 1500 python -m pytest tests/test_gpu_attention.py tests/test_gpu_vit.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
cat $O/attn.txt $O/pmc_resident.txt $O/tests.txt
