#!/bin/bash
# PMC of the reward tower's kernels: LDS bank conflicts / matrix-pipe busy of the resident hd-80 attention against the tiled one, and of the small-M GEMM tiles
set -x
R=$PWD
O=$R/gpurun_out/r6_job27; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
C="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY"
timeout 600 rocprofv3 --pmc $C -d $O/pmc_new -o x -- python $R/scripts/probes/vit_tower_time.py 8 3 > /dev/null 2>&1
python $R/scripts/pmc_db.py $O/pmc_new/x_results.db advgrpo > $O/pmc_tower.txt
ADVGRPO_LIB=$R/adv_grpo_amd/libadvgrpo_experiments.so ADVGRPO_ATTN_NO_RESIDENT=1 timeout 600 rocprofv3 --pmc $C -d $O/pmc_old -o x -- python $R/scripts/probes/vit_tower_time.py 8 3 > /dev/null 2>&1
python $R/scripts/pmc_db.py $O/pmc_old/x_results.db attention > $O/pmc_tiled_attention.txt
rm -rf $O/pmc_new $O/pmc_old
cat $O/pmc_tower.txt $O/pmc_tiled_attention.txt
