# matrix-pipe busy share, VALU / LDS counters of the attention backward kernels (one shape).  Usage: pmc_attention_bwd.sh [shape index] [lib]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$2" ] && export ADVGRPO_LIB=$R/adv_grpo_amd/$2
rm -rf /tmp/pmc_bwd
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY -d /tmp/pmc_bwd -o x -- python $R/scripts/bench_attention_bwd.py ${1:-0} > /dev/null 2>&1
python $R/scripts/pmc_db.py /tmp/pmc_bwd/x_results.db attn_bwd
