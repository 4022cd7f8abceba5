#!/bin/bash
# Re-take of the profile files the late round-4 kernel work changes (head-dim-128 attention backward, head-dim-64 attention forward):
# run through gpurun from the repo root; outputs under gpurun_out/late, copied into profiles/ by hand (names in the commit).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/late; mkdir -p $O
cd $R
timeout 120 python scripts/probes/attn_bwd_d128_time.py 2>/dev/null | grep -v amdgpu > $O/attention_bwd_d128.txt
timeout 300 bash scripts/probes/attn_bwd_d128_variants.sh 0 1 17 2>/dev/null | grep -v amdgpu >> $O/attention_bwd_d128.txt
timeout 120 python scripts/bench_attention.py 2>/dev/null | grep -v amdgpu > $O/attention_fwd_d64.txt
timeout 300 python scripts/bench_gstep_qwen.py 60 fp8 2>/dev/null | grep -v amdgpu > $O/gstep_qwen.txt
timeout 300 python scripts/bench_gstep_qwen.py 60 2>/dev/null | grep -v amdgpu >> $O/gstep_qwen.txt
cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_gq -o x -- python $R/scripts/bench_gstep_qwen.py 6 fp8 > /dev/null 2>&1
timeout 60 python $R/scripts/rocpd_stats.py $O/kt_gq/x_results.db $O/kernel_stats_gstep_qwen_6_blocks.md > /dev/null 2>&1
rm -rf $O/kt_gq
cd $R
timeout 900 python bench.py --config c5 2>/dev/null | tail -1 > $O/bench_c5.json
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_c2_full.json
ls -la $O
