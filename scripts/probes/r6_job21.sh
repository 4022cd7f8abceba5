#!/bin/bash
# where the ViT-H image tower's time goes: kernel trace of the tower alone
set -x
R=$PWD
O=$R/gpurun_out/r6_job21; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 300 python $R/scripts/probes/vit_tower_time.py 8 20 2>/dev/null | grep -v amdgpu > $O/tower.txt
timeout 300 python $R/scripts/probes/vit_tower_time.py 16 20 2>/dev/null | grep -v amdgpu >> $O/tower.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o vit -- python $R/scripts/probes/vit_tower_time.py 8 20 > $O/prof.log 2>&1
python $R/scripts/rocpd_stats.py $O/prof/vit_results.db $O/kernel_stats.md > /dev/null 2>&1 || true
ls -R $O/prof | head -20
find $O/prof -name '*.db' -size +30M -delete
cat $O/tower.txt; head -30 $O/kernel_stats.md
