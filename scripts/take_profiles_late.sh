#!/bin/bash
# Partial re-take at the end of round 5 (run through gpurun from the repo root), after the side-stream cache was removed and the LoRA
# weight gradients became one launch per group: the bench lines whose epoch legs changed (configs 2, 3, 4), the config-2 kernel table, the
# G-step micro-step (timing; kernel tables with the adapter gradients beside the chain and inside it), the grouped token-contracted launches.
#   scripts/take_profiles_late.sh   -> gpurun_out/prof_r5b/*   (copied into profiles/ by hand: names in DESIGN.md 6, round 5)
set -x
R=$PWD
O=$R/gpurun_out/prof_r5b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 5 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err
python $R/bench.py --config c3 --steps 5 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err
python $R/bench.py --config c4 --steps 3 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
rocprofv3 --kernel-trace --stats -d $O/kt_c2 -o x -- python $R/bench.py --steps 3 --warmup 1 --no-epoch --no-cpu-baseline --no-pricing > $O/bench_c2_prof.json 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_c2/x_results.db $O/kernel_stats_c2.md > /dev/null
python $R/scripts/gpu_idle.py $O/kt_c2/x_results.db > $O/gpu_idle_c2.txt
python $R/scripts/bench_gstep.py 2>/dev/null | grep -v amdgpu > $O/gstep.txt
python $R/scripts/bench_gstep.py nowgrad 2>/dev/null | grep -v amdgpu | head -1 >> $O/gstep.txt
python $R/scripts/bench_gstep.py serial 2>/dev/null | grep -v amdgpu | head -1 >> $O/gstep.txt
python $R/scripts/bench_gstep.py fp8 2>/dev/null | grep -v amdgpu >> $O/gstep.txt
rocprofv3 --kernel-trace --stats -d $O/kt_gstep -o x -- python $R/scripts/bench_gstep.py > $O/gstep_under_rocprof.txt 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_gstep/x_results.db $O/kernel_stats_gstep.md > /dev/null
rocprofv3 --kernel-trace --stats -d $O/kt_gstep_s -o x -- python $R/scripts/bench_gstep.py serial > $O/gstep_serial_under_rocprof.txt 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_gstep_s/x_results.db $O/kernel_stats_gstep_serial.md > /dev/null
python $R/scripts/bench_tn.py 2>/dev/null | grep -v amdgpu > $O/tn_grouped.txt
python $R/scripts/bench_attention.py 2>/dev/null | grep -v amdgpu > $O/attention_fwd_d64.txt
rm -rf $O/kt_c2 $O/kt_gstep $O/kt_gstep_s
ls -la $O
