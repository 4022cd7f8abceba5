#!/bin/bash
# Re-takes every profiles/ artefact of a round on the GPU box (run through gpurun from the repo root):
#   scripts/take_profiles.sh r3        -> gpurun_out/prof_r3/*  (copy the summaries into profiles/ afterwards: scripts/collect_profiles.py)
set -x
TAG=${1:-r6}
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAST="--steps 3 --warmup 1 --no-epoch --no-cpu-baseline --no-pricing --schedule serial"     # (kernel tables: the serial schedule only)
# 1. driver-style lines
python $R/bench.py --steps 5 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err
python $R/bench.py --config c4 --steps 3 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
python $R/bench.py --config c3 --steps 5 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err
python $R/bench.py --config c5 --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
rocprofv3 --kernel-trace --stats -d $O/kt_c5 -o x -- python $R/bench.py --config c5 --steps 1 --warmup 1 --no-pricing --no-epoch > $O/bench_c5_prof.json 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_c5/x_results.db $O/kernel_stats_c5.md > /dev/null
rocprofv3 --kernel-trace --stats -d $O/kt_c3 -o x -- python $R/bench.py --config c3 --steps 1 --warmup 1 --no-pricing > $O/bench_c3_prof.json 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_c3/x_results.db $O/kernel_stats_c3_with_epoch_legs.md > /dev/null
python $R/scripts/bench_attention_d128.py 10 > $O/attention_d128.txt 2>/dev/null
python $R/scripts/bench_attention.py 2>/dev/null | grep -v amdgpu > $O/attention_fwd_d64.txt
python $R/scripts/bench_tn.py 2>/dev/null | grep -v amdgpu > $O/tn_grouped.txt
python $R/scripts/bench_attention_bwd.py > $O/attention_bwd.txt 2>/dev/null
# config 5's pieces: its VAE decoder, the head-dim-128 attention backward, one full-size G-step micro-batch, the kernel table of a 6-block one
export PYTHONPATH=$R
python $R/scripts/probes/qwen_vae_time.py 2>/dev/null | grep -v amdgpu > $O/qwen_vae.txt
python $R/scripts/probes/attn_bwd_d128_time.py 2>/dev/null | grep -v amdgpu > $O/attention_bwd_d128.txt
python $R/scripts/bench_gstep_qwen.py 60 fp8 2>/dev/null | grep -v amdgpu > $O/gstep_qwen.txt
python $R/scripts/bench_gstep_qwen.py 60 2>/dev/null | grep -v amdgpu >> $O/gstep_qwen.txt
rocprofv3 --kernel-trace --stats -d $O/kt_gq -o x -- python $R/scripts/bench_gstep_qwen.py 6 fp8 > /dev/null 2>&1
python $R/scripts/rocpd_stats.py $O/kt_gq/x_results.db $O/kernel_stats_gstep_qwen_6_blocks.md > /dev/null
GRAFT_REPO_ROOT=$R bash $R/scripts/probes/pmc_attention_bwd.sh 2 >> $O/attention_bwd.txt 2>/dev/null
cd /tmp
# 2. kernel tables: rollout (timed configuration only) and one G-step micro-batch
rocprofv3 --kernel-trace --stats -d $O/kt_c2 -o x -- python $R/bench.py $FAST > $O/bench_c2_prof.json 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_c2/x_results.db $O/kernel_stats_c2.md > /dev/null
python $R/scripts/gpu_idle.py $O/kt_c2/x_results.db > $O/gpu_idle_c2.txt
python $R/scripts/bench_gstep.py 2>/dev/null | grep -v amdgpu > $O/gstep.txt              # the number: without the profiler attached
python $R/scripts/bench_gstep.py nowgrad 2>/dev/null | grep -v amdgpu | head -1 >> $O/gstep.txt
python $R/scripts/bench_gstep.py serial 2>/dev/null | grep -v amdgpu | head -1 >> $O/gstep.txt
python $R/scripts/bench_gstep.py fp8 2>/dev/null | grep -v amdgpu >> $O/gstep.txt
# class switches of the update half, both variants in ONE process (alternating processes carry a position effect of 1 - 3 %)
python $R/scripts/probes/gstep_ab_inprocess.py fuse_gates 2>/dev/null | tail -2 > $O/gstep_ab_inprocess.txt
python $R/scripts/probes/gstep_ab_inprocess.py overlap_wgrad 2>/dev/null | tail -2 >> $O/gstep_ab_inprocess.txt
python $R/scripts/probes/gstep_ab_inprocess.py merge_one_launch 2>/dev/null | tail -2 >> $O/gstep_ab_inprocess.txt
[ -f $R/adv_grpo_amd/libadvgrpo_base.so ] && python $R/scripts/probes/attention_ab_inprocess.py 2>/dev/null | grep -v amdgpu > $O/attention_fwd_d64_ab.txt
rocprofv3 --kernel-trace --stats -d $O/kt_gstep -o x -- python $R/scripts/bench_gstep.py > $O/gstep_under_rocprof.txt 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_gstep/x_results.db $O/kernel_stats_gstep.md > /dev/null
rocprofv3 --kernel-trace --stats -d $O/kt_gstep_s -o x -- python $R/scripts/bench_gstep.py serial > $O/gstep_serial_under_rocprof.txt 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_gstep_s/x_results.db $O/kernel_stats_gstep_serial.md > /dev/null
# 3. HBM-side traffic (separate passes, MI355X_MICROARCH "HBM"), per config
for cfg in c2 c4 c5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${cfg}_$c -o x -- python $R/bench.py --config $cfg --steps 1 --warmup 1 --no-epoch --no-cpu-baseline --no-pricing --schedule serial > /dev/null 2>&1
  done
  mkdir -p $O/pmc_$cfg && cp -r $O/pmc_${cfg}_FETCH_SIZE $O/pmc_${cfg}_WRITE_SIZE $O/pmc_$cfg/
  python $R/scripts/pmc_traffic.py $O/pmc_$cfg $O/pmc_traffic_$cfg.json > $O/pmc_traffic_$cfg.txt
done
# 4. matrix-pipe busy share of the MFMA kernels
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY -d $O/pmc_mfma -o x -- python $R/bench.py --steps 1 --warmup 1 --no-epoch --no-cpu-baseline --no-pricing --schedule serial > /dev/null 2>&1
python $R/scripts/pmc_db.py $O/pmc_mfma/x_results.db advgrpo > $O/pmc_mfma.txt
rm -rf $O/kt_c2 $O/kt_c3 $O/kt_c5 $O/kt_gstep $O/kt_gstep_s $O/kt_gq $O/pmc_c2* $O/pmc_c4_* $O/pmc_c4 $O/pmc_c5_* $O/pmc_c5 $O/pmc_mfma
ls -la $O
