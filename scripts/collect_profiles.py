#!/usr/bin/env python3
"""Copies the summaries scripts/take_profiles.sh left under gpurun_out/prof_<tag>/ into profiles/ (tracked) under round names and
derives the in-clock matrix-pipe utilisation of the MFMA kernels from the PMC pass:
    util = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs  /  (SQ_WAVE_CYCLES * 4 / (waves_per_SIMD * 1024))
(SQ_WAVE_CYCLES counts quad-cycles summed over waves; the MFMA kernels here run 2 waves per SIMD).
Usage: collect_profiles.py <tag> <round-prefix>      e.g.  collect_profiles.py r3a r3"""
import json, os, re, shutil, sys
tag, pre = sys.argv[1], sys.argv[2]
src, dst = f"gpurun_out/prof_{tag}", "profiles"
names = {"bench_c2.json": f"{pre}_bench_c2_full.json", "bench_c4.json": f"{pre}_bench_c4.json", "bench_c2_prof.json": f"{pre}_bench_c2_prof.json",
         "kernel_stats_c2.md": f"{pre}_bench_kernel_stats.md", "kernel_stats_gstep.md": f"{pre}_gstep_kernel_stats.md", "gstep.txt": f"{pre}_gstep.txt",
         "pmc_traffic_c2.json": f"{pre}_pmc_traffic_c2.json", "pmc_traffic_c4.json": f"{pre}_pmc_traffic_c4.json",
         "pmc_traffic_c5.json": f"{pre}_pmc_traffic_c5.json", "bench_c3.json": f"{pre}_bench_c3.json", "bench_c5.json": f"{pre}_bench_c5.json",
         "kernel_stats_c5.md": f"{pre}_kernel_stats_c5.md", "kernel_stats_c3_with_epoch_legs.md": f"{pre}_kernel_stats_c3_with_epoch_legs.md",
         "attention_d128.txt": f"{pre}_attention_d128.txt", "attention_bwd.txt": f"{pre}_attention_bwd.txt", "gstep_under_rocprof.txt": f"{pre}_gstep_under_rocprof.txt",
         "qwen_vae.txt": f"{pre}_qwen_vae.txt", "attention_bwd_d128.txt": f"{pre}_attention_bwd_d128.txt", "gstep_qwen.txt": f"{pre}_gstep_qwen.txt",
         "kernel_stats_gstep_qwen_6_blocks.md": f"{pre}_kernel_stats_gstep_qwen_6_blocks.md",
         "attention_fwd_d64.txt": f"{pre}_attention_fwd_d64.txt", "tn_grouped.txt": f"{pre}_tn_grouped.txt",
         "kernel_stats_gstep_serial.md": f"{pre}_gstep_serial_kernel_stats.md", "gpu_idle_c2.txt": f"{pre}_gpu_idle_c2.txt",
         "gstep_ab_inprocess.txt": f"{pre}_gstep_ab_inprocess.txt"}
for a, b in names.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
rows = {}
for line in open(os.path.join(src, "pmc_mfma.txt")):
    m = re.match(r"(.{50})\s+(SQ_\w+)\s+([0-9.e+]+) n=(\d+)", line)
    if m:
        rows.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
        rows[m.group(1).strip()]["launches"] = int(m.group(4))
out = {}
for k, c in rows.items():
    if c.get("SQ_INSTS_MFMA", 0) < 1e5:
        continue
    waves_per_simd = 2
    elapsed = c["SQ_WAVE_CYCLES"] * 4 / (waves_per_simd * 1024)
    out[k] = {"launches": c["launches"], "mfma_busy_cycles_per_simd": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024, "elapsed_cycles": elapsed,
              "mfma_util_in_clock": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / elapsed, 4),
              "valu_insts_per_mfma": round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], 2),
              "lds_bank_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]), 4),
              "issue_stall_share_of_wave_cycles": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)}
json.dump(out, open(os.path.join(dst, f"{pre}_pmc_mfma.json"), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_simd"] * kv[1]["launches"]):
    print(f"{k[:60]:60s} util {v['mfma_util_in_clock']:.3f}  VALU/MFMA {v['valu_insts_per_mfma']:6.2f}  bank-conflict {v['lds_bank_conflict_share']:.4f}  issue-stall {v['issue_stall_share_of_wave_cycles']:.3f}")
