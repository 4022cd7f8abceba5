#!/usr/bin/env python3
"""Runs the same short full-size training (config 2 shapes, synthetic weights / prompts) once with bf16 and once with fp8 block
Linears and prints the per-update metrics of both side by side: are the fp8 mode's importance ratios, clip fractions and losses
those of the bf16 mode?  Usage: compare_training_modes.py [epochs]   (writes logs/mode_{bf16,fp8}.jsonl)"""
import json
import os
import subprocess
import sys

epochs = sys.argv[1] if len(sys.argv) > 1 else "6"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
logs = {}
for mode in ("bf16", "fp8"):
    path = os.path.join(root, "logs", f"mode_{mode}.jsonl")
    if os.path.exists(path):
        os.remove(path)
    subprocess.check_call([sys.executable, os.path.join(root, "scripts", "train_sd3_fast.py"), "--config",
                           "config/grpo.py:pickscore_cotrain_sd3_fast", "--no-train-d", "--epochs", epochs, "--batches", "2", "--images-per-prompt", "8",
                           "--linear-dtype", mode, "--log", path], cwd=root, stdout=subprocess.DEVNULL)
    logs[mode] = [json.loads(l) for l in open(path)]
keys = ("loss", "policy_loss", "approx_kl", "clipfrac")
print("update | " + " | ".join(f"{k}: bf16 / fp8" for k in keys))
ups = {m: [r for r in logs[m] if "approx_kl" in r] for m in logs}
for a, b in zip(ups["bf16"], ups["fp8"]):
    print(f"{a['step']:6d} | " + " | ".join(f"{a[k]:+.3e} / {b[k]:+.3e}" for k in keys))
rew = {m: [r["reward_avg"] for r in logs[m] if "reward_avg" in r] for m in logs}
print("reward_avg per epoch  bf16:", " ".join(f"{v:.4f}" for v in rew["bf16"]))
print("reward_avg per epoch  fp8 :", " ".join(f"{v:.4f}" for v in rew["fp8"]))
