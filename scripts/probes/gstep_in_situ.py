"""Where the in-situ G-step loses time against scripts/bench_gstep.py: the epoch leg of bench.py with
the trainer's own per-micro-step HIP events, then the SAME trainer's g_step called again on the same samples (no sampling phase in
front), allocator statistics around both.  Usage: gstep_in_situ.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from adv_grpo_amd import synthetic  # noqa: E402
from adv_grpo_amd.config.experiments import get_config  # noqa: E402
from adv_grpo_amd.mmdit_train import SD3TransformerLoRA  # noqa: E402
from adv_grpo_amd.model_configs import ClipConfig, MMDiTConfig, VaeConfig  # noqa: E402
from adv_grpo_amd.pickscore_scorer import PickScoreScorer  # noqa: E402
from adv_grpo_amd.pipeline import SD3Pipeline  # noqa: E402
from adv_grpo_amd.trainer import SyntheticData, Trainer  # noqa: E402
from adv_grpo_amd.vae import AutoencoderKLDecoder  # noqa: E402

device = torch.device("cuda", 0)
# hypothesis check (round 4): streams created EARLIER in the process shift which hardware queue the adapter-gradient side stream
# lands on (torch hands out pool streams round-robin; HIP maps them onto a few hardware queues)
_dummies = [torch.cuda.Stream(device=device) for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 0)]
for _s in _dummies:                       # really used: a pool stream only exists on the HIP side once something ran on it
    with torch.cuda.stream(_s):
        torch.zeros(16, device=device).add_(1)
torch.cuda.synchronize()
_prio = int(os.environ.get("WGRAD_PRIO", "0"))
cfg = get_config("pickscore_cotrain_sd3_fast", gpu_number=1)
cfg.sample.num_image_per_prompt = 8
cfg.sample.num_batches_per_epoch = 2
cfg.train.gradient_accumulation_steps = 1
cfg.train_d = False
mcfg = MMDiTConfig()
with synthetic.on_device(device):
    tr = SD3TransformerLoRA(synthetic.mmdit_weights(mcfg, 1234), mcfg, device, seed=cfg.seed)
    vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 4321), VaeConfig(), device)
    scorer = PickScoreScorer(device, dtype=torch.bfloat16, model_sd=synthetic.clip_weights(ClipConfig(), 777), clip_cfg=ClipConfig())
if _prio:
    tr._wgrad_stream = torch.cuda.Stream(device=device, priority=-1)
trainer = Trainer(cfg, SD3Pipeline(tr, vae, device), SyntheticData(resolution=cfg.resolution, device=device), scorer, None, 0, 1, log_path=None)


def stats(tag):
    st = torch.cuda.memory_stats()
    print(tag, {k: st[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}, "reserved GiB",
          round(torch.cuda.memory_reserved() / 2 ** 30, 1), "allocated GiB", round(torch.cuda.memory_allocated() / 2 ** 30, 1))


def summarise(tag):
    torch.cuda.synchronize()
    out = {}
    for kind, e0, e1 in trainer.gstep_events:
        out.setdefault(kind, []).append(round(e0.elapsed_time(e1), 2))
    print(tag, out)


trainer.run_epoch()
stats("after warm-up epoch")
if len(sys.argv) > 1:          # pre-heat: N more sampling phases (sustained rollout load, ~0.8 s each) before the measurements
    for _ in range(int(sys.argv[1])):
        trainer.sample_epoch()
    torch.cuda.synchronize()
    print("pre-heated with", sys.argv[1], "sampling phases")
if os.environ.get("PROBE_RUN_EPOCH"):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trainer.run_epoch()
        torch.cuda.synchronize()
        summarise(f"run_epoch {rep} ({time.perf_counter() - t0:.3f} s):")
for rep in range(1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = trainer.sample_epoch()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    samples["advantages"] = torch.randn(samples["rewards"].shape[0], cfg.sample.train_num_steps, device=device)
    trainer.g_step(samples)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    summarise(f"epoch {rep}: sample {t1 - t0:.3f} s, g_step {t2 - t1:.3f} s (right after sampling):")
    stats("  allocator")
    for again in range(2):
        t3 = time.perf_counter()
        trainer.g_step(samples)
        torch.cuda.synchronize()
        summarise(f"  g_step again ({time.perf_counter() - t3:.3f} s, no sampling in front):")
    stats("  allocator")
