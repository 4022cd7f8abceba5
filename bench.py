#!/usr/bin/env python3
"""bench.py -- sampled+scored images/sec of the Adv-GRPO SD3 rollout hot path on MI355X.

One "step" = one pass of the hot path over one prompt group per rank (BASELINE.json config 2):
SD3.5-medium shapes, 512x512, 10 denoise steps with CFG 4.5 (transformer batch 16), G = 8 images per prompt,
SDE window [0, 2) at noise level 0.8 with per-step log-probs, VAE decode, PickScore (CLIP ViT-H/14) reward
of the 8 generated images, and -- for N > 1 -- the packed all-gather of rewards + group ids over RCCL
followed by the redundant group-advantage kernel (the path's only exchange step; the G-sample groups are
sharded across ranks like the reference's DistributedKRepeatSampler, no data-path collective).
Synthetic seeded weights / prompt embeddings (no checkpoints on the box), inputs resident in HBM.

Prints ONE JSON line (rank 0).  Launch for N > 1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")    # before the HIP runtime starts: see adv_grpo_amd/__init__.py (stream -> hardware queue map)

import torch  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
FP8_DENSE_PEAK_TFLOPS = 5000.0    # MI355X_MICROARCH.md: ~5 PF dense fp8 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--no-epoch", action="store_true", help="skip the untimed-for-the-metric full-epoch leg")
    ap.add_argument("--epoch", action="store_true", help="(kept for compatibility: the full-epoch leg, with the LoRA-gradient "
                                                         "all-reduce of the G-step, now runs by default at every N)")
    ap.add_argument("--spawn", action="store_true", help="take the self-spawn path (torch.distributed.run, RCCL process group) at N = 1 "
                                                         "as well: exercises the launcher on a one-GPU box")
    ap.add_argument("--event-stride", type=int, default=13,
                    help="HIP events around every Nth launch of each GEMM kernel in the timed steps (1 = every launch).  The roofline's "
                         "per-launch average is then a 1-in-N sample; a prime N walks through the 4 / 6 launches of a block.  Events "
                         "around all ~1220 GEMM launches of a step cost 1.7 %% of the step (measured, LABNOTES.md 6)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 = the headline (BASELINE config 2); c3 = BASELINE config 3's rank-local work (config 2's rollout with the co-trained "
                         "DINOv2-patch discriminator as reward; its epoch leg runs one D epoch and one G epoch); c4 = secondary line, SD3.5-large 1024^2 G=4 (BASELINE config 4 shapes); "
                         "c5 = secondary line, Qwen-Image MMDiT 1024^2 G=8, DINO reward, fp8 Linears (BASELINE config 5 shapes)")
    ap.add_argument("--decode-in-future", type=int, default=0, help="experiment (measured, no gain: LABNOTES 6 round 5): 1 = the VAE decode runs inside "
                    "the reward future as well (the rollout returns latents, output_type='latent')")
    ap.add_argument("--rollout-priority", type=int, default=0, help="experiment (measured, no gain): -1 = the timed rollouts on a high-priority HIP stream")
    ap.add_argument("--dry-run", action="store_true",
                    help="N > 1 (or --spawn): bring the process group up, check that every rank is there (all-reduce of ones, device "
                         "names), time the path's two collectives alone, print that as the JSON line and stop -- no model is built")
    ap.add_argument("--groups-in-flight", type=int, default=2, help="experiment: prompt groups in flight per rank on the trainer schedule (the Trainer default is 2)")
    ap.add_argument("--vae-one-stream", action="store_true", help="experiment: decode a group on ONE stream (default: two half batches on two streams)")
    ap.add_argument("--cfg-streams", action="store_true", help="experiment: the two halves of the CFG batch as two forwards on two HIP streams")
    ap.add_argument("--sync-scoring", action="store_true",
                    help="score each group on the launch stream right after its decode instead of on the reward-future stream (SURVEY 8a11)")
    ap.add_argument("--no-pricing", action="store_true",
                    help="skip the untimed legs that price the alternative modes (split-bf16 VAE, LoRA side path): profiling runs, "
                         "so that the rocprof summary holds the timed configuration only")
    ap.add_argument("--schedule", default="trainer", choices=["trainer", "serial"],
                    help="what the timed steps run.  trainer (default) = the Trainer's own default schedule (config sample.groups_in_flight = 2, "
                         "trainer.py sample_epoch): two prompt groups rolled out at the same time, each on its own HIP stream from its own host "
                         "thread, scoring as reward futures on one scoring stream, gather + advantage per group on the main thread in group order; "
                         "a serial leg of the same number of steps follows and carries the per-kernel roofline.  serial = one prompt group at a "
                         "time on the launch stream (the headline of rounds 1-5)")
    ap.add_argument("--vae-mode", default="bf16x3", choices=["bf16", "bf16x3"],
                    help="decoder arithmetic inside the timed step: the fp32-equivalent split-bf16 mode (default: the reference "
                         "decodes in fp32, TP:481) or plain bf16; the 'vae' object of the JSON line prices both either way")
    return ap.parse_args()


class PowerSampler:
    """rocm-smi samples (shader clock, socket power) of one GPU beside the timed steps, on a host thread: the rollout runs at
    the package power limit and the sustained clock, not the nominal one, is what the MFMA peak scales with."""

    def __init__(self, gpu_index, period=0.5):
        import threading
        self.gpu, self.period, self.samples, self._stop = gpu_index, period, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "-d", str(self.gpu), "--showpower", "--showclocks"], capture_output=True, text=True,
                                     timeout=5).stdout
                clk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
                pw = re.search(r"Power \(W\): ([0-9.]+)", out)
                if clk and pw:
                    self.samples.append((int(clk.group(1)), float(pw.group(1))))
            except Exception:
                return
            self._stop.wait(self.period)

    def __enter__(self):
        if self.gpu >= 0:                 # (only rank 0 samples)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._t.is_alive():
            self._t.join(timeout=6)

    def summary(self):
        if not self.samples:
            return None
        clk = sorted(c for c, _ in self.samples)
        pw = sorted(p for _, p in self.samples)
        med = lambda v: v[len(v) // 2]
        return {"samples": len(self.samples), "sclk_mhz_median": med(clk), "sclk_mhz_min": clk[0], "sclk_mhz_max": clk[-1],
                "socket_power_w_median": med(pw), "socket_power_w_max": pw[-1],
                "note": "rocm-smi every 0.5 s during the timed steps (rank 0's GPU); nominal peak assumes 2400 MHz"}


C5_TEXT_TOKENS = 128      # synthetic prompt length of the config-5 line (Qwen-Image prompts are variable-length Qwen2.5-VL states)


def build_c5(device, vae_mode="bf16x3"):
    """BASELINE config 5's model set: the Qwen-Image MMDiT (60 blocks, 24 x 128, fp8 Linears), Qwen-Image's own VAE decoder (the
    Wan-style causal 3-D VAE run on one frame, adv_grpo_amd/qwen_vae.py) and the DINOv2-B/14 patch scorer + head."""
    from adv_grpo_amd import synthetic, vit
    from adv_grpo_amd.model_configs import DinoConfig, QwenMMDiTConfig, QwenVaeConfig
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.qwen_mmdit import QwenImageTransformer2DModel
    from adv_grpo_amd.qwen_vae import AutoencoderKLQwenImageDecoder
    qcfg, vcfg, dcfg = QwenMMDiTConfig(), QwenVaeConfig(), DinoConfig()
    with synthetic.on_device(device):
        tr = QwenImageTransformer2DModel(synthetic.qwen_mmdit_weights(qcfg, 4242, dtype=torch.bfloat16), qcfg, device)
        vae = AutoencoderKLQwenImageDecoder(synthetic.qwen_vae_decoder_weights(vcfg, 2468, dtype=torch.bfloat16), vcfg, device, mode=vae_mode)
        dino = vit.DinoV2(synthetic.dino_weights(dcfg, 888), dcfg, device)
        head = vit.DinoHead(synthetic.dino_head_weights(dcfg.hidden, 512, 999), device)
    tr.enable_fp8()
    return SD3Pipeline(tr, vae, device), (dino, head)


def build(device, large=False, vae_mode="bf16x3"):
    from adv_grpo_amd import synthetic, vit
    from adv_grpo_amd.mmdit import SD3Transformer2DModel
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    from adv_grpo_amd.model_configs import ClipConfig, MMDiTConfig, VaeConfig
    mcfg, vcfg, ccfg = MMDiTConfig(), VaeConfig(), ClipConfig()
    if large:     # stabilityai/stable-diffusion-3.5-large: 38 joint blocks, 38 heads x 64 = 2432, no dual-attention blocks
        mcfg = MMDiTConfig(num_layers=38, num_heads=38, dual_attention_layers=(), pos_embed_max_size=192)
    with synthetic.on_device(device):
        tr = SD3Transformer2DModel(synthetic.mmdit_weights(mcfg, 1234), mcfg, device)
        vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(vcfg, 4321, fp16_checkpoint=True), vcfg, device, mode=vae_mode)
        # config 2's reward is the co-trained bf16 scorer (TP:514); config 4's `pickscore` reward is the fp32 scorer of the
        # reward factory (RW:561-574): the fp32-equivalent split-bf16 towers
        from adv_grpo_amd import vit_x3
        clip = (vit_x3.CLIPModelX3 if large else vit.CLIPModel)(synthetic.clip_weights(ccfg, 777), ccfg, device)
    return SD3Pipeline(tr, vae, device), clip


def pmc_traffic(kernel, config="c2"):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes OF THE SAME CONFIG (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs of this same command, scripts/pmc_traffic.py: (2*FETCH_SIZE + WRITE_SIZE)*1024, the factor 2
    being the gfx950 correction of MI355X_MICROARCH.md); None when no pass of that config is committed."""
    root = os.path.dirname(os.path.abspath(__file__))
    files = (f"profiles/r6_pmc_traffic_{config}.json", f"profiles/r5_pmc_traffic_{config}.json", f"profiles/r4_pmc_traffic_{config}.json", f"profiles/r3_pmc_traffic_{config}.json") + \
        (("profiles/r2_pmc_traffic.json", "profiles/r1_pmc_traffic.json") if config == "c2" else ())
    for rel in files:
        path = os.path.join(root, rel)
        if not os.path.exists(path):
            continue
        tot, n = 0.0, 0
        fp8 = kernel.endswith("_fp8")                 # ops' name of the fp8 instantiations: gemm8p_kernel<PAIR, class, true>
        base = kernel[:-4] if fp8 else kernel
        for name, v in json.load(open(path)).items():
            short = name.replace(" ", "").split("advgrpo::")[-1]
            if (short == base or short.startswith(base + "<")) and short.endswith(",true>") == fp8:   # all instantiations of that operand type
                tot += v["hbm_bytes_per_launch"] * v["launches"]
                n += v["launches"]
        if n:
            return round(tot / n), rel
    return None, None


def full_epoch(device, world=1, rank=0, adversarial=False, qwen=False, large=False):
    """the whole sample -> score -> gather -> advantage -> G-step loop and its phases, outside the timed
    region of the headline metric: config 2 (pickscore_cotrain_sd3_fast preset, 8 images per prompt so that one rank
    holds whole groups), 2 prompt groups per epoch = 16 images, 2 optimizer steps; the second epoch is reported.
    adversarial (config 3): the dino_cotrain_sd3_patch_fast preset with its DINOv2-B/14 + head discriminator (TD:156-232,
    1091-1115), d_times = 2 here so that a D epoch and a G epoch alternate: after a warm-up pair, one of each is timed.
    qwen (config 5): the same adversarial preset on the Qwen-Image model set at 1024^2 -- QwenImageTransformerLoRA (60 blocks, fp8
    Linears in rollout AND replay, one activation checkpoint per block), Qwen-Image's own VAE decoder, DINOv2-B/14 + head; ONE prompt
    group (8 images) per epoch."""
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.config.experiments import get_config
    from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
    from adv_grpo_amd.model_configs import ClipConfig, MMDiTConfig, VaeConfig
    from adv_grpo_amd.pickscore_scorer import PickScoreScorer
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.trainer import SyntheticData, Trainer
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    cfg = get_config("pickscore_sd3_fast" if large else ("dino_cotrain_sd3_patch_fast" if adversarial else "pickscore_cotrain_sd3_fast"),
                     gpu_number=world)
    cfg.sample.num_image_per_prompt = 8
    cfg.sample.num_batches_per_epoch = 1 if qwen else 2
    cfg.train.gradient_accumulation_steps = 1
    if qwen:
        cfg.resolution = 1024
        cfg.linear_dtype = "fp8"
    if large:      # config 4: SD3.5-large 1024^2, G = 4, the PickScore + OCR multi-reward preset (config/grpo.py:379-427), no discriminator
        cfg.resolution = 1024
        cfg.sample.num_image_per_prompt = cfg.sample.mini_num_image_per_prompt = 4
    if adversarial:
        cfg.d_times = 2                      # D epoch, G epoch, D epoch, ... (the shipped preset: 9 D epochs per G epoch, TD:1097)
    else:
        cfg.train_d = False                  # G epochs only (the D/G gate depends on random rewards here)
    mcfg = MMDiTConfig(num_layers=38, num_heads=38, dual_attention_layers=(), pos_embed_max_size=192) if large else MMDiTConfig()
    head = None
    with synthetic.on_device(device):
        if qwen:
            from adv_grpo_amd.model_configs import QwenMMDiTConfig, QwenVaeConfig
            from adv_grpo_amd.qwen_mmdit_train import QwenImageTransformerLoRA
            from adv_grpo_amd.qwen_vae import AutoencoderKLQwenImageDecoder
            tr = QwenImageTransformerLoRA(synthetic.qwen_mmdit_weights(QwenMMDiTConfig(), 4242, dtype=torch.bfloat16), QwenMMDiTConfig(), device,
                                          seed=cfg.seed)
            vae = AutoencoderKLQwenImageDecoder(synthetic.qwen_vae_decoder_weights(QwenVaeConfig(), 2468, dtype=torch.bfloat16), QwenVaeConfig(), device)
        else:
            tr = SD3TransformerLoRA(synthetic.mmdit_weights(mcfg, 1234), mcfg, device, seed=cfg.seed)
            vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 4321, fp16_checkpoint=True), VaeConfig(), device)
        if large:
            from adv_grpo_amd import rewards
            rewards.configure_pickscore(synthetic.clip_weights(ClipConfig(), 777), ClipConfig())
            rewards.configure_ocr(lambda img: "prompt")        # PaddleOCR is not in the image: the host plugin runs with a stand-in recogniser
            scorer = None
        elif adversarial:
            from adv_grpo_amd import vit
            from adv_grpo_amd.d_step import DinoHeadTrainable
            from adv_grpo_amd.model_configs import DinoConfig
            scorer = vit.DinoV2(synthetic.dino_weights(DinoConfig(), 888), DinoConfig(), device)
            head = DinoHeadTrainable(device=device, seed=0)
        else:
            scorer = PickScoreScorer(device, dtype=torch.bfloat16, model_sd=synthetic.clip_weights(ClipConfig(), 777), clip_cfg=ClipConfig())
    data = SyntheticData(n_tokens=C5_TEXT_TOKENS, ctx_dim=3584, pooled_dim=8, resolution=cfg.resolution, device=device) if qwen else \
        SyntheticData(resolution=cfg.resolution, device=device)
    trainer = Trainer(cfg, SD3Pipeline(tr, vae, device), data, scorer, head, rank, world, log_path=None)
    for _ in range(2 if adversarial else 1):
        trainer.run_epoch()                  # warm-up: one epoch (config 2) / a D epoch and a G epoch (config 3)
    trainer.timers.clear()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    epoch_s = []
    t0 = time.perf_counter()
    for _ in range(2 if adversarial else 1):
        te = time.perf_counter()
        out = trainer.run_epoch()
        torch.cuda.synchronize()
        epoch_s.append((out["phase"], time.perf_counter() - te))
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    # the G-step from the inside: HIP events the trainer recorded around every micro-step / gradient all-reduce / optimizer step
    g_inside = {}
    for kind, e0, e1 in getattr(trainer, "gstep_events", []):
        g_inside.setdefault(kind, []).append(e0.elapsed_time(e1))
    g_inside = {k: {"n": len(v), "mean_ms": round(sum(v) / len(v), 2), "min_ms": round(min(v), 2), "max_ms": round(max(v), 2),
                    **({"ms_in_order": [round(x, 1) for x in v]} if len(v) <= 8 else {})}
                for k, v in g_inside.items()}
    phases = dict(trainer.timers)
    if world > 1:                              # slowest rank per phase (every rank walks the phases in the same order)
        keys = sorted(phases)
        pm = torch.tensor([phases[k] for k in keys], dtype=torch.float64, device=device)
        dist.all_reduce(pm, op=dist.ReduceOp.MAX)
        phases = dict(zip(keys, pm.tolist()))
    images = world * cfg.sample.num_batches_per_epoch * cfg.sample.mini_num_image_per_prompt * len(epoch_s)
    if large:
        adv_note = {"peak_memory_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                    "config4_note": "SD3.5-large (38 blocks, D = 2432) LoRA, 1024^2, two groups of G = 4 per epoch, reward = 0.5 PickScore (fp32-equivalent "
                                    "scorer) + 0.5 OCR (host plugin, stand-in recogniser), no discriminator: sample + 4 micro-steps at CFG batch 8 + 2 "
                                    "optimizer steps"}
    adv = {}
    if adversarial:      # the shipped preset runs 9 D epochs per G epoch (d_times = 10): the blended rate of this rank's timings
        d_s = next(t for ph, t in epoch_s if ph == "D")
        g_s = next(t for ph, t in epoch_s if ph == "G")
        adv = {"epochs_timed": [ph for ph, _ in epoch_s], "d_epoch_s": round(d_s, 3), "g_epoch_s": round(g_s, 3),
               "images_per_s_at_d_times_10": round(world * cfg.sample.num_batches_per_epoch * 8 * 10 / (9 * d_s + g_s), 3),
               "adversarial_note": "D epoch = sample + score (generated and reference images through DINOv2-B/14 @ 518 + head) + train_dino "
                                   f"(hinge on CLS + 0.3 x hinge on 64 patches, {8 * cfg.sample.num_batches_per_epoch} real + {8 * cfg.sample.num_batches_per_epoch} generated images, Adam on the head, TD:156-232); "
                                   "G epoch = the same sampling + the GRPO update; phases_s sums both epochs (this rank's host clock)"}
    if qwen:
        adv["peak_memory_gib"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
        adv["config5_note"] = ("Qwen-Image MMDiT LoRA (r = 32 on the 8 attention projections of all 60 blocks: 189 M padded adapter parameters), "
                               "1024^2, ONE group of 8 images per epoch: G epoch = sampling + 2 SDE timesteps x (forward with one checkpoint "
                               "per block + backward with per-block recomputation) at CFG batch 16 + clip + AdamW + EMA; fp8 Linears in the "
                               "rollout and in the replay (straight-through backward)")
    if large:
        adv.update(adv_note)
    return {"images": images, "seconds": round(dt, 3), "images_per_s_full_epoch": round(images / dt, 3), **adv,
            "phases_s": {k: round(v, 4) for k, v in phases.items()}, "phases_are": "max over ranks" if world > 1 else "rank 0",
            "g_step_inside": g_inside, "g_step_inside_is": "HIP events on the launch stream around each call, this rank, in situ (after the "
                                                           "sampling phase of the same epoch)",
            "exchanges": "reward all-gather once per epoch; all-reduce of the flat LoRA gradient before each optimizer step (TP:1165)",
            "note": ("sample = rollout + VAE decode; score = the DINOv2 patch discriminator on generated AND reference images; g_step = 1 group x 2 SDE "
                     "timesteps fwd+bwd at CFG batch 16 + 1 clip+AdamW step + EMA (+ for N > 1 the all-reduce of the 755 MB flat LoRA gradient "
                     "before the optimizer step)") if qwen else
                    ("sample = rollout + VAE decode; score = PickScore of generated AND reference images; g_step = "
                     "2 groups x 2 SDE timesteps fwd+bwd at CFG batch 16 + 2 clip+AdamW steps + EMA (+ for N > 1 the all-reduce of "
                     "the 37.6 MB flat LoRA gradient before each optimizer step); reward scoring runs on the worker stream and "
                     "overlaps the next group's rollout, `score` is the wait at the end of the sampling loop")}


def shared_gpu_collectives(dist):
    """ADVGRPO_BENCH_SHARED_GPU=1: every rank on cuda:0, process group over gloo, device tensors staged through the host inside the
    collectives.  NOT a measurement mode (the ranks share one GPU, the numbers mean nothing): it exists so that the N > 1 branches of
    this file and of the Trainer (rank partition, reward gather, gradient average, state broadcast, scaling diagnostics, the epoch
    leg at world 2) can execute end to end on a ONE-GPU box -- RCCL refuses two ranks on one device ("Duplicate GPU detected"), and
    no multi-GPU box has been available to this build."""
    dist.init_process_group("gloo")
    real = {k: getattr(dist, k) for k in ("all_reduce", "all_gather_into_tensor", "broadcast")}

    def all_reduce(t, *a, **kw):
        if not t.is_cuda:
            return real["all_reduce"](t, *a, **kw)
        h = t.cpu()
        real["all_reduce"](h, *a, **kw)
        t.copy_(h)

    def all_gather_into_tensor(out, t, *a, **kw):
        if not t.is_cuda:
            return real["all_gather_into_tensor"](out, t, *a, **kw)
        ho, hi = out.cpu(), t.cpu()
        real["all_gather_into_tensor"](ho, hi, *a, **kw)
        out.copy_(ho)

    def broadcast(t, *a, **kw):
        if not t.is_cuda:
            return real["broadcast"](t, *a, **kw)
        h = t.cpu()
        real["broadcast"](h, *a, **kw)
        t.copy_(h)
    dist.all_reduce, dist.all_gather_into_tensor, dist.broadcast = all_reduce, all_gather_into_tensor, broadcast


def cpu_baseline():
    """BASELINE config 1 on the host cores through the CPU oracle (kind "port": the reference's wheels --
    diffusers / timm / peft -- are not installed, so its own Python cannot run): SD3.5-medium shapes, 256x256,
    4 steps, G=2, CFG, VAE decode, PickScore, fp32, torch CPU threads = all cores."""
    from adv_grpo_amd import synthetic
    from oracle import mmdit as o_m
    from oracle import rewards as o_rw
    from oracle import rollout as o_r
    from oracle import vae as o_v
    from oracle import vit as o_t
    from oracle.scheduler import FlowMatchEulerScheduler
    cores = torch.get_num_threads()
    mcfg, vcfg, ccfg = o_m.MMDiTConfig(), o_v.VaeConfig(), o_t.ClipConfig()
    Wm = synthetic.mmdit_weights(mcfg, 1234)
    Wv = synthetic.vae_decoder_weights(vcfg, 4321)
    Wc = synthetic.clip_weights(ccfg, 777)
    pe, ppe, npe, nppe = synthetic.prompt_embeddings(7)
    G, hw, steps = 2, 256, 4
    ids = synthetic.clip_input_ids(G, 3)
    tr = lambda x, t, c, p: o_m.mmdit_forward(Wm, mcfg, x, t, c, p)
    t0 = time.time()
    with torch.no_grad():
        img, _, _, _ = o_r.rollout(tr, lambda z: o_v.vae_decode(Wv, vcfg, z), FlowMatchEulerScheduler(), prompt_embeds=pe,
                                   pooled_prompt_embeds=ppe, negative_prompt_embeds=npe,
                                   negative_pooled_prompt_embeds=nppe, num_inference_steps=steps, guidance_scale=4.5,
                                   height=hw, width=hw, noise_level=0.8, mini_num_image_per_prompt=G, train_num_steps=2,
                                   process_index=0, sample_num_steps=steps, random_timestep=0)
        px = torch.nn.functional.interpolate(img, size=(224, 224), mode="bicubic", antialias=True)
        px = (px - torch.tensor(o_rw.CLIP_MEAN)[None, :, None, None]) / torch.tensor(o_rw.CLIP_STD)[None, :, None, None]
        s = o_rw.pickscore_from_embeddings(o_t.clip_image_features(Wc, ccfg, px), o_t.clip_text_features(Wc, ccfg, ids),
                                           Wc["logit_scale"])
    dt = time.time() - t0
    assert torch.isfinite(s).all()
    # ---- config 2's unit costs, measured (SURVEY 8d "one C2 step if time allows"): one CFG transformer forward for one
    # image at 512^2 (batch 2, 1229 tokens), one 512^2 VAE decode, one PickScore pair -> 10 x forward + decode + score
    with torch.no_grad():
        x2 = torch.randn(2, 16, 64, 64)
        t0 = time.time()
        o_m.mmdit_forward(Wm, mcfg, x2, torch.full((2,), 900.0), torch.cat([npe, pe]), torch.cat([nppe, ppe]))
        t_fwd = time.time() - t0
        t0 = time.time()
        img2 = o_v.postprocess(o_v.vae_decode(Wv, vcfg, torch.randn(1, 16, 64, 64) / vcfg.scaling_factor + vcfg.shift_factor))
        t_vae = time.time() - t0
        t0 = time.time()
        px2 = torch.nn.functional.interpolate(img2, size=(224, 224), mode="bicubic", antialias=True)
        o_rw.pickscore_from_embeddings(o_t.clip_image_features(Wc, ccfg, px2), o_t.clip_text_features(Wc, ccfg, ids[:1]), Wc["logit_scale"])
        t_clip = time.time() - t0
    c2_s_per_image = 10 * t_fwd + t_vae + t_clip
    return {"value": G / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"config 1 end to end: SD3.5-medium 256x256, 4 steps, G=2, CFG 4.5, VAE decode + PickScore, fp32 torch-CPU "
                      f"oracle ({dt:.1f} s for {G} images); config 2 from its measured unit costs on the same cores: one CFG "
                      f"transformer forward per image at 512^2 {t_fwd:.2f} s, one 512^2 VAE decode {t_vae:.2f} s, one PickScore "
                      f"{t_clip:.2f} s -> 10 steps = {c2_s_per_image:.1f} s per sampled+scored image",
            "config2_images_per_s": round(1.0 / c2_s_per_image, 5),
            "config2_unit_costs_s": {"transformer_cfg_forward_512": round(t_fwd, 3), "vae_decode_512": round(t_vae, 3),
                                     "pickscore": round(t_clip, 3)}}


def scaling_diagnostics(dist, device, world, rank, G, T, dt_local, dt_max, steps, step, D):
    """What a poor 1 -> N curve would need to name its cause, from the same run (N > 1, or --spawn at N = 1): every rank's own time
    for the timed steps, the path's two collectives timed alone with HIP events on the launch stream (the packed reward all-gather of
    TP:926-966 and the all-reduce of the flat 18.78 M-parameter f32 LoRA gradient, TP:1165), and a SOLO leg -- the same step
    without the exchange, no barrier, each rank on its own -- whose rate stands in for "N = 1 in this run"."""
    def gathered(x):
        out = torch.zeros(world, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(out, torch.tensor([x], dtype=torch.float64, device=device))
        return out.tolist()

    def timed_ms(fn, reps):
        fn()
        dist.barrier()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps
    per_rank_ms = [t / steps * 1e3 for t in gathered(dt_local)]
    rw = torch.rand(G, T, device=device)
    gi = torch.full((G,), rank, dtype=torch.int32, device=device)
    ag_ms = timed_ms(lambda: D.gather_rewards(rw, gi), 20)
    flat = torch.ones(18_776_064, dtype=torch.float32, device=device)          # 191 adapters x 98 304 parameters
    ar_ms = timed_ms(lambda: D.average_gradients(flat), 5)
    del flat
    step(0, exchange=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(2):
        step(1 + it, exchange=False)
    torch.cuda.synchronize()
    solo_ms = gathered((time.perf_counter() - t0) / 2 * 1e3)
    value = world * G * steps / dt_max
    solo_rank0 = G / (solo_ms[0] * 1e-3)
    return {"ms_per_step_by_rank": {"min": round(min(per_rank_ms), 2), "max": round(max(per_rank_ms), 2),
                                    "all": [round(v, 2) for v in per_rank_ms]},
            "reward_all_gather_ms": round(ag_ms, 4), "lora_gradient_all_reduce_ms": round(ar_ms, 3),
            "lora_gradient_bytes": 18_776_064 * 4,
            "solo_ms_per_step_by_rank": {"min": round(min(solo_ms), 2), "max": round(max(solo_ms), 2), "all": [round(v, 2) for v in solo_ms]},
            "value_at_n1_same_run": round(solo_rank0, 3),
            "value_over_n_times_n1": round(value / (world * solo_rank0), 4),
            "value_over_sum_of_solo_rates": round(value / sum(G / (m * 1e-3) for m in solo_ms), 4),
            "note": "solo = the same step without the exchange, every rank at the same time on its own GPU (rank 0's rate = value_at_n1_same_run); "
                    "value / (N x that) below 1 is what the barrier + the slowest rank + the collectives cost; collectives timed alone, "
                    "HIP events around back-to-back calls"}


def host_rss_mb():
    import resource
    return round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)        # (Linux: kilobytes)


def print_dry_run(dist, device, ranks_seen, world, rank):
    """--dry-run: what the first multi-GPU line must show before anything expensive runs -- every rank present, one GPU each, the two
    collectives of the path alive and timed (reward all-gather TP:926-966, the 75 MB LoRA-gradient all-reduce TP:1165)."""
    from adv_grpo_amd import distributed as D
    res = {"dry_run": True, "n_gpus": world, "ranks_seen": ranks_seen, "host_rss_mb_rank0": host_rss_mb()}
    if dist is not None:
        names = [None] * world
        dist.all_gather_object(names, f"rank {rank}: {torch.cuda.get_device_name(device)} (cuda:{device.index}, pid {os.getpid()})")
        G, T = 8, 2
        rw, gi = torch.rand(G, T, device=device), torch.full((G,), rank, dtype=torch.int32, device=device)
        flat = torch.ones(18_776_064, dtype=torch.float32, device=device)

        def timed_ms(fn, reps):
            fn()
            dist.barrier()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / reps
        res.update(ranks=names, reward_all_gather_ms=round(timed_ms(lambda: D.gather_rewards(rw, gi), 20), 4),
                   lora_gradient_all_reduce_ms=round(timed_ms(lambda: D.average_gradients(flat), 5), 3), lora_gradient_bytes=18_776_064 * 4,
                   all_reduce_of_ones_after_averaging=float(flat[0].item()))
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_command(argv, gpus, port=None, script=None):
    """The launch line of scripts/grpo_pickscore.sh:7-11 (one process per GPU of one node), for a plain
    `python bench.py --gpus N` started without a launcher: the same script re-executed under torch.distributed.run."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script or os.path.abspath(__file__)] + list(argv)


def self_spawn(args):
    """--gpus N > 1 without WORLD_SIZE in the environment: start the N ranks ourselves and pass rank 0's JSON line on."""
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus))))
    raise SystemExit(subprocess.run(spawn_command(sys.argv[1:], args.gpus), env=env).returncode)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()))
        return
    if (args.gpus > 1 or args.spawn) and "WORLD_SIZE" not in os.environ:
        self_spawn(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    shared_gpu = os.environ.get("ADVGRPO_BENCH_SHARED_GPU", "0") == "1"     # integration-test aid, see shared_gpu_collectives()
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.spawn:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("NCCL_DEBUG", "WARN")                # RCCL's own warnings on stderr (the driver keeps the tail)
        if shared_gpu:
            shared_gpu_collectives(dist)
        else:
            # (a finite timeout: a rank that died before the rendezvous ends the others with an error instead of a hang)
            dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=600))      # RCCL over xGMI
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    ranks_seen = 1
    if dist is not None:                                       # RCCL sanity: every rank contributes a one
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        assert ranks_seen == world, f"all-reduce of ones saw {ranks_seen} ranks, expected {world}"
    if args.dry_run:
        print_dry_run(dist, device, ranks_seen, world, rank)
        return

    from adv_grpo_amd import distributed as D
    from adv_grpo_amd import ops, stat_tracking, synthetic, vit
    from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
    from adv_grpo_amd.sampler import DistributedKRepeatSampler
    from adv_grpo_amd.trainer import rollout_seed

    c3, c4, c5 = args.config == "c3", args.config == "c4", args.config == "c5"
    if c5:
        pipe, (dino, dino_head) = build_c5(device, vae_mode=args.vae_mode)
        clip = None
        from adv_grpo_amd import rewards
        dino_score = rewards.dino_patch_cotrain_score(device)
    elif c3:    # config 2's generator, the co-trained DINOv2-B/14 patch discriminator + head as the reward (RW:375-434)
        pipe, _unused = build(device, large=False, vae_mode=args.vae_mode)
        del _unused
        clip = None
        from adv_grpo_amd import rewards, vit as _vit
        from adv_grpo_amd.model_configs import DinoConfig
        with synthetic.on_device(device):
            dino = _vit.DinoV2(synthetic.dino_weights(DinoConfig(), 888), DinoConfig(), device)
            dino_head = _vit.DinoHead(synthetic.dino_head_weights(768, 512, 999), device)
        dino_score = rewards.dino_patch_cotrain_score(device)
    else:
        pipe, clip = build(device, large=c4, vae_mode=args.vae_mode)
    G, STEPS, T, RES = (4, 10, 2, 1024) if c4 else ((8, 10, 2, 1024) if c5 else (8, 10, 2, 512))
    if args.cfg_streams:
        pipe.cfg_two_streams = True
    if args.vae_one_stream and hasattr(pipe.vae, "two_streams"):
        pipe.vae.two_streams = False
    sampler = DistributedKRepeatSampler(range(25432), 1, 1, world, rank, seed=42)   # k = 1: one group per rank
    # synthetic prompts: one embedding set per dataset index is not needed for timing; a fixed set per rank
    text_tower = None
    if c5:     # Qwen-Image: 3584-wide text states, no pooled vector (a dummy one travels through the rollout's signature)
        pe, ppe, npe, nppe = (t.to(device=device, dtype=torch.bfloat16)
                              for t in synthetic.prompt_embeddings(7 + rank, n_tokens=C5_TEXT_TOKENS, ctx_dim=3584, pooled_dim=8))
        if rank == 0 and not args.no_pricing:
            # the prompt encoder itself (Qwen2.5-VL language model, 28 layers, random weights), OUTSIDE the timed region as for every
            # config (inputs are resident when the step starts): its states replace the random ones, its time is reported
            from adv_grpo_amd.model_configs import QwenTextConfig
            from adv_grpo_amd.qwen_text_encoder import DROP_IDX, Qwen25VLTextEncoder
            tcfg = QwenTextConfig()
            with synthetic.on_device(device):
                enc = Qwen25VLTextEncoder(synthetic.qwen_text_weights(tcfg, 1357, dtype=torch.bfloat16), tcfg, device)
            tok = torch.randint(0, tcfg.vocab_size, (2, DROP_IDX + C5_TEXT_TOKENS), generator=torch.Generator().manual_seed(11)).to(device)
            tmask = torch.ones(2, DROP_IDX + C5_TEXT_TOKENS, dtype=torch.long)
            enc.encode_prompt(tok, tmask)
            torch.cuda.synchronize()
            tt = time.perf_counter()
            for _ in range(3):
                emb, _m = enc.encode_prompt(tok, tmask)
            torch.cuda.synchronize()
            text_tower = {"ms_per_prompt_pair": round((time.perf_counter() - tt) / 3 * 1e3, 2), "tokens": DROP_IDX + C5_TEXT_TOKENS,
                          "note": "Qwen2.5-VL language model (28 layers, 3584 wide, random weights) on (prompt, negative prompt) token ids "
                                  "after the chat template; the first 34 template tokens dropped; outside the timed step"}
            pe, npe = emb[:1].contiguous(), emb[1:].contiguous()
            del enc, emb
            torch.cuda.empty_cache()
    else:
        pe, ppe, npe, nppe = (t.to(device=device, dtype=torch.bfloat16) for t in synthetic.prompt_embeddings(7 + rank))
    # one prompt per group (TP:813-817 repeat the group's prompt G times): the scorers see G equal prompts
    ids = synthetic.clip_input_ids(1, 3 + rank).repeat(G, 1).to(device)

    # Reward futures (TP:668 executor, TP:816-817 submit, TP:839-856 resolve): the scorer of group i runs on its own HIP
    # stream behind an event and is resolved one step later, so it overlaps the rollout of group i + 1 -- the rollouts themselves stay
    # serial (one prompt group at a time on the launch stream).  --sync-scoring puts it back on the launch stream.
    score_stream = None if args.sync_scoring else torch.cuda.Stream(device=device)

    def score(image):
        if c5 or c3:    # the co-trained DINOv2 patch scorer (RW:375-434): bicubic -> 518, ViT-B/14, 64 random patches, head
            scores, _ = dino_score(dino, dino_head, image.to(torch.bfloat16), None, None)
        elif c4:    # fp32 scorer (RW:561-574)
            from adv_grpo_amd import vit_x3
            scores = vit_x3.pickscore_scores_f32(clip.get_image_features(images=image.to(torch.bfloat16)), clip.get_text_features(ids[:1]).expand(G, -1).contiguous(),    # (TP:816: images.to(bf16) before any reward fn)
                                                 clip.logit_scale)
        else:       # (the text tower on the group's ONE distinct prompt, as PickScoreScorer does from the host strings)
            scores = vit.pickscore_scores(clip.get_image_features(images=image.to(torch.bfloat16)),
                                          clip.get_text_features(ids[:1]).expand(G, -1).contiguous(), clip.logit_scale)
        return scores

    import threading
    score_lock = threading.Lock()       # the towers keep per-model workspaces: scoring calls are enqueued one at a time (trainer.py: one scoring worker)

    def prompt_of(it):
        sampler.set_epoch(it)
        return next(iter(sampler))[0]

    def rollout_and_submit(it, worker=None, prompt_idx=None):
        """Rollout (+ decode) of one prompt group on the current stream; its scoring on `worker` (None: the current stream)."""
        if prompt_idx is None:
            prompt_idx = prompt_of(it)
        image, lats, lps, tss = pipeline_with_logprob_random(
            pipe, prompt_embeds=pe, pooled_prompt_embeds=ppe, negative_prompt_embeds=npe,
            negative_pooled_prompt_embeds=nppe, num_inference_steps=STEPS, guidance_scale=4.5,
            height=RES, width=RES, noise_level=0.8, mini_num_image_per_prompt=G, train_num_steps=T,
            process_index=rank, sample_num_steps=STEPS, random_timestep=0, seed=rollout_seed(42, it, rank),
            output_type="latent" if (worker is not None and args.decode_in_future) else "pt")
        if worker is None:
            with score_lock:
                return prompt_idx, lps, score(image), None
        ready = torch.cuda.Event()
        ready.record()
        with score_lock, torch.cuda.stream(worker):
            worker.wait_event(ready)
            if args.decode_in_future:
                latents = image
                image = pipe.vae.decode_to_image(latents)                   # PF:667-670, moved behind the event
                latents.record_stream(worker)
            scores = score(image)
            done = torch.cuda.Event()
            done.record(worker)
        if not args.decode_in_future:
            image.record_stream(worker)
        return prompt_idx, lps, scores, done

    def finish(pending, exchange=True):
        prompt_idx, lps, scores, done = pending
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
            scores.record_stream(torch.cuda.current_stream())
        rewards = scores.unsqueeze(1).repeat(1, T)                          # TP:926-928
        gids = torch.full((G,), prompt_idx, dtype=torch.int32, device=device)
        if not exchange:                                                    # (the solo leg of the scaling diagnostics below)
            return stat_tracking.group_advantage(rewards, gids, True), torch.stack(lps, 1)
        rewards, gids = D.gather_rewards(rewards, gids)                     # TP:930-966 packed into one all-gather
        adv = stat_tracking.group_advantage(rewards, gids, True)            # TP:970 (global_std)
        return D.ungather(adv, world, rank), torch.stack(lps, 1)            # TP:995-999

    def step(it, exchange=True):
        """One group start to finish on the current stream (the pricing legs, the two-groups leg, the scaling diagnostics)."""
        return finish(rollout_and_submit(it), exchange)

    def steps_pipelined(first, n):
        """n timed steps: rollout of group i, then the (already submitted) scores of group i - 1 are gathered and normalised."""
        pend, out = None, None
        for it in range(n):
            cur = rollout_and_submit(first + it, score_stream)
            if pend is not None:
                out = finish(pend)
            pend = cur
        return finish(pend)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The Trainer's default schedule (config sample.groups_in_flight = 2; trainer.py sample_epoch, TP:668,816-817): two rollout worker threads,
    # one HIP stream each (measured concurrent, like the Trainer's), scoring as reward futures on the one scoring stream; the reward gather and
    # the group advantage of every group run on the MAIN thread in group order (so the collectives of all ranks stay in one order).
    in_flight = max(2, args.groups_in_flight) if args.schedule == "trainer" else 1
    roll_pool, roll_tls, roll_streams = None, threading.local(), []
    if in_flight > 1:
        from concurrent.futures import ThreadPoolExecutor
        roll_pool = ThreadPoolExecutor(max_workers=in_flight, thread_name_prefix="bench-rollout")
        for _ in range(in_flight):     # measured concurrent with the launch stream, with each other AND with the scoring stream (a rollout stream that
            # shares a hardware queue with the scoring stream serialises that group's rollout behind the other group's reward future)
            partners = [torch.cuda.current_stream(device)] + ([score_stream] if score_stream is not None else []) + list(roll_streams)
            roll_streams.append(ops.concurrent_stream(device, partners))
        roll_lock = threading.Lock()
    # The decoder's side stream for decodes issued on the launch stream (serial leg, pricing legs), fixed NOW (a lazily created one cost the serial
    # leg 10 %): a rollout stream when there are any -- they are idle whenever such a decode runs, and a fifth live stream cost the in-flight leg
    # 2.5 % (vae.py set_side_streams) -- else one chosen by measurement.  Trainer.__init__ does the same.
    if hasattr(pipe.vae, "prepare_streams") and getattr(pipe.vae, "mode", None) == "bf16x3":
        if roll_streams:
            pipe.vae.set_side_streams(torch.cuda.current_stream(device), list(roll_streams))
        else:
            pipe.vae.prepare_streams([torch.cuda.current_stream(device)], also=[score_stream] if score_stream is not None else [])

    def steps_in_flight(first, n):
        """n timed steps on the Trainer's schedule: every group is submitted to the two rollout workers up front, its scores are a future on
        the scoring stream, and the main thread finishes the groups (gather, advantage) in order as they complete."""
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)

        def work(it, prompt_idx):
            torch.cuda.set_device(device)
            st = getattr(roll_tls, "stream", None)
            if st is None:                                   # one stream per WORKER THREAD (trainer.py sample_epoch)
                with roll_lock:
                    st = roll_tls.stream = roll_streams.pop()
            st.wait_event(ready)
            with torch.cuda.stream(st):
                pend = rollout_and_submit(it, score_stream if score_stream is not None else None, prompt_idx)
                fin = torch.cuda.Event()
                fin.record(st)
            return pend, fin
        futs = [roll_pool.submit(work, first + i, prompt_of(first + i)) for i in range(n)]
        out = None
        for f in futs:
            pend, fin = f.result()
            main.wait_event(fin)
            for t in pend[1]:                                # log-probs: produced on a rollout stream, consumed on this one
                t.record_stream(main)
            if pend[3] is None:
                pend[2].record_stream(main)
            out = finish(pend)
        return out

    timed_steps = steps_in_flight if in_flight > 1 else steps_pipelined
    # as Trainer.sample_epoch: with two groups in flight the decoder keeps a group on ONE stream (its two-half-batches split is for a decoder
    # that has the GPU to itself: the serial leg and the pricing legs below)
    vae_split = getattr(pipe.vae, "two_streams", None)
    if vae_split is not None and in_flight > 1:
        pipe.vae.two_streams = False
    if args.warmup:
        timed_steps(0, args.warmup)
    ops.PROFILE, ops.PROFILE_STRIDE = ([] if in_flight == 1 else None), max(1, args.event_stride)
    sync()
    with PowerSampler(local_rank if rank == 0 and os.environ.get("ADVGRPO_BENCH_NO_SMI", "0") != "1" else -1) as power:
        t0 = time.perf_counter()
        if args.rollout_priority:
            hp = torch.cuda.Stream(device=device, priority=args.rollout_priority)
            hp.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(hp):
                out = timed_steps(args.warmup, args.steps)
            torch.cuda.current_stream().wait_stream(hp)
        else:
            out = timed_steps(args.warmup, args.steps)           # every group's scores are resolved inside the timed region
        sync()
        dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all()
    if vae_split is not None and not args.vae_one_stream:
        pipe.vae.two_streams = vae_split
    # The serial leg (trainer schedule only): the same number of steps, one prompt group at a time on the launch stream.  Per-kernel
    # durations are only defined here -- with two groups in flight a launch's HIP-event interval includes the other stream's kernels -- so
    # the `roofline` object is taken from this leg and `serial` carries its whole-step figures.  Every rank runs it (it has the collectives).
    serial_dt = None
    if in_flight > 1:
        steps_pipelined(0, 1)
        ops.PROFILE = []
        sync()
        ts = time.perf_counter()
        out_s = steps_pipelined(args.warmup, args.steps)
        sync()
        serial_dt = time.perf_counter() - ts
        prof, ops.PROFILE = ops.PROFILE, None
        assert torch.isfinite(out_s[0]).all() and torch.isfinite(out_s[1]).all()
        if dist is not None:
            tmax_s = torch.tensor([serial_dt], dtype=torch.float64, device=device)
            dist.all_reduce(tmax_s, op=dist.ReduceOp.MAX)
            serial_dt = tmax_s.item()
    scaling_diag = None
    if dist is not None:
        dt_local = dt
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
        scaling_diag = scaling_diagnostics(dist, device, world, rank, G, T, dt_local, dt, args.steps, step, D)

    # The epoch leg runs HERE, right behind the timed steps, for every configuration whose two model sets fit one GPU side by side (all but
    # config 5): behind the pricing legs below -- which create and drop a dozen HIP streams -- its rollout and adapter-gradient streams landed
    # on shared hardware queues (round 5, same box: sampling phase 0.89 s against 0.74 s, and 109 ms micro-steps with a re-used side stream).
    run_epoch = not args.no_epoch                  # at every N: for N > 1 this is the leg with the LoRA-gradient all-reduce (TP:1165)
    ep_early = None
    if run_epoch and not c5:                        # every rank takes part (LoRA-gradient all-reduce, reward gather)
        ep_early = full_epoch(device, world, rank, adversarial=c3, qwen=False, large=c4)

    if rank == 0:
        # ---- roofline of the dominant kernel from the HIP events recorded around every GEMM launch
        per = {}
        for name, flops, s, e, _shape in prof:
            a = per.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1; a[1] += flops; a[2] += s.elapsed_time(e) * 1e-3
        dom = max(per, key=lambda k: per[k][2])
        n, fl, tsec = per[dom]
        achieved = fl / tsec / 1e12
        traffic, traffic_src = pmc_traffic(dom, args.config)
        stride = max(1, args.event_stride)
        kpeak = FP8_DENSE_PEAK_TFLOPS if dom.endswith("_fp8") else BF16_DENSE_PEAK_TFLOPS     # the dense MFMA peak of the kernel's operand type
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 1), "peak": kpeak,
                    "unit": "TFLOP/s", "frac": round(achieved / kpeak, 4), "traffic": traffic,
                    "traffic_unit": "HBM-side bytes per launch", "traffic_source": traffic_src,
                    "launches": n * stride, "launches_timed": n, "event_stride": stride,
                    "avg_launch_us": round(tsec / n * 1e6, 2),
                    "algorithmic_flops_per_launch": fl / n, "share_of_step_time": round(tsec * stride / (serial_dt or dt), 3),
                    "how": ("HIP events on the launch stream around every event_stride-th launch of the kernel inside the SERIAL leg (`serial`: the same "
                            "steps, one prompt group at a time): with two groups in flight a launch's event interval includes the other stream's "
                            "kernels, so per-kernel durations are only defined on the serial schedule; share_of_step_time is of the serial step")
                           if serial_dt else "HIP events on the launch stream around every event_stride-th launch of the kernel inside the timed steps"}
        images = world * G * args.steps
        # algorithmic FLOPs per sampled+scored image: 10*2*2.219 + 2.51 + 0.38 TFLOP at config 2;
        # SD3.5-large 1024^2 (config 4 shapes): 30.02 TFLOP per sample-forward, VAE x4 pixels
        per_image_tflop = (10 * 2 * 30.02 + 4 * 2.51 + 0.38) if c4 else (10 * 2 * 2.219 + 2.51 + (0.30 if c3 else 0.38))
        mixed_peak_s = None
        if c5:      # Qwen-Image at 1024^2: 4096 packed positions + the text tokens; Linears on the fp8 MFMA, attention / VAE / DINO on bf16
            from adv_grpo_amd.qwen_mmdit import flops_per_sample_forward
            qc, n_img = pipe.transformer.cfg, (RES // 16) ** 2
            f_fwd = flops_per_sample_forward(qc, n_img, C5_TEXT_TOKENS) / 1e12
            f_attn = 4.0 * qc.dim * (n_img + C5_TEXT_TOKENS) ** 2 * qc.num_layers / 1e12
            from adv_grpo_amd.qwen_vae import flops_decode
            f_vae = flops_decode(pipe.vae.cfg, RES // 8, RES // 8) / 1e12        # Qwen-Image's own decoder on one frame: 4.71 TFLOP at 1024^2
            per_image_tflop = 10 * 2 * f_fwd + f_vae + 0.30
            # seconds per image at the peaks of the units the work runs on: what "1.0 of the roofline" would be for this mix
            mixed_peak_s = (10 * 2 * (f_fwd - f_attn)) / FP8_DENSE_PEAK_TFLOPS + (10 * 2 * f_attn + f_vae + 0.30) / BF16_DENSE_PEAK_TFLOPS
        # the decoder on its own (bf16 MFMA / f32 accumulate; the reference decodes in fp32, TP:481 -- DESIGN 3 deviation 1)
        # and the fp32-equivalent split-bf16 mode (3 bf16 MFMA products per f32 product, f32 between the kernels) beside it
        from adv_grpo_amd import synthetic
        from adv_grpo_amd.vae import AutoencoderKLDecoder
        lat = torch.randn(G, 16, RES // 8, RES // 8, device=device).to(torch.bfloat16)
        vae_ms = {}
        for mode in ("bf16", "bf16x3"):
            if args.no_pricing and mode != pipe.vae.mode:
                continue
            if mode == pipe.vae.mode:
                dec = pipe.vae
            else:
                with synthetic.on_device(device):
                    if c5:
                        from adv_grpo_amd.qwen_vae import AutoencoderKLQwenImageDecoder
                        dec = AutoencoderKLQwenImageDecoder(synthetic.qwen_vae_decoder_weights(pipe.vae.cfg, 2468, dtype=torch.bfloat16), pipe.vae.cfg, device, mode=mode)
                    else:
                        dec = AutoencoderKLDecoder(synthetic.vae_decoder_weights(pipe.vae.cfg, 4321, fp16_checkpoint=True), pipe.vae.cfg, device, mode=mode)
            dec.decode_to_image(lat)
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(3):
                dec.decode_to_image(lat)
            torch.cuda.synchronize()
            vae_ms[mode] = (time.perf_counter() - tv) / 3 * 1e3
            del dec
        # the same fp32-equivalent decode when the checkpoint's weights are NOT exact in fp16 (every convolution then takes the three
        # split-bf16 products): the timed decoder assumes the released SD3 / SD3.5 VAE, an fp16 checkpoint upcast at TP:481
        # the TF32-class decoder (one fp16 product per f32 product in the wide 3x3 convolutions: AutoencoderKLDecoder(f16_single=True)); a priced
        # leg: the reference sets allow_tf32 = True (config/base.py:22-23, TP:537-538) but the default here stays fp32-EQUIVALENT
        if not c5 and pipe.vae.mode == "bf16x3" and not args.no_pricing:
            with synthetic.on_device(device):
                dec = AutoencoderKLDecoder(synthetic.vae_decoder_weights(pipe.vae.cfg, 4321, fp16_checkpoint=True), pipe.vae.cfg, device, mode="bf16x3",
                                           f16_single=True)
            dec.decode_to_image(lat)
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(3):
                dec.decode_to_image(lat)
            torch.cuda.synchronize()
            vae_ms["f16x1"] = (time.perf_counter() - tv) / 3 * 1e3
            del dec
        vae_ran = pipe.vae.arithmetic()["text"] if hasattr(pipe.vae, "arithmetic") else pipe.vae.mode
        if not c5 and pipe.vae.mode == "bf16x3" and not args.no_pricing:
            with synthetic.on_device(device):
                dec = AutoencoderKLDecoder(synthetic.vae_decoder_weights(pipe.vae.cfg, 4321, fp16_checkpoint=False), pipe.vae.cfg, device, mode="bf16x3")
            assert dec.arithmetic()["f16x2"] == 0
            dec.decode_to_image(lat)
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(3):
                dec.decode_to_image(lat)
            torch.cuda.synchronize()
            vae_ms["bf16x3_every_conv"] = (time.perf_counter() - tv) / 3 * 1e3
            del dec
        sdt = serial_dt or dt
        step_ms = sdt / args.steps * 1e3       # the pricing legs below are serial-schedule legs: compared with the serial step
        # the rollout with PEFT's LoRA arithmetic (side path as a K-extension of the adapted Linears, mmdit_train.py) instead
        # of LoRA merged into the bf16 weights: same step, other transformer object
        lora_ms = {"merged": round(step_ms, 2)}
        if not c3 and not c4 and not c5 and world == 1 and not args.no_pricing:
            from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
            merged_tr = pipe.transformer
            with synthetic.on_device(device):
                pipe.transformer = SD3TransformerLoRA(synthetic.mmdit_weights(merged_tr.cfg, 1234), merged_tr.cfg, device,
                                                      lora_mode="side")
            step(0)
            torch.cuda.synchronize()
            tl = time.perf_counter()
            for it in range(2):
                step(1 + it)
            torch.cuda.synchronize()
            lora_ms["side"] = round((time.perf_counter() - tl) / 2 * 1e3, 2)
            pipe.transformer = merged_tr
        # the same steps with the scorer on the launch stream (no reward future): what the overlap of scoring with the next rollout buys
        scoring = {"mode": "sync (--sync-scoring)" if score_stream is None else
                   "reward future: the group's scorer runs on its own HIP stream behind an event and is resolved one step later (TP:668,816-817,839-856); "
                   "rollouts serial, one prompt group at a time"}
        if score_stream is not None and world == 1 and not args.no_pricing:
            step(0)
            torch.cuda.synchronize()
            tl = time.perf_counter()
            for it in range(3):
                step(1 + it)
            torch.cuda.synchronize()
            sync_ms = (time.perf_counter() - tl) / 3 * 1e3
            scoring.update(ms_per_step_if_sync=round(sync_ms, 2), value_if_sync=round(G / (sync_ms * 1e-3), 3),
                           frac_of_bf16_mfma_peak_if_sync=round(per_image_tflop * G / (sync_ms * 1e-3) / BF16_DENSE_PEAK_TFLOPS, 4))
        # the same steps with the two halves of the CFG batch as two forwards on two HIP streams (one prompt group at a time still; the
        # kernels of one half run in the ragged last rounds / epilogue bursts of the other's).  Priced, not the headline: with two streams a
        # launch's HIP-event duration includes the other stream's kernels, so the per-kernel roofline is only defined for the schedule above.
        cfg_streams = None
        if not c5 and world == 1 and not args.no_pricing and not args.cfg_streams:
            pipe.cfg_two_streams = True
            keep_prof, ops.PROFILE = ops.PROFILE, None
            steps_pipelined(0, 1)
            torch.cuda.synchronize()
            tl = time.perf_counter()
            steps_pipelined(1, 3)
            torch.cuda.synchronize()
            c_ms = (time.perf_counter() - tl) / 3 * 1e3
            pipe.cfg_two_streams = False
            ops.PROFILE = keep_prof
            cfg_streams = {"ms_per_step": round(c_ms, 2), "value": round(G / (c_ms * 1e-3), 3),
                           "frac_of_bf16_mfma_peak": round(per_image_tflop * G / (c_ms * 1e-3) / BF16_DENSE_PEAK_TFLOPS, 4),
                           "note": "unconditional and conditional half of the CFG batch as two batch-8 forwards on two HIP streams, combined in the SDE step as before: "
                                   "bit-identical samples (tests/test_gpu_rollout.py); pipeline attribute cfg_two_streams"}
        # the same step with the block Linears of the MMDiT on fp8 e4m3 operands (BASELINE config 5's "fp8 MFMA path";
        # quantize.hip + gemm8p_fp8.hip).  Priced, not the headline: the reference has no fp8 arithmetic to match.
        fp8 = None
        bf16_linears = None
        if c5 and world == 1 and not args.no_pricing:     # config 5 is timed WITH the fp8 Linears it names; the bf16 Linears priced beside it
            keep8, pipe.transformer.fp8 = pipe.transformer.fp8, None
            step(0)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            step(1)
            torch.cuda.synchronize()
            b_ms = (time.perf_counter() - tb) * 1e3
            pipe.transformer.fp8 = keep8
            bf16_linears = {"ms_per_step": round(b_ms, 2), "value": round(G / (b_ms * 1e-3), 3), "fp8_speedup_over_bf16_step": round(b_ms / step_ms, 3),
                            "note": "the same step with the block Linears on bf16 operands (the eight-phase bf16 kernel): what the fp8 MFMA path buys"}
        if not c5 and not c3 and world == 1 and not args.no_pricing:
            pipe.transformer.enable_fp8()
            step(0)
            torch.cuda.synchronize()
            ops.PROFILE = []
            t8 = time.perf_counter()
            for it in range(2):
                step(1 + it)
            torch.cuda.synchronize()
            fp8_ms = (time.perf_counter() - t8) / 2 * 1e3
            prof8, ops.PROFILE = ops.PROFILE, None
            k8 = [(fl, s_.elapsed_time(e_) * 1e-3) for name, fl, s_, e_, _ in prof8 if name == "gemm8p_kernel_fp8"]
            fl8, t8s = sum(f for f, _ in k8), sum(t for _, t in k8)
            pipe.transformer.fp8 = None
            fp8 = {"ms_per_step": round(fp8_ms, 2), "value": round(G / (fp8_ms * 1e-3), 3), "speedup_vs_bf16_step": round(step_ms / fp8_ms, 3),
                   "gemm8p_kernel_fp8": {"launches_timed": len(k8), "event_stride": ops.PROFILE_STRIDE,
                                         "achieved_tflops": round(fl8 / t8s / 1e12, 1), "peak_tflops": FP8_DENSE_PEAK_TFLOPS,
                                         "frac": round(fl8 / t8s / 1e12 / FP8_DENSE_PEAK_TFLOPS, 4),
                                         "share_of_step_time": round(t8s * ops.PROFILE_STRIDE / 2 / (fp8_ms * 1e-3), 3)},
                   "note": "block Linears (QKV / out-projection / feed-forward, both streams) on e4m3 operands: per-token x per-channel f32 "
                           "scales, f32 accumulation, v_mfma_f32_16x16x128_f8f6f4; activations quantised on the fly; everything else bf16. "
                           "Velocity within 7e-2 of the fp32 oracle (measured 4.3e-2; bf16 path 1.3e-2) (tests/test_gpu_fp8.py); no reference arithmetic exists for this mode"}
        # two prompt groups in flight on two HIP streams (their rollouts are independent until the reward gather): kernels
        # of one group run in the GEMM tails / epilogue bursts of the other.  Priced, not the headline: with two streams a
        # launch's HIP-event duration includes the other stream's kernels, so the per-kernel roofline is only defined for
        # the serial schedule above.
        overlap = None
        if in_flight == 1 and not c3 and not c4 and world == 1 and not args.no_pricing:
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(2)

            def on_stream(st, it):
                torch.cuda.set_device(device)
                with torch.cuda.stream(st):
                    return step(it)

            def pair(it):             # one host thread per group: the launches of the two groups interleave from the start
                futs = [pool.submit(on_stream, st, it + k) for k, st in enumerate(streams)]
                return [f.result() for f in futs]
            torch.cuda.synchronize()
            pair(0)
            torch.cuda.synchronize()
            to = time.perf_counter()
            for it in range(2):
                pair(2 + 2 * it)
            torch.cuda.synchronize()
            ov_ms = (time.perf_counter() - to) / 4 * 1e3
            overlap = {"groups_in_flight": 2, "ms_per_step": round(ov_ms, 2), "value": round(G / (ov_ms * 1e-3), 3),
                       "effective_tflops_per_gpu": round(per_image_tflop * G / (ov_ms * 1e-3), 1),
                       "frac_of_bf16_mfma_peak": round(per_image_tflop * G / (ov_ms * 1e-3) / BF16_DENSE_PEAK_TFLOPS, 4),
                       "gemm8p_by_events_when_overlapped": "not meaningful: 0.28 of peak by per-launch HIP events, because a launch's "
                                                           "event interval then includes the other stream's kernels (LABNOTES.md 6)",
                       "note": "two independent prompt groups on two HIP streams, one host thread each; same kernels, same results"}
        res = {
            "metric": "sampled+scored images/sec (whole node), SD3.5-large 1024^2 10-step G=4 (secondary line, BASELINE config 4 shapes)"
            if c4 else ("sampled+scored images/sec (whole node), Qwen-Image MMDiT 1024^2 10-step G=8, DINO reward, fp8 Linears (secondary line, "
                        "BASELINE config 5 shapes)" if c5 else ("sampled+scored images/sec (whole node), SD3-med 512^2 10-step G=8, DINOv2-patch "
                        "adversarial reward (BASELINE config 3, rank-local work)" if c3 else "sampled+scored images/sec (whole node), SD3-med 512^2 10-step G=8 GRPO")),
            "value": round(images / dt, 3), "unit": "images/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp8" if c5 else "bf16", "data": "synthetic",
            "config": {"workload": ("BASELINE config 5 shapes: Qwen-Image MMDiT (60 blocks, 24 heads x 128, D=3072; block Linears on fp8 e4m3 operands, "
                                    "per-token x per-channel scales, f32 accumulation; attention / norms / embedders bf16) 1024x1024 = 4096 packed "
                                    f"latent positions + {C5_TEXT_TOKENS} text tokens (3584-wide states of the Qwen2.5-VL text tower on synthetic token ids), 10 steps, CFG 4.5 as u + s (t - u) "
                                    "(the rollout function's combine, PF:640-642, not QwenImagePipeline's norm-rescaled one), G=8, SDE window 2 @ noise 0.8 on "
                                    "the SD3 sigma table (shift 3), VAE decode (" + pipe.vae.mode + ") with Qwen-Image's own decoder (AutoencoderKLQwenImage on one "
                                    "frame: widths 384 / 192 / 96, per-pixel RMS norm), DINOv2-B/14 patch reward + head (RW:375-434), reward all-gather + group advantage")
                                   if c5 else (("BASELINE config 4 shapes: SD3.5-large (38 blocks, D=2432) LoRA-merged 1024x1024, 10 steps, CFG 4.5, "
                                    "G=4, SDE window 2 @ noise 0.8, VAE decode (fp32-equivalent; 3x3 convolutions: " + vae_ran + "), fp32-equivalent PickScore reward (ocr: stand-in "
                                    "recogniser -- no PaddleOCR in this image, the OCR half of the reward is a constant and its host-side recognition costs nothing here), "
                                    "reward all-gather + group advantage") if c4 else
                                   ("BASELINE config " + ("3" if c3 else "2") + ": SD3.5-medium LoRA-merged 512x512, 10 steps, CFG 4.5, G=8, "
                                    "SDE window 2 @ noise 0.8, VAE decode (fp32-equivalent; 3x3 convolutions: " + vae_ran + "), " +
                                    ("co-trained DINOv2 ViT-B/14 @ 518 patch discriminator + head as reward (RW:375-434)" if c3 else
                                     "PickScore (CLIP ViT-H/14) reward") + ", reward all-gather + group advantage")), "global_batch": world * G,
                       "transformer_batch_per_gpu": 2 * G, "parallelism": f"dp{world} (prompt groups sharded)",
                       "schedule": ("trainer default (config sample.groups_in_flight = 2, adv_grpo_amd/trainer.py sample_epoch): two prompt groups per rank rolled "
                                    "out at the same time, one HIP stream + one host thread each (a group's VAE decode on that one stream), scores as reward futures on one scoring stream, "
                                    "reward gather + group advantage per group on the main thread in group order; samples bit-identical to the serial "
                                    "schedule (tests/test_gpu_trainer.py::test_groups_in_flight_do_not_change_the_samples)")
                       if in_flight > 1 else "serial: one prompt group at a time on the launch stream, scores as reward futures"},
            "effective_tflops_per_gpu": round(per_image_tflop * images / dt / world, 1),
            "frac_of_bf16_mfma_peak": round(per_image_tflop * images / dt / world / BF16_DENSE_PEAK_TFLOPS, 4),
            **({"frac_of_mixed_mfma_peak": round(mixed_peak_s * images / dt / world, 4),
                "mixed_peak_is": "Linear FLOPs priced at the dense fp8 peak (5 PFLOP/s), attention / VAE / DINO FLOPs at the dense bf16 peak (2.5 PFLOP/s): "
                                 "time at those peaks / measured time"} if c5 else {}),
            "roofline": roofline,
            "serial": None if serial_dt is None else {
                "ms_per_step": round(serial_dt / args.steps * 1e3, 2), "value": round(images / serial_dt, 3), "steps": args.steps,
                "frac": round(mixed_peak_s * images / serial_dt / world, 4) if mixed_peak_s is not None else
                round(per_image_tflop * images / serial_dt / world / BF16_DENSE_PEAK_TFLOPS, 4),
                "frac_of": "the mixed fp8 / bf16 MFMA peak (frac_of_mixed_mfma_peak)" if mixed_peak_s is not None else "the bf16 MFMA peak",
                "note": "the same steps with one prompt group at a time on the launch stream (the headline schedule of rounds 1-5), timed right behind the "
                        "timed steps with the same barrier + synchronize bracket; `roofline` and the pricing legs (vae / scoring / cfg_two_streams / "
                        "fp8_linears / lora) belong to this schedule"},
            "vae": {"mode": pipe.vae.mode,
                    "mode_ran": vae_ran,
                    "assumption": None if c5 else "the released SD3 / SD3.5 VAE is an fp16 checkpoint, upcast by vae.to(torch.float32) (TP:447,481): a 3x3 convolution whose "
                                  "weight tensor is exact in fp16 (checked per tensor at load time) runs as two fp16 products per f32 product (f16x2), the others "
                                  "as three bf16 products (bf16x3); synthetic weights are rounded through fp16 to model that checkpoint",
                    "modes": {"bf16": "bf16 operands and activations, f32 accumulate",
                              "bf16x3": "the fp32-equivalent decoder: f32 between kernels; per 3x3 convolution either f16x2 (fp16-exact weight: activations as an "
                                        "fp16 hi+lo pair, 2 MFMA products per f32 product) or bf16x3 (split-bf16 hi+lo operands, 3 MFMA products); image within "
                                        "3e-5 of the fp32 decode either way (tests/test_gpu_vae.py)",
                              "bf16x3_every_conv": "the same decoder on weights that are NOT exact in fp16: every 3x3 convolution on three bf16 products",
                              "f16x1": "TF32-class opt-in (f16_single=True): ONE fp16 product per f32 product in the 31 wide 3x3 convolutions (the activation's "
                                       "fp16 hi half = a TF32-precision operand, weights exact, f32 accumulate, f32 between kernels); the reference runs its "
                                       "fp32 VAE with allow_tf32 = True (config/base.py:22-23, TP:537-538); image no further from the fp32 decode than a "
                                       "simulated-TF32 decode (tests/test_gpu_vae.py); a priced leg, not the timed mode"},
                    "ms_per_group_decode": {k: round(v, 2) for k, v in vae_ms.items()},
                    "value_if_weights_not_fp16_exact": round(images / (sdt + args.steps * (vae_ms["bf16x3_every_conv"] - vae_ms[pipe.vae.mode]) * 1e-3), 3)
                    if "bf16x3_every_conv" in vae_ms else None,
                    "frac_of_bf16_mfma_peak_if_weights_not_fp16_exact":
                        round(per_image_tflop * images / (sdt + args.steps * (vae_ms["bf16x3_every_conv"] - vae_ms[pipe.vae.mode]) * 1e-3) / world / BF16_DENSE_PEAK_TFLOPS, 4)
                        if "bf16x3_every_conv" in vae_ms else None,
                    "value_if_tf32_class": round(images / (sdt + args.steps * (vae_ms["f16x1"] - vae_ms[pipe.vae.mode]) * 1e-3), 3) if "f16x1" in vae_ms else None,
                    "frac_of_bf16_mfma_peak_if_tf32_class":
                        round(per_image_tflop * images / (sdt + args.steps * (vae_ms["f16x1"] - vae_ms[pipe.vae.mode]) * 1e-3) / world / BF16_DENSE_PEAK_TFLOPS, 4)
                        if "f16x1" in vae_ms else None,
                    "share_of_step_time": round(vae_ms[pipe.vae.mode] / step_ms, 4),
                    # what the headline would be with the other decoder swapped in (only the decode time changes)
                    "value_if_bf16x3": round(images / (sdt + args.steps * (vae_ms["bf16x3"] - vae_ms[pipe.vae.mode]) * 1e-3), 3)
                    if "bf16x3" in vae_ms else None,
                    "value_if_bf16": round(images / (sdt + args.steps * (vae_ms["bf16"] - vae_ms[pipe.vae.mode]) * 1e-3), 3)
                    if "bf16" in vae_ms else None},
            "scaling_diagnostics": scaling_diag,
            "host_rss_mb_rank0": host_rss_mb(),
            "clock_and_power": power.summary(),
            "overlap": overlap,
            "scoring": scoring,
            "cfg_two_streams": cfg_streams,
            "fp8_linears": fp8,
            **({"ocr": "stand-in recogniser (constant reward half): PaddleOCR is not in this image, adv_grpo_amd.ocr.OcrScorer runs with a callable that "
                       "returns a fixed string, so half of config 4's reward is a constant and the recognition phase costs nothing in this line"} if c4 else {}),
            **({"bf16_linears": bf16_linears, "text_tower": text_tower} if c5 else {}),
            "lora": None if c5 else {"mode": "merged",
                     "modes": {"merged": "W_eff = bf16(W + s B A) in the rollout and training forward (the timed configuration)",
                               "side": "PEFT's y = W x + s B (A x) as K + 192 / K + 64 extra columns of the adapted Linears: the "
                                       "product's log-prob change after the first AdamW step is within 0.2 % of the exact one "
                                       "(merged: 2.2 %; tests/test_gpu_train.py)"},
                     "ms_per_step": lora_ms,
                     "value_if_side": round(world * G / (lora_ms["side"] * 1e-3), 3) if "side" in lora_ms else None},
        }
    if ep_early is not None and rank == 0:
        res["epoch"] = ep_early
    if run_epoch and c5:                            # config 5: the rollout's model set has to go first (155 GiB for the epoch's own)
        del dino, dino_head
        del pipe, clip
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        ep = full_epoch(device, world, rank, adversarial=True, qwen=True, large=False)
        if rank == 0:
            res["epoch"] = ep
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not c3 and not c4 and not c5:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


def main_with_error_forwarding():
    """A rank that raises says so where the driver looks: rank 0 prints a JSON line with `error` (the contract's one line), every rank
    leaves bench_error_rank<r>.json beside the script's working directory's gpurun_out/ (kept by gpurun), then the exception goes on."""
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:          # noqa: BLE001 (re-raised)
        import traceback
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        info = {"error": f"{type(e).__name__}: {e}", "rank": rank, "n_gpus": world, "traceback_tail": traceback.format_exc()[-2000:]}
        try:
            os.makedirs("gpurun_out", exist_ok=True)
            with open(os.path.join("gpurun_out", f"bench_error_rank{rank}.json"), "w") as f:
                json.dump(info, f)
        except OSError:
            pass
        if rank == 0:
            print(json.dumps({"metric": "sampled+scored images/sec (whole node), SD3-med 512^2 10-step G=8 GRPO", "value": None, **info}))
        raise


if __name__ == "__main__":
    main_with_error_forwarding()
