#!/usr/bin/env python3
"""Launcher of the Adv-GRPO epoch loop (stand-in for scripts/train_sd3_fast_{pickscore,dino_patch}.py upstream):

  python scripts/train_sd3_fast.py --config config/grpo.py:dino_cotrain_sd3_patch_fast [--epochs N] [--layers L]
  python scripts/train_sd3_fast.py --config config/grpo.py:dino_cotrain_sd3_patch_fast --model Qwen/Qwen-Image --resolution 1024 \
         --linear-dtype fp8 --images-per-prompt 8 --batches 1          (BASELINE config 5)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_sd3_fast.py --config ...

No checkpoints / tokenizers / reference images exist on this platform: models get seeded synthetic weights of the
real architecture and data comes from trainer.SyntheticData."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")    # before the HIP runtime starts (adv_grpo_amd/__init__.py: stream -> hardware queue map)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--layers", type=int, default=None, help="reduce the MMDiT depth (smoke runs)")
    ap.add_argument("--batches", type=int, default=None, help="override sample.num_batches_per_epoch")
    ap.add_argument("--images-per-prompt", type=int, default=None,
                    help="override sample.num_image_per_prompt (the 16/8 presets need >= 2 ranks: k = 2 must divide n*b)")
    ap.add_argument("--eval", action="store_true", help="run the eval loop every eval_freq epochs (with the co-trained scorer: "
                    "the stand-alone PickScore / image-similarity scorers need real checkpoints)")
    ap.add_argument("--log", default="logs/train.jsonl")
    ap.add_argument("--groups-in-flight", type=int, default=None,
                    help="override sample.groups_in_flight (prompt groups rolled out concurrently on separate HIP streams)")
    ap.add_argument("--lora-mode", default="merged", choices=["merged", "side"],
                    help="merged: LoRA folded into the bf16 weights (default); side: PEFT's side-path arithmetic (TP:490-511)")
    ap.add_argument("--no-train-d", action="store_true", help="keep the discriminator frozen: every epoch is a G-step epoch")
    ap.add_argument("--linear-dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: the block Linears of the MMDiT forward (rollout AND replay) on e4m3 operands (BASELINE config 5's fp8 MFMA "
                         "path; needs --lora-mode merged); the backward stays bf16 (straight-through)")
    ap.add_argument("--vae-mode", default="bf16x3", choices=["bf16", "bf16x3"],
                    help="decoder arithmetic: bf16 (default) or the fp32-equivalent split-bf16 mode (the reference decodes in fp32, TP:481)")
    ap.add_argument("--model", default=None,
                    help="override config.pretrained.model (config/grpo.py:324): stabilityai/stable-diffusion-3.5-medium (default of the presets), "
                         "stabilityai/stable-diffusion-3.5-large (BASELINE config 4) or Qwen/Qwen-Image (BASELINE config 5: Qwen-Image MMDiT with "
                         "LoRA + per-block recomputation, Qwen-Image VAE decoder, 3584-wide prompt states)")
    ap.add_argument("--resolution", type=int, default=None, help="override config.resolution (config/grpo.py:330)")
    args = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    from adv_grpo_amd import synthetic, vit
    from adv_grpo_amd.config.experiments import parse_config_flag
    from adv_grpo_amd.d_step import DinoHeadTrainable
    from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
    from adv_grpo_amd.model_configs import ClipConfig, DinoConfig, MMDiTConfig, VaeConfig
    from adv_grpo_amd.pickscore_scorer import PickScoreScorer
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.trainer import SyntheticData, Trainer
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    cfg = parse_config_flag(args.config, gpu_number=world)
    if args.images_per_prompt:
        cfg.sample.num_image_per_prompt = args.images_per_prompt
        cfg.sample.mini_num_image_per_prompt = min(cfg.sample.mini_num_image_per_prompt, args.images_per_prompt)     # G = 4 (config 4)
    if args.groups_in_flight:
        cfg.sample.groups_in_flight = args.groups_in_flight
    if args.no_train_d:
        cfg.train_d = False
    if args.batches:
        cfg.sample.num_batches_per_epoch = args.batches
        cfg.train.gradient_accumulation_steps = max(1, args.batches // 2)
    if args.model:
        cfg.pretrained.model = args.model
    if args.resolution:
        cfg.resolution = args.resolution
    name = str(cfg.pretrained.model).lower()
    qwen, large = "qwen" in name, "3.5-large" in name
    if qwen and args.lora_mode != "merged":
        raise SystemExit("Qwen-Image: --lora-mode merged only")
    mcfg = MMDiTConfig(num_layers=38, num_heads=38, dual_attention_layers=(), pos_embed_max_size=192) if large else MMDiTConfig()
    if args.layers is not None:
        mcfg = MMDiTConfig(num_layers=args.layers, num_heads=mcfg.num_heads, pos_embed_max_size=mcfg.pos_embed_max_size,
                           dual_attention_layers=() if large else tuple(range(min(13, args.layers))))
    with synthetic.on_device(device):
        if qwen:
            from adv_grpo_amd.model_configs import QwenMMDiTConfig, QwenVaeConfig
            from adv_grpo_amd.qwen_mmdit_train import QwenImageTransformerLoRA
            from adv_grpo_amd.qwen_vae import AutoencoderKLQwenImageDecoder
            qcfg = QwenMMDiTConfig() if args.layers is None else QwenMMDiTConfig(num_layers=args.layers)
            tr = QwenImageTransformerLoRA(synthetic.qwen_mmdit_weights(qcfg, 4242, dtype=torch.bfloat16), qcfg, device, seed=cfg.seed)
            vae = AutoencoderKLQwenImageDecoder(synthetic.qwen_vae_decoder_weights(QwenVaeConfig(), 2468, dtype=torch.bfloat16), QwenVaeConfig(),
                                                device, mode=args.vae_mode)
        else:
            tr = SD3TransformerLoRA(synthetic.mmdit_weights(mcfg, 1234), mcfg, device, seed=cfg.seed, lora_mode=args.lora_mode)
            vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 4321, fp16_checkpoint=True), VaeConfig(), device, mode=args.vae_mode)
        head = None
        if any(k.startswith("dino") for k in cfg.reward_fn.keys()):
            scorer = vit.DinoV2(synthetic.dino_weights(DinoConfig(), 888), DinoConfig(), device)
            head = DinoHeadTrainable(device=device, seed=cfg.seed)
        else:
            scorer = PickScoreScorer(device, dtype=torch.bfloat16, model_sd=synthetic.clip_weights(ClipConfig(), 777), clip_cfg=ClipConfig())
    if args.linear_dtype == "fp8":
        tr.enable_fp8()
    # reward factories that need a backbone (adv_grpo.rewards builds them from checkpoints; none can be downloaded here): the fp32
    # scorers get synthetic weights of the real architecture, on the fp32-equivalent towers (vit_x3.py)
    from adv_grpo_amd import rewards, vit_x3
    wanted = set(cfg.reward_fn.keys()) | (set(cfg.eval_reward_fn.keys()) if args.eval and cfg.get("eval_reward_fn") else set())
    with synthetic.on_device(device):
        if "pickscore" in wanted:
            rewards.configure_pickscore(synthetic.clip_weights(ClipConfig(), 777), ClipConfig())
        if "image_similarity" in wanted:
            rewards.configure_dino(vit_x3.DinoV2X3(synthetic.dino_weights(DinoConfig(), 888), DinoConfig(), device))
    if "ocr" in wanted:
        try:
            import paddleocr  # noqa: F401  (adv_grpo/ocr.py:8-65 uses it when present)
        except ImportError:
            rewards.configure_ocr(lambda img: "")      # no recogniser in this image: every OCR reward is the empty-read score
    pipe = SD3Pipeline(tr, vae, device)
    data = SyntheticData(n_tokens=128, ctx_dim=3584, pooled_dim=8, resolution=cfg.resolution, device=device) if qwen else \
        SyntheticData(resolution=cfg.resolution, device=device)
    if cfg.train.lora_path:                                                      # TP:506-509
        from adv_grpo_amd import checkpoint
        tr.load_lora_state(checkpoint.load_lora(cfg.train.lora_path)[0])
    trainer = Trainer(cfg, pipe, data, scorer, head, rank, world, log_path=args.log)
    for _ in range(args.epochs):
        if args.eval and trainer.epoch % cfg.eval_freq == 0:                     # TP:712-713
            ev = trainer.evaluate(eval_reward_fn={trainer.reward_key: 1})
            if rank == 0:
                print(json.dumps({"epoch": trainer.epoch, **ev}))
        if trainer.epoch % cfg.save_freq == 0 and trainer.epoch > 0:             # TP:714-715
            trainer.save_checkpoint()
        info = trainer.run_epoch()
        if rank == 0:
            print(json.dumps({"epoch": trainer.epoch, **{k: (v if not hasattr(v, "item") else v.item()) for k, v in info.items()},
                              "timers_s": {k: round(v, 3) for k, v in trainer.timers.items()}}))


if __name__ == "__main__":
    main()
