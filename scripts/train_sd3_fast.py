#!/usr/bin/env python3
"""Launcher of the Adv-GRPO epoch loop (stand-in for scripts/train_sd3_fast_{pickscore,dino_patch}.py upstream):

  python scripts/train_sd3_fast.py --config config/grpo.py:dino_cotrain_sd3_patch_fast [--epochs N] [--layers L]
  python scripts/train_sd3_fast.py --config config/grpo.py:dino_cotrain_sd3_patch_fast --model Qwen/Qwen-Image --resolution 1024 \
         --linear-dtype fp8 --images-per-prompt 8 --batches 1          (BASELINE config 5)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_sd3_fast.py --config ...

  python scripts/train_sd3_fast.py --config ... --weights <local snapshot of config.pretrained.model> \
         --pickscore-weights <PickScore_v1 dir> | --dino-weights <timm vit_base_patch14_dinov2 dir>      (real checkpoints, adv_grpo_amd/hub.py)

No checkpoints / tokenizers / reference images exist on this platform: without --weights the models get seeded synthetic weights of the
real architecture and data comes from trainer.SyntheticData.  With --weights every component's config.json and every tensor
name / shape is validated against the architecture before anything is loaded (a mismatch is an error, hub.HubError)."""
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
    ap.add_argument("--cfg-streams", action="store_true",
                    help="sample.cfg_two_streams: the unconditional and conditional halves of every rollout forward on two HIP streams (same samples)")
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
    ap.add_argument("--weights", default=None,
                    help="local Hugging Face snapshot directory of config.pretrained.model (model_index.json, transformer/, vae/, ...): what "
                         "StableDiffusion3Pipeline.from_pretrained (TP:447-449) would have downloaded; validated and loaded by adv_grpo_amd/hub.py")
    ap.add_argument("--pickscore-weights", default=None, help="local directory of yuvalkirstain/PickScore_v1 (a transformers CLIPModel; pickscore_scorer.py:8-14)")
    ap.add_argument("--dino-weights", default=None, help="local timm directory of vit_base_patch14_dinov2.lvd142m (TD:589)")
    args = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    from adv_grpo_amd.config.experiments import parse_config_flag
    cfg = parse_config_flag(args.config, gpu_number=world)
    from adv_grpo_amd import rewards as _rw                      # (before any device is touched: a refused experiment costs nothing)
    for rname in list(cfg.reward_fn.keys()):
        if rname in _rw._OUT_OF_SCOPE:
            raise SystemExit(f"{args.config}: reward '{rname}': " + _rw._WHY_NOT.get(rname, "outside the accelerated hot path (SURVEY.md 2.1 row 5)"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    from adv_grpo_amd import synthetic, vit
    from adv_grpo_amd.d_step import DinoHeadTrainable
    from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
    from adv_grpo_amd.model_configs import ClipConfig, DinoConfig, MMDiTConfig, VaeConfig
    from adv_grpo_amd.pickscore_scorer import PickScoreScorer
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.trainer import SyntheticData, Trainer
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    if args.images_per_prompt:
        cfg.sample.num_image_per_prompt = args.images_per_prompt
        cfg.sample.mini_num_image_per_prompt = min(cfg.sample.mini_num_image_per_prompt, args.images_per_prompt)     # G = 4 (config 4)
    if args.groups_in_flight:
        cfg.sample.groups_in_flight = args.groups_in_flight
    if args.cfg_streams:
        cfg.sample.cfg_two_streams = True
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
    snap = None
    if args.weights is None and os.path.isdir(str(cfg.pretrained.model)):          # from_pretrained(<local directory>), TP:447-449
        args.weights = str(cfg.pretrained.model)
    if args.weights:
        from adv_grpo_amd import hub
        snap = hub.load_pipeline(args.weights)
        if args.weights == str(cfg.pretrained.model):
            qwen = snap["kind"] == "qwen"
        if (snap["kind"] == "qwen") != qwen:
            raise SystemExit(f"--weights {args.weights} holds a {snap['kind']} pipeline but config.pretrained.model is {cfg.pretrained.model!r}")
        if args.layers is not None:
            raise SystemExit("--layers reduces the depth of a synthetic model; a checkpoint is loaded as it is")
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
            tw, qcfg = snap["transformer"] if snap else (synthetic.qwen_mmdit_weights(qcfg, 4242, dtype=torch.bfloat16), qcfg)
            vw, vcfg = snap["vae"] if snap else (synthetic.qwen_vae_decoder_weights(QwenVaeConfig(), 2468, dtype=torch.bfloat16), QwenVaeConfig())
            tr = QwenImageTransformerLoRA(tw, qcfg, device, seed=cfg.seed)
            vae = AutoencoderKLQwenImageDecoder(vw, vcfg, device, mode=args.vae_mode)
        else:
            tw, mcfg = snap["transformer"] if snap else (synthetic.mmdit_weights(mcfg, 1234), mcfg)
            vw, vcfg = snap["vae"] if snap else (synthetic.vae_decoder_weights(VaeConfig(), 4321, fp16_checkpoint=True), VaeConfig())
            tr = SD3TransformerLoRA(tw, mcfg, device, seed=cfg.seed, lora_mode=args.lora_mode)
            vae = AutoencoderKLDecoder(vw, vcfg, device, mode=args.vae_mode)
        snap = None                                              # the host copies of a 2 - 20 B parameter checkpoint are not kept
        head = None
        if any(k.startswith("dino") for k in cfg.reward_fn.keys()):
            if args.dino_weights:
                from adv_grpo_amd import hub
                dw, dcfg = hub.load_timm_dinov2(args.dino_weights)
            else:
                dw, dcfg = synthetic.dino_weights(DinoConfig(), 888), DinoConfig()
            scorer = vit.DinoV2(dw, dcfg, device)
            head = DinoHeadTrainable(device=device, seed=cfg.seed)
        else:
            if args.pickscore_weights:
                from adv_grpo_amd import hub
                pw, pcfg = hub.load_pickscore(args.pickscore_weights)
            else:
                pw, pcfg = synthetic.clip_weights(ClipConfig(), 777), ClipConfig()
            scorer = PickScoreScorer(device, dtype=torch.bfloat16, model_sd=pw, clip_cfg=pcfg)
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
