"""GPU: the epoch loop end to end on a reduced-depth stack -- D-step epoch then G-step epoch (DINO variant), and a
G-step epoch with the PickScore reward; checks the plumbing (shapes, gate, optimizer step, parameter movement)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(reward, **over):
    from adv_grpo_amd import synthetic, vit
    from adv_grpo_amd.config.experiments import get_config
    from adv_grpo_amd.d_step import DinoHeadTrainable
    from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
    from adv_grpo_amd.model_configs import ClipConfig, DinoConfig, MMDiTConfig, VaeConfig
    from adv_grpo_amd.pickscore_scorer import PickScoreScorer
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.trainer import SyntheticData, Trainer
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    cfg = get_config("dino_cotrain_sd3_patch_fast" if reward == "dino" else "pickscore_cotrain_sd3_fast", gpu_number=1)
    cfg.resolution = 256
    cfg.sample.num_steps = 4
    cfg.sample.num_image_per_prompt = cfg.sample.mini_num_image_per_prompt = 2
    cfg.sample.num_batches_per_epoch = 2
    cfg.train.gradient_accumulation_steps = 1
    cfg.update(over)
    mcfg = MMDiTConfig(num_layers=2, num_heads=4, joint_attention_dim=256, pooled_projection_dim=128,
                       pos_embed_max_size=96, dual_attention_layers=(0,))
    tr = SD3TransformerLoRA({k: v.to(torch.bfloat16) for k, v in synthetic.mmdit_weights(mcfg, 1).items()}, mcfg, "cuda")
    vae = AutoencoderKLDecoder({k: v.to(torch.bfloat16) for k, v in synthetic.vae_decoder_weights(VaeConfig(), 2).items()},
                               VaeConfig(), "cuda")
    head = None
    if reward == "dino":
        dc = DinoConfig(layers=2)
        scorer = vit.DinoV2({k: v.to(torch.bfloat16) for k, v in synthetic.dino_weights(dc, 3).items()}, dc, "cuda")
        head = DinoHeadTrainable(device="cuda", seed=0)
    else:
        cc = ClipConfig(v_layers=2, t_layers=2)
        scorer = PickScoreScorer("cuda", model_sd=synthetic.clip_weights(cc, 4), clip_cfg=cc)
    data = SyntheticData(n_prompts=100, n_tokens=21, ctx_dim=256, pooled_dim=128, resolution=256)
    return Trainer(cfg, SD3Pipeline(tr, vae, "cuda"), data, scorer, head), tr, head


def test_dino_variant_d_then_g():
    tr_, model, head = _build("dino", d_times=2)
    h0, p0 = head.params.clone(), model.params.clone()
    a = tr_.run_epoch()                       # epoch 0: (0+1) % 2 != 0 -> D-step
    assert a["phase"] == "D" and torch.isfinite(torch.tensor(a["train/d_loss"]))
    assert not torch.equal(head.params, h0) and torch.equal(model.params, p0)
    b = tr_.run_epoch()                       # epoch 1: (1+1) % 2 == 0 -> G-step
    assert b["phase"] == "G" and tr_.global_step >= 2
    assert not torch.equal(model.params, p0) and torch.isfinite(model.params).all()
    assert (model.grads == 0).all()


def test_pickscore_variant_g_step():
    tr_, model, _ = _build("pickscore", train_d=False)
    p0 = model.params.clone()
    b = tr_.run_epoch()
    assert b["phase"] == "G" and not torch.equal(model.params, p0) and torch.isfinite(model.params).all()


def test_pickscore_variant_d_step_gate():
    """mean(reference reward) < mean(generated reward) -> D-step on the CLIP scorer's last layer (TP:1025-1037)."""
    tr_, model, _ = _build("pickscore", train_d=True)
    w0 = tr_.scorer.model.v_enc.layers[-1]["fc1.w"].clone()
    p0 = model.params.clone()
    phases = [tr_.run_epoch()["phase"] for _ in range(2)]
    assert set(phases) <= {"D", "G"}
    if "D" in phases:
        assert not torch.equal(tr_.scorer.model.v_enc.layers[-1]["fc1.w"], w0)
    if "G" in phases:
        assert not torch.equal(model.params, p0)


def test_eval_loop_is_deterministic_swaps_ema_and_checkpoint_round_trips(tmp_path):
    """eval() (TP:269-382) + save_ckpt (TP:389-398) + lora_path reload (TP:506-509) on the reduced stack."""
    from adv_grpo_amd import checkpoint
    tr_, model, _ = _build("pickscore", train_d=False, save_dir=str(tmp_path))
    tr_.cfg.sample.eval_num_steps = 6
    tr_.cfg.sample.test_batch_size = 3
    tr_.run_epoch()                                   # moves the LoRA parameters and creates / updates the EMA copy
    model.ema_step(7)
    live = model.params.clone()
    a = tr_.evaluate(eval_reward_fn={"pickscore_cotrain": 1})
    b = tr_.evaluate(eval_reward_fn={"pickscore_cotrain": 1})
    assert set(a) == set(b) == {"eval_reward_pickscore_cotrain", "eval_reward_avg"}
    # fixed seed-0 latents and noise 0: repeatable up to the order of the f64 atomic sums in the VAE's GroupNorm statistics
    assert all(v == v and abs(v - b[k]) <= 5e-3 * abs(v) for k, v in a.items())
    assert torch.equal(model.params, live)            # EMA swapped back (copy_temp_to)
    # checkpoint: PEFT layout, EMA weights written; reloading them reproduces the EMA model's prediction
    path = tr_.save_checkpoint()
    state, cfg = checkpoint.load_lora(path)
    assert cfg["r"] == 32 and cfg["lora_alpha"] == 64
    ema_state = {k: v for k, v in state.items()}
    key = "transformer_blocks.0.attn.to_q.lora_A.weight"
    from adv_grpo_amd.mmdit_train import RANK
    assert torch.equal(ema_state[key].cuda(), model.A_view(model.adapters[key[:-len(".lora_A.weight")]], model.ema)[:RANK])
    x = torch.randn(2, 16, 32, 32, device="cuda").to(torch.bfloat16)
    t = torch.tensor([500.0, 500.0], device="cuda")
    ctx = torch.randn(2, 21, 256, device="cuda").to(torch.bfloat16)
    pooled = torch.randn(2, 128, device="cuda").to(torch.bfloat16)
    model.load_lora_state(ema_state)
    y1 = model(x, t, ctx, pooled)[0].clone()
    model.load_lora_state({k: v.clone() for k, v in state.items()})
    assert torch.equal(model(x, t, ctx, pooled)[0], y1)
