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
