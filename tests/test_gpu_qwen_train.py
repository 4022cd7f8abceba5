"""GPU parity of the update half of BASELINE config 5 (Qwen-Image MMDiT in the G-step): the head-dim-128 attention backward and
the QK-norm + rotary backward against torch autograd, then the LoRA gradients of the model against autograd through the fp32
oracle (oracle/qwen_mmdit.py, PARITY UNPINNED w.r.t. diffusers)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


def _attn_ref(q, k, v, H):
    B, Sq, HD = q.shape
    D = HD // H
    qh, kh, vh = (t.view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    return (torch.softmax((qh @ kh.transpose(-1, -2)) * D ** -0.5, dim=-1) @ vh).transpose(1, 2).reshape(B, Sq, HD)


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 333, 333), (2, 3, 64, 64), (1, 1, 700, 129), (1, 2, 31, 77), (2, 24, 1152, 1152)])
def test_attention_backward_d128_matches_autograd(B, H, Sq, Skv):
    """dq / dk / dv of the head-dim-128 attention from views of ONE packed gradient buffer (as the model calls it), ragged tiles
    on both sides; against fp32 autograd of softmax(q k^T / sqrt(d)) v on the same bf16 inputs."""
    from adv_grpo_amd import ops
    D = 128
    g = torch.Generator(device="cuda").manual_seed(Sq + Skv)
    S = max(Sq, Skv)
    qkv = torch.randn(B, S, 3 * H * D, device="cuda", generator=g).to(bf16)
    q, k, v = qkv[:, :Sq, :H * D], qkv[:, :Skv, H * D:2 * H * D], qkv[:, :Skv, 2 * H * D:]
    d_o = torch.randn(B, Sq, H * D, device="cuda", generator=g).to(bf16)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device="cuda")
    o = ops.attention(q, k, v, H, lse=lse)
    dqkv = torch.zeros(B, S, 3 * H * D, dtype=bf16, device="cuda")
    dq, dk, dv = dqkv[:, :Sq, :H * D], dqkv[:, :Skv, H * D:2 * H * D], dqkv[:, :Skv, 2 * H * D:]
    ops.attention_bwd(q, k, v, o, d_o, lse, H, dq, dk, dv)
    qf, kf, vf = (x.float().clone().requires_grad_(True) for x in (q, k, v))
    _attn_ref(qf, kf, vf, H).backward(d_o.float())
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        rel = ((got.float() - want).norm() / want.norm()).item()
        assert rel < 2e-2, (name, rel)
    # deterministic: a second launch gives the same bits
    again = torch.zeros_like(dqkv)
    ops.attention_bwd(q, k, v, o, d_o, lse, H, again[:, :Sq, :H * D], again[:, :Skv, H * D:2 * H * D], again[:, :Skv, 2 * H * D:])
    assert torch.equal(again, dqkv)


def test_attention_backward_d128_refuses_rows_beyond_32_bit_offsets():
    """The head-dim-128 backward addresses its tiles with 32-bit byte offsets from a per-(b, h) base: a view whose rows * row pitch
    reaches 2^30 elements must fail loudly (include/advgrpo.h), not wrap around."""
    from adv_grpo_amd import ops
    from adv_grpo_amd._lib import AdvGrpoError
    D, H, S = 128, 1, 64
    ld = (1 << 24) + 128                                      # 64 rows x 2^24 elements = 2^30
    big = torch.zeros(S * ld, dtype=bf16, device="cuda")      # 2 GiB
    q = big.as_strided((1, S, H * D), (S * ld, ld, 1))
    k = torch.randn(1, S, H * D, device="cuda").to(bf16)
    v = torch.randn(1, S, H * D, device="cuda").to(bf16)
    lse = torch.empty(1, H, S, dtype=torch.float32, device="cuda")
    o = ops.attention(q.contiguous(), k, v, H, lse=lse)
    d_o = torch.randn(1, S, H * D, device="cuda").to(bf16)
    dq, dk, dv = (torch.empty(1, S, H * D, dtype=bf16, device="cuda") for _ in range(3))
    with pytest.raises(AdvGrpoError, match="2\^30"):
        ops.attention_bwd(q, k, v, o, d_o, lse, H, dq, dk, dv)
    ops.attention_bwd(k, k, v, o, d_o, lse, H, dq, dk, dv)    # (the same call on ordinary views goes through)


@pytest.mark.parametrize("hd,H,Ni,Nt,B", [(128, 24, 100, 13, 2), (128, 4, 64, 7, 3), (64, 6, 50, 5, 2)])
def test_qk_norm_rope_backward_matches_autograd(hd, H, Ni, Nt, B):
    """In-place gradient of per-head RMSNorm (image / text weights) + rotary embedding on the q | k part of the joint buffer,
    against fp32 autograd of the oracle's two steps; v's gradient is left alone."""
    from adv_grpo_amd import ops
    from oracle import qwen_mmdit as o
    g = torch.Generator(device="cuda").manual_seed(hd + Ni + 1)
    S, D = Ni + Nt, H * hd
    qkv = (torch.randn(B * S, 3 * D, device="cuda", generator=g) * 2).to(bf16)
    w_img = (1 + 0.2 * torch.randn(2, hd, device="cuda", generator=g)).to(bf16)
    w_txt = (1 + 0.2 * torch.randn(2, hd, device="cuda", generator=g)).to(bf16)
    ang = torch.rand(S, hd // 2, device="cuda", generator=g) * 6.28
    rope = torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).reshape(S, hd).contiguous()
    freqs = torch.polar(torch.ones_like(ang), ang)
    dy = torch.randn(B * S, 3 * D, device="cuda", generator=g).to(bf16)
    # forward on the kernels (saved output + 1/rms), backward in place
    rs = torch.empty(B * S, 2 * H, dtype=torch.float32, device="cuda")
    y = ops.qk_norm_rope(qkv.clone(), S, Ni, 2 * H, hd, w_img, w_txt, H, rope=rope, rs_out=rs)
    dx = ops.qk_norm_rope_bwd(dy.clone(), y, rs, S, Ni, 2 * H, hd, w_img, w_txt, H, rope=rope)
    # autograd of the oracle's steps in fp32
    x = qkv.float().view(B, S, 3, H, hd).clone().requires_grad_(True)
    parts = []
    for part in range(2):
        xi = o.apply_rope(o._rms(x[:, :Ni, part], w_img[part].float()), freqs[:Ni])
        xt = o.apply_rope(o._rms(x[:, Ni:, part], w_txt[part].float()), freqs[Ni:])
        parts.append(torch.cat([xi, xt], dim=1))
    out = torch.stack(parts + [x[:, :, 2]], dim=2)
    out.backward(dy.float().view(B, S, 3, H, hd))
    got = dx.float().view(B, S, 3, H, hd)
    assert torch.equal(got[:, :, 2], dy.float().view(B, S, 3, H, hd)[:, :, 2])
    rel = ((got[:, :, :2] - x.grad[:, :, :2]).norm() / x.grad[:, :, :2].norm()).item()
    assert rel < 1e-2, rel


# ---------------------------------------------------------------- the model: LoRA gradients vs autograd through the oracle
def _cos(a, b):
    return (torch.dot(a.flatten().double(), b.flatten().double()) / (a.double().norm() * b.double().norm() + 1e-30)).item()


def _lora_init(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    D, sd = cfg.dim, {}
    for i in range(cfg.num_layers):
        for n in ("to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj", "to_add_out"):
            key = f"transformer_blocks.{i}.attn.{n}"
            sd[key + ".lora_A.weight"] = (torch.randn(32, D, generator=g) / 32).to(bf16).float()
            sd[key + ".lora_B.weight"] = (torch.randn(D, 32, generator=g) * 0.02).to(bf16).float()
    return sd


def _lora_case(cfg, seed, B, hw, Nt, per_sample_t=False, fp8=False):
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.qwen_mmdit_train import QwenImageTransformerLoRA
    from oracle import lora as o_lora
    from oracle import qwen_mmdit as o
    W = {k: v.to(bf16) for k, v in synthetic.qwen_mmdit_weights(cfg, seed).items()}
    lora = _lora_init(cfg, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    lat = torch.randn(B, 16, hw, hw, generator=g).to(bf16)
    ctx = torch.randn(B, Nt, cfg.joint_attention_dim, generator=g).to(bf16)
    t = (torch.tensor([913.3488, 700.0, 500.0, 300.0][:B]) if per_sample_t else torch.full((B,), 913.3488)).float()
    model = QwenImageTransformerLoRA(dict(W), cfg, "cuda", lora_state=lora)
    if fp8:
        model.enable_fp8()
    v, saved = model.forward_train(lat.cuda(), t.cuda(), ctx.cuda())
    (v_inf,) = model(lat.cuda(), t.cuda(), ctx.cuda())
    assert torch.equal(v, v_inf)                                              # training forward == rollout forward
    dv = torch.randn(v.shape, generator=g).to(bf16)
    model.backward(saved, dv.cuda())
    grads = model.lora_grads()
    W32 = {k: x.float().cuda() for k, x in W.items()}
    lo = {k: x.cuda().requires_grad_(True) for k, x in lora.items()}
    out = o.qwen_forward(o_lora.effective_weights(W32, lo), cfg, lat.float().cuda(), t.cuda() / 1000, ctx.float().cuda())
    assert ((v.float() - out).norm() / out.norm()).item() < (8e-2 if fp8 else 3e-2)
    (out * dv.float().cuda()).sum().backward()
    worst, worst_ratio = 1.0, 1.0
    c_min, r_lo, r_hi = (0.9, 0.8, 1.25) if fp8 else (0.97, 0.9, 1.1)
    for k, gr in grads.items():
        ref = lo[k].grad
        if ref is None or ref.norm().item() == 0:          # add_q_proj / to_add_out of the last block: nothing reads its text-stream output
            assert gr.abs().max().item() == 0, k
            continue
        c, ratio = _cos(gr, ref), (gr.norm() / ref.norm()).item()
        worst, worst_ratio = min(worst, c), max(worst_ratio, max(ratio, 1 / ratio))
        assert c > c_min and r_lo < ratio < r_hi, (k, c, ratio)
    print("qwen LoRA grads" + (" (fp8 forward, straight-through)" if fp8 else "") + ": worst cosine", worst, "worst norm ratio", worst_ratio)
    return model


@pytest.mark.parametrize("per_sample_t", [False, True])
def test_qwen_lora_backward_vs_autograd_small(per_sample_t):
    """3 blocks of 4 heads x 128, 8 x 8 packed positions + 13 text tokens: every adapter's A and B gradient (image and text stream,
    q / k through the QK-norm + rotary backward, v, both output projections) against fp32 autograd through W + s B A of the oracle;
    one shared timestep (the rollout's) and one per sample (the replay's, TP:246-251)."""
    from oracle.qwen_mmdit import QwenMMDiTConfig
    cfg = QwenMMDiTConfig(num_layers=3, num_heads=4, joint_attention_dim=256)
    model = _lora_case(cfg, 41, B=4, hw=16, Nt=13, per_sample_t=per_sample_t)
    # the optimiser step of the SD3 model's flat vectors moves the merged weights
    w0 = model.blocks[1]["qkv.w"].clone()
    model.optimizer_step(lr=1e-3)
    assert not torch.equal(model.blocks[1]["qkv.w"], w0) and model.grads.abs().max().item() == 0


def test_qwen_lora_backward_vs_autograd_full_width():
    """The real width (24 x 128 = 3072, 3584-wide text states) on two blocks at 32 x 32 packed positions + 64 text tokens."""
    from oracle.qwen_mmdit import QwenMMDiTConfig
    _lora_case(QwenMMDiTConfig(num_layers=2), 43, B=2, hw=64, Nt=64)


def test_qwen_g_step_chain_rollout_replay_loss_backward_adamw(tmp_path):
    """Config 5's loop in small: the SD3 rollout function drives the LoRA model (4 steps, SDE window 2), then every trained timestep
    goes through g_step.micro_step (compute_log_prob TP:233-267 -> GRPO loss TP:1111-1130 -> backward TP:1165): the replayed
    log-probs are the rollout's up to the bf16 cast of the stored latents (importance ratio 1 before the first update), the G-step is bitwise repeatable,
    and one clip + AdamW step (TP:1166-1171) moves the policy."""
    from adv_grpo_amd import g_step, synthetic
    from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
    from adv_grpo_amd.model_configs import QwenVaeConfig
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.qwen_mmdit_train import QwenImageTransformerLoRA
    from adv_grpo_amd.qwen_vae import AutoencoderKLQwenImageDecoder
    from oracle.qwen_mmdit import QwenMMDiTConfig
    cfg = QwenMMDiTConfig(num_layers=2, num_heads=4, joint_attention_dim=256)
    W = {k: v.to(bf16) for k, v in synthetic.qwen_mmdit_weights(cfg, 5).items()}
    model = QwenImageTransformerLoRA(W, cfg, "cuda", lora_state=_lora_init(cfg, 6))
    vcfg = QwenVaeConfig()
    vae = AutoencoderKLQwenImageDecoder(synthetic.qwen_vae_decoder_weights(vcfg, 7, dtype=bf16), vcfg, "cuda", mode="bf16")
    pipe = SD3Pipeline(model, vae, "cuda")
    g = torch.Generator().manual_seed(8)
    pe, npe = (torch.randn(1, 20, cfg.joint_attention_dim, generator=g).to(bf16).cuda() for _ in range(2))
    pooled = torch.zeros(1, 8, dtype=bf16, device="cuda")
    G, T = 4, 2
    images, lats, lps, tss = pipeline_with_logprob_random(
        pipe, prompt_embeds=pe, pooled_prompt_embeds=pooled, negative_prompt_embeds=npe, negative_pooled_prompt_embeds=pooled,
        num_inference_steps=4, guidance_scale=4.0, height=256, width=256, noise_level=0.8, mini_num_image_per_prompt=G,
        train_num_steps=T, process_index=0, sample_num_steps=4, random_timestep=0, seed=77)
    assert images.shape == (G, 3, 256, 256) and torch.isfinite(images).all()
    lat = torch.stack(lats, dim=1)
    sample = {"latents": lat[:, :-1], "next_latents": lat[:, 1:], "timesteps": torch.stack(tss, dim=1)}
    old = torch.stack(lps, dim=1)
    embeds = torch.cat([npe.repeat(G, 1, 1), pe.repeat(G, 1, 1)])                # negative first (TP:1084-1091)
    adv = torch.tensor([1.0, -0.5, 0.25, -0.75], device="cuda")
    kw = dict(guidance_scale=4.0, noise_level=0.8, adv_clip_max=5, clip_range=1e-4)
    first = int(pipe.last_random_timestep)
    for j in range(T):
        info = g_step.micro_step(model, pipe.scheduler, sample, j, embeds, None, old[:, j], adv, step_index=first + j, **kw)
        # on-policy replay: ratio == 1 up to the bf16 cast of the stored latents (the rollout's log-prob comes from the f32 sample
        # BEFORE the cast, PF:646-660, the replay's from the stored one, TP:258 -- as in the reference and the SD3 trainer tests)
        assert (info["log_prob"] - old[:, j]).abs().max().item() < 3e-4 and 0.0 <= float(info["approx_kl"]) < 1e-7, j
    g1 = model.grads.clone()
    assert torch.isfinite(g1).all() and g1.abs().max().item() > 0
    model.grads.zero_()
    for j in range(T):
        g_step.micro_step(model, pipe.scheduler, sample, j, embeds, None, old[:, j], adv, step_index=first + j, **kw)
    assert torch.equal(model.grads, g1)                                        # bitwise repeatable G-step
    model.optimizer_step(lr=1e-3)
    model.ema_step(1)
    after = g_step.micro_step(model, pipe.scheduler, sample, 0, embeds, None, old[:, 0], adv, step_index=first, **kw)
    assert not torch.equal(after["log_prob"], old[:, 0]) and torch.isfinite(after["log_prob"]).all()
    # save_ckpt (TP:389-398) in PEFT layout and the lora_path reload (TP:506-509): a fresh model with the saved adapters gives the same bits
    from adv_grpo_amd import checkpoint
    model.save_pretrained(str(tmp_path / "ckpt"), use_ema=False)
    state, meta = checkpoint.load_lora(str(tmp_path / "ckpt"))
    assert meta["r"] == 32 and meta["lora_alpha"] == 64 and len(state) == 2 * 8 * cfg.num_layers
    fresh = QwenImageTransformerLoRA({k: v.to(bf16) for k, v in synthetic.qwen_mmdit_weights(cfg, 5).items()}, cfg, "cuda")
    fresh.load_lora_state(state)
    x = lat[:2, 0]
    t = torch.full((2,), 500.0, device="cuda")
    assert torch.equal(fresh(x, t, embeds[:2])[0], model(x, t, embeds[:2])[0])
    # without the kept attention outputs (the leaner checkpoint) the gradients are the same bits
    model.grads.zero_()
    g_step.micro_step(model, pipe.scheduler, sample, 0, embeds, None, old[:, 0], adv, step_index=first, **kw)
    g_keep = model.grads.clone()
    model.grads.zero_()
    model.keep_attention = False
    g_step.micro_step(model, pipe.scheduler, sample, 0, embeds, None, old[:, 0], adv, step_index=first, **kw)
    assert torch.equal(model.grads, g_keep)


def test_qwen_lora_backward_fp8_replay_straight_through():
    """fp8 Linears (the arithmetic BASELINE config 5 names): the training forward issues the rollout's e4m3 launches (bit-identical
    output), the backward is the bf16 Linear's from the bf16 activations kept beside the e4m3 rows; its LoRA gradients against fp32
    autograd through the (unquantised) oracle -- a stated property with looser bounds, no reference arithmetic exists."""
    from oracle.qwen_mmdit import QwenMMDiTConfig
    _lora_case(QwenMMDiTConfig(num_layers=3, num_heads=4, joint_attention_dim=256), 47, B=4, hw=16, Nt=20, fp8=True)      # (the fp8 GEMM's row epilogue wants >= 16 text tokens)


@pytest.mark.parametrize("fp8", [False, True])
def test_qwen_forward_reference_is_the_adapter_free_model(fp8):
    """forward_reference (the KL term's reference policy under train.beta > 0, TP:1105-1108 disable_adapter): with non-zero adapters it
    gives the bits of a model built WITHOUT them and differs from the adapted forward; the adapted forward is untouched afterwards."""
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.model_configs import QwenMMDiTConfig
    from adv_grpo_amd.qwen_mmdit import QwenImageTransformer2DModel
    from adv_grpo_amd.qwen_mmdit_train import QwenImageTransformerLoRA
    cfg = QwenMMDiTConfig(num_layers=2, num_heads=4, joint_attention_dim=256)
    W = {k: v.to(bf16) for k, v in synthetic.qwen_mmdit_weights(cfg, 5).items()}
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(2, 16, 16, 16, generator=g).to(bf16).cuda()
    ctx = torch.randn(2, 20, cfg.joint_attention_dim, generator=g).to(bf16).cuda()
    t = torch.full((2,), 700.0).cuda()
    model = QwenImageTransformerLoRA(dict(W), cfg, "cuda", lora_state=_lora_init(cfg, 6))
    plain = QwenImageTransformer2DModel(dict(W), cfg, "cuda")
    if fp8:
        model.enable_fp8(); plain.enable_fp8()
    (v_pol,) = model(lat, t, ctx)
    v_ref = model.forward_reference(lat, t, ctx, None)
    (v_plain,) = plain(lat, t, ctx)
    assert torch.equal(v_ref, v_plain) and not torch.equal(v_ref, v_pol)
    assert torch.equal(model(lat, t, ctx)[0], v_pol)
