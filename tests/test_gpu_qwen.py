"""GPU parity of BASELINE config 5's transformer (Qwen-Image MMDiT on the gfx950 kernels) against the fp32 torch oracle.

The oracle (oracle/qwen_mmdit.py) is PARITY UNPINNED w.r.t. diffusers (absent from the image; the reference itself has no
Qwen-Image code beyond README.md:75 / config/grpo.py:324,330), so what is checked is that the HIP path computes the function
the restated architecture defines.  Tolerances: the bf16 HIP path against the fp32 oracle must stay within 2x of what torch's
own bf16 execution of the SAME oracle code does (as for the SD3 model, tests/test_gpu_mmdit.py); the fp8 mode has no reference
arithmetic and is bounded as a stated property (DESIGN.md deviation list)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


# ---------------------------------------------------------------- head-dim-128 attention
def _attn_ref(q, k, v, H):
    B, Sq, HD = q.shape
    D = HD // H
    qh, kh, vh = (t.float().view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    s = (qh @ kh.transpose(-1, -2)) * D ** -0.5
    lse2 = torch.logsumexp(s, dim=-1) * 1.4426950408889634
    o = torch.softmax(s, dim=-1) @ vh
    return o.transpose(1, 2).reshape(B, Sq, HD), lse2


@pytest.mark.parametrize("B,H,Sq,Skv", [(2, 3, 4224, 4224), (1, 2, 333, 333), (2, 1, 64, 64), (1, 2, 1, 77), (1, 1, 700, 129),
                                        (1, 4, 256, 4301)])
def test_attention_d128_vs_torch(B, H, Sq, Skv):
    """The packed-QKV view convention of the MMDiT (q | k | v column slices of one buffer), aligned and ragged lengths, one
    query, fewer keys than a tile, more than one query block."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(Sq * 7 + Skv)
    D = 128
    S = max(Sq, Skv)
    qkv = torch.randn(B, S, 3 * H * D, device="cuda", generator=g).to(bf16)
    q, k, v = qkv[:, :Sq, :H * D], qkv[:, :Skv, H * D:2 * H * D], qkv[:, :Skv, 2 * H * D:]
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device="cuda")
    o = ops.attention(q, k, v, H, lse=lse)
    ref, lse_ref = _attn_ref(q, k, v, H)
    err = (o.float() - ref).abs().max().item()
    assert err < 2e-2, err
    assert _rel(o, ref) < 6e-3, _rel(o, ref)
    assert (lse - lse_ref).abs().max().item() < 2e-2


def test_attention_d128_is_bitwise_reproducible_and_takes_peaked_rows():
    """Scores with a spread of ~ +-40 (far from the N(0,1) logits of random weights: rows dominated by a few keys), twice."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    B, H, S, D = 1, 2, 1000, 128
    q = (torch.randn(B, S, H * D, device="cuda", generator=g) * 3).to(bf16)
    k = (torch.randn(B, S, H * D, device="cuda", generator=g) * 3).to(bf16)
    v = torch.randn(B, S, H * D, device="cuda", generator=g).to(bf16)
    o1 = ops.attention(q, k, v, H)
    o2 = ops.attention(q, k, v, H)
    assert torch.equal(o1, o2)
    ref, _ = _attn_ref(q, k, v, H)
    assert (o1.float() - ref).abs().max().item() < 3e-2


def test_attention_d128_scores_far_outside_the_first_tiles_window():
    """Keys of the first tile score ~ -300 below the later ones (2^-430 relative to the reference maximum of tile 0 is outside
    the f32 range the never-rescaled probabilities live in): the workgroup must notice (row sum) and take the running-maximum
    loop; and rows whose first tile dominates by the same margin must come out exact as well."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    B, H, S, D = 1, 1, 512, 128
    q = torch.randn(B, S, D, device="cuda", generator=g).to(bf16)
    k = torch.randn(B, S, D, device="cuda", generator=g).to(bf16)
    v = torch.randn(B, S, D, device="cuda", generator=g).to(bf16)
    k[:, 64:] += 30.0 * q[:, :1].sign()              # query 0 (and its look-alikes) scores the later keys far higher
    q2 = q.clone(); q2[:, 1:] = q[:, :1] * (torch.rand(1, S - 1, 1, device="cuda", generator=g) + 0.5).to(bf16)
    for qq in (q, q2):
        o = ops.attention(qq, k, v, H)
        ref, _ = _attn_ref(qq, k, v, H)
        assert torch.isfinite(o.float()).all()
        assert (o.float() - ref).abs().max().item() < 3e-2


# ---------------------------------------------------------------- QK-norm + rotary
@pytest.mark.parametrize("hd,H,Ni,Nt,B", [(128, 24, 100, 13, 2), (128, 4, 64, 7, 3), (64, 6, 50, 5, 2)])
def test_qk_norm_rope_vs_oracle(hd, H, Ni, Nt, B):
    from adv_grpo_amd import ops
    from oracle import qwen_mmdit as o
    g = torch.Generator(device="cuda").manual_seed(hd + Ni)
    S, D = Ni + Nt, H * hd
    qkv = (torch.randn(B * S, 3 * D, device="cuda", generator=g) * 2).to(bf16)
    w_img = (1 + 0.2 * torch.randn(2, hd, device="cuda", generator=g)).to(bf16)
    w_txt = (1 + 0.2 * torch.randn(2, hd, device="cuda", generator=g)).to(bf16)
    ang = torch.rand(S, hd // 2, device="cuda", generator=g) * 6.28
    rope = torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).reshape(S, hd).contiguous()
    freqs = torch.polar(torch.ones_like(ang), ang)
    ref = qkv.clone().view(B, S, 3, H, hd)
    for part, (wi, wt) in enumerate(((w_img[0], w_txt[0]), (w_img[1], w_txt[1]))):     # q, k
        xi = o.apply_rope(o._rms(ref[:, :Ni, part], wi), freqs[:Ni])
        xt = o.apply_rope(o._rms(ref[:, Ni:, part], wt), freqs[Ni:])
        ref[:, :Ni, part], ref[:, Ni:, part] = xi, xt
    rs = torch.empty(B * S, 2 * H, dtype=torch.float32, device="cuda")
    out = ops.qk_norm_rope(qkv.clone(), S, Ni, 2 * H, hd, w_img, w_txt, H, rope=rope, rs_out=rs)
    out = out.view(B, S, 3, H, hd)
    assert torch.equal(out[:, :, 2], ref[:, :, 2])                      # v untouched
    d = (out[:, :, :2].float() - ref[:, :, :2].float()).abs()
    # one bf16 rounding after the rotation on both sides; the f32 products may contract differently: <= 1 bf16 ulp of |y| ~ 4
    assert d.max().item() <= 2 ** -5 and d.mean().item() < 2e-4, (d.max().item(), d.mean().item())
    x = qkv.view(B, S, 3, H, hd)[:, :, :2].float()
    rs_ref = torch.rsqrt(x.pow(2).mean(-1) + 1e-6).reshape(B * S, 2 * H)
    assert torch.allclose(rs, rs_ref, rtol=1e-5, atol=1e-6)
    # norm only (rope=None) equals the SD3 RMSNorm step
    out2 = ops.qk_norm_rope(qkv.clone(), S, Ni, 2 * H, hd, w_img, w_txt, H).view(B, S, 3, H, hd)
    ref2 = qkv.clone().view(B, S, 3, H, hd)
    for part in range(2):
        ref2[:, :Ni, part], ref2[:, Ni:, part] = o._rms(ref2[:, :Ni, part], w_img[part]), o._rms(ref2[:, Ni:, part], w_txt[part])
    d2 = (out2.float() - ref2.float()).abs()         # (1/rms from another summation order: a last-bit flip of a bf16 rounding, rarely)
    assert d2.max().item() <= 2 ** -5 and (d2 > 0).float().mean().item() < 2e-3, (d2.max().item(), (d2 > 0).float().mean().item())


# ---------------------------------------------------------------- the model
def _inputs(cfg, B, hw, Nt, seed):
    g = torch.Generator().manual_seed(seed + 1)
    lat = torch.randn(B, 16, hw, hw, generator=g).to(bf16)
    ctx = torch.randn(B, Nt, cfg.joint_attention_dim, generator=g).to(bf16)
    return lat, ctx


def _run(cfg, B, hw, Nt, seed, on_device=False, fp8=True, torch_bf16=True):
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.qwen_mmdit import QwenImageTransformer2DModel
    from oracle import qwen_mmdit as o

    def weights():
        if on_device:
            with synthetic.on_device("cuda"):
                return synthetic.qwen_mmdit_weights(cfg, seed, dtype=bf16)
        return {k: v.to(bf16) for k, v in synthetic.qwen_mmdit_weights(cfg, seed).items()}
    lat, ctx = _inputs(cfg, B, hw, Nt, seed)
    t = torch.full((B,), 913.3488, dtype=torch.float32)
    model = QwenImageTransformer2DModel(weights(), cfg, "cuda")
    (out,), inter = model(lat.cuda(), t.cuda(), ctx.cuda(), None, return_intermediates=True)
    out8 = None
    if fp8:
        model.enable_fp8()
        out8 = model(lat.cuda(), t.cuda(), ctx.cuda(), None)[0]
    del model
    torch.cuda.empty_cache()
    Wb = weights()
    tb = None
    if torch_bf16:
        Wbc = {k: v.cuda() for k, v in Wb.items()}
        tb = o.qwen_forward(Wbc, cfg, lat.cuda(), t.cuda() / 1000, ctx.cuda())
        del Wbc
    W32 = {k: v.float().cuda() for k, v in Wb.items()}
    del Wb
    ref, rinter = o.qwen_forward(W32, cfg, lat.float().cuda(), t.cuda() / 1000, ctx.float().cuda(), return_intermediates=True)
    return out, out8, ref, tb, inter, rinter


def test_qwen_small_config_every_block():
    """4 blocks of 4 heads x 128 (D = 512), 16 x 16 packed positions + 29 text tokens: every block's two streams."""
    from oracle.qwen_mmdit import QwenMMDiTConfig
    cfg = QwenMMDiTConfig(num_layers=4, num_heads=4, joint_attention_dim=256)
    out, out8, ref, tb, inter, rinter = _run(cfg, B=3, hw=32, Nt=29, seed=11)
    for k in ("x0", "c0", "temb", "x1", "c1", "x2", "c2", "x3", "x4", "c4"):
        assert _rel(inter[k], rinter[k]) < 2e-2, (k, _rel(inter[k], rinter[k]))
    e_hip, e_torch, e8 = _rel(out, ref), _rel(tb, ref), _rel(out8, ref)
    print("qwen small: rel err hip", e_hip, "torch-bf16", e_torch, "fp8", e8)
    assert e_hip < max(2 * e_torch, 1e-2), (e_hip, e_torch)
    assert e8 < 8e-2, e8


def test_qwen_per_sample_timesteps_and_shared_timestep_agree():
    """The replay forward of a G-step hands one timestep per sample (TP:246-251); the rollout one shared timestep (a stride-0
    expand): the two modulation paths (B rows / one row) must give the same bits."""
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.qwen_mmdit import QwenImageTransformer2DModel
    from oracle.qwen_mmdit import QwenMMDiTConfig
    cfg = QwenMMDiTConfig(num_layers=2, num_heads=2, joint_attention_dim=128)
    model = QwenImageTransformer2DModel({k: v.to(bf16) for k, v in synthetic.qwen_mmdit_weights(cfg, 3).items()}, cfg, "cuda")
    lat, ctx = _inputs(cfg, 4, 16, 9, 2)
    t1 = torch.tensor(500.0, device="cuda")
    a = model(lat.cuda(), t1.expand(4), ctx.cuda())[0]
    b = model(lat.cuda(), t1.repeat(4), ctx.cuda())[0]
    assert torch.equal(a, b)
    pre = model.precompute_mods(torch.tensor([900.0, 500.0], device="cuda"))
    c = model(lat.cuda(), t1.expand(4), ctx.cuda(), mods=pre[1], context=model.embed_context(ctx.cuda()))[0]
    assert torch.equal(a, c)


def test_qwen_full_width_at_1024():
    """The real width (24 heads x 128 = 3072, 3584-wide text stream) at BASELINE config 5's resolution -- 1024 x 1024 = 4096
    packed positions + 128 text tokens, CFG pair -- on a reduced depth: the shapes that dispatch the eight-phase GEMM with
    K = 3072 / 12288 and the head-dim-128 attention at S = 4224; bf16 and fp8 Linears."""
    from oracle.qwen_mmdit import QwenMMDiTConfig
    cfg = QwenMMDiTConfig(num_layers=3)
    out, out8, ref, tb, inter, rinter = _run(cfg, B=2, hw=128, Nt=128, seed=31, on_device=True)
    assert out.shape == (2, 16, 128, 128)
    e_hip, e_torch, e8 = _rel(out, ref), _rel(tb, ref), _rel(out8, ref)
    print("qwen width 3072 @1024^2, 3 blocks: rel err hip", e_hip, "torch-bf16", e_torch, "fp8", e8)
    assert e_hip < max(2 * e_torch, 2e-2), (e_hip, e_torch)
    assert _rel(inter["x3"], rinter["x3"]) < 3e-2 and _rel(inter["c3"], rinter["c3"]) < 3e-2
    assert e8 < 8e-2, e8


def test_qwen_ragged_text_length_and_non_square_latents():
    """77 text tokens (S = 1101, ragged key tiles) on a 48 x 80 latent grid (24 x 40 packed positions: different rotary extents
    on the two axes)."""
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.qwen_mmdit import QwenImageTransformer2DModel
    from oracle import qwen_mmdit as o
    cfg = o.QwenMMDiTConfig(num_layers=2, num_heads=4, joint_attention_dim=256)
    Wb = {k: v.to(bf16) for k, v in synthetic.qwen_mmdit_weights(cfg, 41).items()}
    g = torch.Generator().manual_seed(42)
    lat = torch.randn(2, 16, 48, 80, generator=g).to(bf16)
    ctx = torch.randn(2, 77, cfg.joint_attention_dim, generator=g).to(bf16)
    t = torch.full((2,), 602.151, dtype=torch.float32)
    model = QwenImageTransformer2DModel(dict(Wb), cfg, "cuda")
    out = model(lat.cuda(), t.cuda(), ctx.cuda())[0]
    ref = o.qwen_forward({k: v.float().cuda() for k, v in Wb.items()}, cfg, lat.float().cuda(), t.cuda() / 1000, ctx.float().cuda())
    tb = o.qwen_forward({k: v.cuda() for k, v in Wb.items()}, cfg, lat.cuda(), t.cuda() / 1000, ctx.cuda())
    assert _rel(out, ref) < max(2 * _rel(tb, ref), 1e-2), (_rel(out, ref), _rel(tb, ref))


def test_qwen_full_depth_at_1024():
    """BASELINE config 5's transformer at FULL size: 60 blocks, D = 3072, 20.4 B parameters, 1024 x 1024 (4096 packed positions)
    + 128 text tokens, CFG pair, against the fp32 oracle; bf16 Linears and the fp8 Linears the config names."""
    from oracle.qwen_mmdit import QwenMMDiTConfig
    cfg = QwenMMDiTConfig()
    out, out8, ref, tb, inter, rinter = _run(cfg, B=2, hw=128, Nt=128, seed=33, on_device=True, torch_bf16=False)
    assert out.shape == (2, 16, 128, 128) and torch.isfinite(out.float()).all() and torch.isfinite(out8.float()).all()
    e_hip, e8 = _rel(out, ref), _rel(out8, ref)
    print("qwen-image 60 blocks @1024^2: rel err bf16", e_hip, "fp8", e8)
    assert e_hip < 4e-2, e_hip
    for k in ("x1", "x30", "x60"):
        assert _rel(inter[k], rinter[k]) < 6e-2, (k, _rel(inter[k], rinter[k]))
    assert e8 < 1.5e-1, e8


def test_qwen_rollout_through_the_sd3_rollout_function():
    """pipeline_with_logprob_random drives the Qwen-Image transformer unchanged (same call signature): a 4-step rollout with a
    2-step SDE window at 256 x 256, replayed with the injected noise -> the same latents, bit for bit; fp8 Linears."""
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
    from adv_grpo_amd.model_configs import VaeConfig
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.qwen_mmdit import QwenImageTransformer2DModel
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    from oracle.qwen_mmdit import QwenMMDiTConfig
    cfg = QwenMMDiTConfig(num_layers=2, num_heads=4, joint_attention_dim=256)
    tr = QwenImageTransformer2DModel({k: v.to(bf16) for k, v in synthetic.qwen_mmdit_weights(cfg, 3).items()}, cfg, "cuda")
    tr.enable_fp8()
    vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 4321), VaeConfig(), "cuda", mode="bf16")
    pipe = SD3Pipeline(tr, vae, "cuda")
    g = torch.Generator().manual_seed(8)
    pe, npe = (torch.randn(1, 20, cfg.joint_attention_dim, generator=g).to(bf16).cuda() for _ in range(2))
    pooled = torch.zeros(1, 8, dtype=bf16, device="cuda")
    kw = dict(prompt_embeds=pe, pooled_prompt_embeds=pooled, negative_prompt_embeds=npe, negative_pooled_prompt_embeds=pooled,
              num_inference_steps=4, guidance_scale=4.0, height=256, width=256, noise_level=0.8, mini_num_image_per_prompt=2,
              train_num_steps=2, process_index=0, sample_num_steps=4, random_timestep=0)
    img, lats, lps, tss = pipeline_with_logprob_random(pipe, seed=77, **kw)
    assert img.shape == (2, 3, 256, 256) and len(lats) == 3 and len(lps) == 2
    assert all(torch.isfinite(lp).all() for lp in lps)
    img2, lats2, lps2, _ = pipeline_with_logprob_random(pipe, seed=77, **kw)
    assert all(torch.equal(a, b) for a, b in zip(lats, lats2)) and all(torch.equal(a, b) for a, b in zip(lps, lps2))
