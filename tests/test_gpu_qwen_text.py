"""GPU parity of config 5's prompt encoder (adv_grpo_amd/qwen_text_encoder.py: the Qwen2.5-VL language model on text-only input) against the
fp32 torch oracle, which is itself PINNED against the installed transformers (tests/test_oracle_qwen_text.py).  Tolerance: the bf16 HIP path
must stay within 2x of torch's own bf16 run of the same oracle code."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


@pytest.mark.parametrize("hd", [128, 64])
def test_rope_half_and_causal_softmax_kernels(hd):
    from adv_grpo_amd import _lib
    from oracle import qwen_text as o
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    B, T, H, KV = 2, 37, 6, 2
    cfg = o.QwenTextConfig(hidden_size=H * hd, num_heads=H, num_kv_heads=KV)
    qkv = torch.randn(B * T, (H + 2 * KV) * hd, device="cuda", generator=g).to(bf16)
    cos, sin = o.rotary_tables(cfg, T, "cuda")
    x = qkv.float().view(B, T, H + 2 * KV, hd)
    ref = x.clone()
    ref[:, :, :H + KV] = x[:, :, :H + KV] * cos[None, :, None] + o.rotate_half(x[:, :, :H + KV]) * sin[None, :, None]
    cs = torch.stack([cos[:, :hd // 2], sin[:, :hd // 2]], dim=-1).contiguous()
    got = qkv.clone()
    _lib.check(lib.advgrpo_rope_half(got.data_ptr(), got.stride(0), B * T, T, 0, H + KV, hd, cs.data_ptr(), _lib.stream_ptr()))
    d = (got.float().view(B, T, H + 2 * KV, hd) - ref).abs()
    assert d.max().item() <= 2 ** -6 and torch.equal(got.view(B, T, -1, hd)[:, :, H + KV:], qkv.view(B, T, -1, hd)[:, :, H + KV:])
    n = 128
    sc = torch.randn(5 * n, n, device="cuda", generator=g) * 3
    p16 = torch.empty(5 * n, n, dtype=bf16, device="cuda")
    _lib.check(lib.advgrpo_softmax_rows_causal(sc.data_ptr(), p16.data_ptr(), 5 * n, n, _lib.stream_ptr()))
    mask = torch.full((n, n), float("-inf"), device="cuda").triu(1).repeat(5, 1)
    want = torch.softmax(sc + mask, dim=-1)
    assert (p16.float() - want).abs().max().item() < 4e-3 and (p16.float().triu(1)[:n] == 0).all()


@pytest.mark.parametrize("B,T", [(3, 23), (2, 100)])
def test_text_encoder_small_vs_oracle(B, T):
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.model_configs import QwenTextConfig
    from adv_grpo_amd.qwen_text_encoder import Qwen25VLTextEncoder
    from oracle import qwen_text as o
    cfg = QwenTextConfig(vocab_size=500, hidden_size=512, intermediate_size=1024, num_layers=3, num_heads=4, num_kv_heads=2)
    W = {k: v.to(bf16) for k, v in synthetic.qwen_text_weights(cfg, 7).items()}
    ids = torch.randint(0, 500, (B, T), generator=torch.Generator().manual_seed(T))
    enc = Qwen25VLTextEncoder(W, cfg, "cuda")
    out = enc(ids.cuda())
    ocfg = o.QwenTextConfig(vocab_size=500, hidden_size=512, intermediate_size=1024, num_layers=3, num_heads=4, num_kv_heads=2)
    ref = o.text_model_forward({k: v.float().cuda() for k, v in W.items()}, ocfg, ids.cuda())
    tb = o.text_model_forward({k: v.cuda() for k, v in W.items()}, ocfg, ids.cuda())
    e_hip, e_torch = _rel(out, ref), _rel(tb, ref)
    print("qwen text small: rel err hip", e_hip, "torch-bf16", e_torch)
    assert e_hip < max(2 * e_torch, 1e-2), (e_hip, e_torch)
    # prompt embeds: template drop + right padding, a shorter second sample
    mask = torch.ones(B, T, dtype=torch.long)
    mask[1, T - 5:] = 0
    emb, msk = enc.encode_prompt(ids.cuda(), mask, drop_idx=8)
    remb, rmsk = o.qwen_prompt_embeds({k: v.float().cuda() for k, v in W.items()}, ocfg, ids.cuda(), mask.cuda(), drop_idx=8)
    assert emb.shape == remb.shape and torch.equal(msk, rmsk) and _rel(emb, remb) < max(2 * e_torch, 1e-2)
    assert (emb[1, T - 5 - 8:] == 0).all()


def test_text_encoder_real_width_two_layers():
    """The real width (3584, 28 query / 4 key-value heads of 128, SwiGLU 18944) on two layers and a 60-token prompt batch."""
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.model_configs import QwenTextConfig
    from adv_grpo_amd.qwen_text_encoder import Qwen25VLTextEncoder
    from oracle import qwen_text as o
    cfg = QwenTextConfig(vocab_size=2000, num_layers=2)
    with synthetic.on_device("cuda"):
        W = synthetic.qwen_text_weights(cfg, 9, dtype=bf16)
    ids = torch.randint(0, 2000, (2, 60), generator=torch.Generator().manual_seed(4)).cuda()
    out = Qwen25VLTextEncoder(W, cfg, "cuda")(ids)
    ocfg = o.QwenTextConfig(vocab_size=2000, num_layers=2)
    ref = o.text_model_forward({k: v.float() for k, v in W.items()}, ocfg, ids)
    tb = o.text_model_forward(W, ocfg, ids)
    e_hip, e_torch = _rel(out, ref), _rel(tb, ref)
    print("qwen text real width: rel err hip", e_hip, "torch-bf16", e_torch)
    assert out.shape == (2, 60, 3584) and e_hip < max(2 * e_torch, 1e-2), (e_hip, e_torch)


def test_prompt_encoder_feeds_the_rollout():
    """Config 5 end to end in small: tokens -> Qwen2.5-VL text tower (template dropped) -> the SD3 rollout function driving the Qwen-Image
    MMDiT -> Qwen-Image VAE decode: finite images and log-probs, bit-reproducible."""
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
    from adv_grpo_amd.model_configs import QwenTextConfig, QwenVaeConfig
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.qwen_mmdit import QwenImageTransformer2DModel
    from adv_grpo_amd.qwen_text_encoder import Qwen25VLTextEncoder
    from adv_grpo_amd.qwen_vae import AutoencoderKLQwenImageDecoder
    from oracle.qwen_mmdit import QwenMMDiTConfig
    tcfg = QwenTextConfig(vocab_size=400, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1)
    enc = Qwen25VLTextEncoder({k: v.to(bf16) for k, v in synthetic.qwen_text_weights(tcfg, 3).items()}, tcfg, "cuda")
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(0, 400, (2, 34 + 21), generator=g)           # (prompt, negative prompt) after the chat template
    mask = torch.ones(2, 55, dtype=torch.long)
    mask[1, 50:] = 0
    emb, msk = enc.encode_prompt(ids.cuda(), mask)
    assert emb.shape == (2, 21, 256) and msk[1].sum().item() == 16
    mcfg = QwenMMDiTConfig(num_layers=2, num_heads=4, joint_attention_dim=256)
    tr = QwenImageTransformer2DModel({k: v.to(bf16) for k, v in synthetic.qwen_mmdit_weights(mcfg, 3).items()}, mcfg, "cuda")
    vcfg = QwenVaeConfig()
    vae = AutoencoderKLQwenImageDecoder(synthetic.qwen_vae_decoder_weights(vcfg, 7, dtype=bf16), vcfg, "cuda", mode="bf16")
    pipe = SD3Pipeline(tr, vae, "cuda")
    pooled = torch.zeros(1, 8, dtype=bf16, device="cuda")
    kw = dict(prompt_embeds=emb[:1], pooled_prompt_embeds=pooled, negative_prompt_embeds=emb[1:], negative_pooled_prompt_embeds=pooled,
              num_inference_steps=4, guidance_scale=4.0, height=256, width=256, noise_level=0.8, mini_num_image_per_prompt=2,
              train_num_steps=2, process_index=0, sample_num_steps=4, random_timestep=0)
    img, lats, lps, _ = pipeline_with_logprob_random(pipe, seed=5, **kw)
    img2, lats2, lps2, _ = pipeline_with_logprob_random(pipe, seed=5, **kw)
    assert img.shape == (2, 3, 256, 256) and torch.isfinite(img).all() and all(torch.isfinite(lp).all() for lp in lps)
    assert torch.equal(img, img2) and all(torch.equal(a, b) for a, b in zip(lps, lps2))


@pytest.mark.parametrize("n,rows_per", [(64, 64), (320, 320), (512, 512)])
def test_causal_softmax_sizes(n, rows_per):
    """Row lengths from one 64-lane pass to the 512-column maximum (one wave per row, eight columns per lane)."""
    from adv_grpo_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(n)
    sc = torch.randn(2 * rows_per, n, device="cuda", generator=g) * 4
    p16 = torch.empty(2 * rows_per, n, dtype=bf16, device="cuda")
    _lib.check(lib.advgrpo_softmax_rows_causal(sc.data_ptr(), p16.data_ptr(), 2 * rows_per, n, _lib.stream_ptr()))
    mask = torch.full((n, n), float("-inf"), device="cuda").triu(1).repeat(2, 1)
    assert (p16.float() - torch.softmax(sc + mask, dim=-1)).abs().max().item() < 4e-3
    with pytest.raises(Exception):
        _lib.check(lib.advgrpo_softmax_rows_causal(sc.data_ptr(), p16.data_ptr(), 4, 576, _lib.stream_ptr()))


def test_softmax_bwd_rows_matches_autograd():
    """advgrpo_softmax_bwd_rows (the materialised attention backward of the tune_layer = -k discriminator path): P and scale * dS against
    torch autograd of softmax over the valid columns; padding rows and columns come out zero."""
    from adv_grpo_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(9)
    n, nv, blocks = 320, 257, 3
    sc = (torch.randn(blocks * n, n, device="cuda", generator=g) * 2).requires_grad_(True)
    dp = torch.randn(blocks * n, n, device="cuda", generator=g)
    p16 = torch.empty(blocks * n, n, dtype=bf16, device="cuda")
    ds16 = torch.empty_like(p16)
    _lib.check(lib.advgrpo_softmax_bwd_rows(sc.data_ptr(), dp.data_ptr(), p16.data_ptr(), ds16.data_ptr(), blocks * n, n, nv, 0.25, _lib.stream_ptr()))
    P = torch.softmax(sc[:, :nv], dim=-1)
    (P * dp[:, :nv]).sum().backward()
    valid = (torch.arange(blocks * n, device="cuda") % n) < nv
    assert (p16.float()[valid][:, :nv] - P.detach()[valid]).abs().max().item() < 4e-3
    assert (ds16.float()[valid][:, :nv] - 0.25 * sc.grad[valid][:, :nv]).abs().max().item() < 4e-3
    assert (p16[~valid] == 0).all() and (ds16[~valid] == 0).all() and (p16[:, nv:] == 0).all() and (ds16[:, nv:] == 0).all()
