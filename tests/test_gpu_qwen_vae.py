"""GPU parity of the Qwen-Image VAE decoder path (adv_grpo_amd/qwen_vae.py, BASELINE config 5's decode) against the fp32 torch
oracle (oracle/qwen_vae.py, itself "parity unpinned" vs diffusers)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rmsnorm_nhwc_all_widths_and_outputs():
    """The per-pixel RMS norm at the decoder's widths (384 / 192 / 96-in-128 with zero padding) in its three output forms, f32
    and bf16 input, against F.normalize * sqrt(C) * gamma (+ SiLU)."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    for C, real in ((384, 384), (192, 192), (128, 96), (64, 64)):
        for silu in (False, True):
            x = torch.randn(3, 5, 7, C, device="cuda", generator=g) * 3 + 0.2
            x[..., real:] = 0
            gamma = torch.rand(C, device="cuda", generator=g) + 0.5
            gamma[real:] = 0
            ref = torch.nn.functional.normalize(x[..., :real], dim=-1) * real ** 0.5 * gamma[:real]
            if silu:
                ref = torch.nn.functional.silu(ref)
            y3 = ops.rmsnorm_nhwc(x, gamma, real ** 0.5, silu=silu, out="x3").float()
            hi, mid, lo = y3[..., :C], y3[..., C:2 * C], y3[..., 2 * C:]
            assert torch.equal(hi, mid)
            assert (hi[..., :real] + lo[..., :real] - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
            assert (hi[..., real:] == 0).all() and (lo[..., real:] == 0).all()
            yp = ops.rmsnorm_nhwc(x, gamma, real ** 0.5, silu=silu, out="x3pair").float()
            assert torch.equal(yp[..., :C], hi) and torch.equal(yp[..., 2 * C:], lo)
            xb = x.to(torch.bfloat16)
            refb = torch.nn.functional.normalize(xb.float()[..., :real], dim=-1) * real ** 0.5 * gamma[:real]
            if silu:
                refb = torch.nn.functional.silu(refb)
            yb = ops.rmsnorm_nhwc(xb, gamma, real ** 0.5, silu=silu, out="bf16").float()
            assert ((yb[..., :real] - refb).abs() <= 2 ** -8 * refb.abs() + 1e-4).all() and (yb[..., real:] == 0).all()
    # a pixel of zeros stays zero (F.normalize's eps), no NaN
    z = torch.zeros(1, 2, 2, 192, device="cuda")
    assert (ops.rmsnorm_nhwc(z, torch.ones(192, device="cuda"), 192 ** 0.5, out="x3") == 0).all()


def test_latents_mix_matches_denormalise_then_post_quant_conv():
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(12)
    z = torch.randn(2, 16, 5, 6, device="cuda", generator=g)
    inv_std, mean = torch.rand(16, device="cuda", generator=g) + 0.3, torch.randn(16, device="cuda", generator=g)
    P, b = torch.randn(16, 16, device="cuda", generator=g) / 4, torch.randn(16, device="cuda", generator=g)
    ref = torch.einsum("oc,bchw->bhwo", P.double(), (z / inv_std.view(1, -1, 1, 1) + mean.view(1, -1, 1, 1)).double()) + b.double()
    y3 = ops.latents_mix_to_nhwc(z, 64, inv_std, mean, P, b, x3=True).float()
    assert y3.shape == (2, 5, 6, 192)
    # hi + lo carries 16 significant bits
    assert ((y3[..., :16] + y3[..., 128:144] - ref.float()).abs() <= 2 ** -15 * ref.float().abs() + 1e-6).all() and (y3[..., 16:64] == 0).all()
    yb = ops.latents_mix_to_nhwc(z.to(torch.bfloat16), 64, inv_std, mean, P, b).float()
    refb = torch.einsum("oc,bchw->bhwo", P, z.to(torch.bfloat16).float() / inv_std.view(1, -1, 1, 1) + mean.view(1, -1, 1, 1)) + b
    assert (yb[..., :16] - refb).abs().max().item() < 2e-2 and (yb[..., 16:] == 0).all()


def test_conv3x3_bf16x2_two_products_match_fp32_conv():
    """One-piece bf16 weights x bf16-pair activations (two products) against the f32 convolution of the same bf16-exact weights, at
    the decoder's widths (incl. the zero-padded 96-in-128 stage), with upsampling, bias and residual."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(13)
    for (B, H, W, Ci, Co, up) in [(2, 16, 24, 64, 384, False), (1, 12, 12, 384, 192, True), (2, 9, 11, 128, 128, False), (1, 20, 20, 192, 384, False)]:
        x = torch.randn(B, H, W, Ci, device="cuda", generator=g) * 2
        wt = (torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) / (3 * Ci ** 0.5)).to(torch.bfloat16)
        bias = torch.randn(Co, device="cuda", generator=g)
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        res = torch.randn(B, Ho, Wo, Co, device="cuda", generator=g)
        y = ops.conv3x3_f16x2(ops.split_x3(x, order=2), wt.permute(0, 2, 3, 1).reshape(Co, -1).contiguous(), bias=bias, upsample=up,
                              residual=res, bf16_pieces=True)
        xin = x.permute(0, 3, 1, 2)
        if up:
            xin = torch.nn.functional.interpolate(xin, scale_factor=2.0, mode="nearest")
        ref = torch.nn.functional.conv2d(xin.double(), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 1) + res.double()
        err = (y.double() - ref).abs().max().item()
        assert err < 3e-5 * ref.abs().max().item(), (B, H, W, Ci, Co, up, err)
        # and the three-product kernel on the same operands agrees to the same level
        y3 = ops.conv3x3_x3(ops.split_x3(x, order=2), ops.split_x3(wt.float().permute(0, 2, 3, 1).contiguous(), order=1).reshape(Co, -1),
                            bias=bias, upsample=up, residual=res)
        assert (y - y3).abs().max().item() < 3e-5 * ref.abs().max().item()


def _decode_both(cfg, seed, B, h, w, mode, **kw):
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.qwen_vae import AutoencoderKLQwenImageDecoder
    from oracle import qwen_vae as o
    W = synthetic.qwen_vae_decoder_weights(cfg, seed, dtype=torch.bfloat16)          # the released checkpoint is bf16
    lat = torch.randn(B, 16, h, w, generator=torch.Generator().manual_seed(seed + h)).to(torch.bfloat16)
    dec = AutoencoderKLQwenImageDecoder(W, cfg, "cuda", mode=mode, **kw)
    img = dec.decode_to_image(lat.cuda())
    ocfg = o.QwenVaeConfig(base_dim=cfg.base_dim, z_dim=cfg.z_dim, dim_mult=cfg.dim_mult, num_res_blocks=cfg.num_res_blocks)
    with torch.no_grad():
        ref = o.decode_to_image({k: v.float().cuda() for k, v in W.items()}, ocfg, lat.float().cuda())
    return dec, lat, img, ref


@pytest.mark.parametrize("B,h,w", [(2, 16, 16), (3, 16, 12), (1, 64, 64)])
def test_qwen_vae_decode_bf16x3_vs_fp32_oracle(B, h, w):
    """The real widths (384 / 192 / 96) at small latents, a non-square one, and the 512^2 size; mode bf16x3 against the fp32
    oracle: the SD3 decoder's bounds (mean abs <= 2e-5, max abs <= 1e-3 on the [0,1] image)."""
    from adv_grpo_amd.model_configs import QwenVaeConfig
    dec, lat, img, ref = _decode_both(QwenVaeConfig(), 21, B, h, w, "bf16x3")
    assert img.shape == (B, 3, 8 * h, 8 * w) and img.dtype == torch.float32
    err = (img - ref).abs()
    print("qwen vae x3 image err mean", err.mean().item(), "max", err.max().item(), "ref std", ref.std().item())
    assert err.mean().item() < 2e-5 and err.max().item() < 1e-3
    # two half batches on two streams: bit-identical to the single-stream decode
    if B > 1:
        dec.two_streams = False
        one = dec.decode_to_image(lat.cuda())
        torch.cuda.synchronize()
        assert torch.equal(one, img)


def test_qwen_vae_three_product_path_for_weights_that_are_not_bf16_exact():
    """bf16_weights=False forces the three split-bf16 products everywhere (what an fp32 checkpoint would take): same bounds, and the
    two arithmetic paths agree with each other far inside them."""
    from adv_grpo_amd.model_configs import QwenVaeConfig
    dec2, lat, img2, ref = _decode_both(QwenVaeConfig(), 24, 2, 16, 16, "bf16x3")
    assert any(k.endswith("@bf16") for k in dec2.w)
    dec3, _, img3, _ = _decode_both(QwenVaeConfig(), 24, 2, 16, 16, "bf16x3", bf16_weights=False)
    assert not any(k.endswith("@bf16") for k in dec3.w)
    e2, e3 = (img2 - ref).abs(), (img3 - ref).abs()
    print("two products", e2.mean().item(), e2.max().item(), "three", e3.mean().item(), e3.max().item())
    assert e2.mean().item() < 2e-5 and e3.mean().item() < 2e-5 and e2.max().item() < 1e-3 and e3.max().item() < 1e-3
    # what bench.py prints as vae.mode_ran must exist for this decoder too (round 6: the SD3 decoder's new f16x1 switch broke the inherited method)
    assert isinstance(dec2.arithmetic()["text"], str) and dec2.f16_single is False


def test_qwen_vae_decode_bf16_mode_vs_fp32_oracle():
    from adv_grpo_amd.model_configs import QwenVaeConfig
    _, _, img, ref = _decode_both(QwenVaeConfig(), 22, 2, 32, 32, "bf16")
    err = (img - ref).abs()
    print("qwen vae bf16 image err mean", err.mean().item(), "max", err.max().item())
    assert err.mean().item() < 4e-3 and err.max().item() < 6e-2


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
def test_qwen_vae_decode_at_1024_vs_fp32_oracle(mode):
    """BASELINE config 5's decode size: a 128 x 128 latent -> 1024^2 image (mid attention over 16384 keys, the 96-wide stage at
    1024^2), both modes, same bounds."""
    from adv_grpo_amd.model_configs import QwenVaeConfig
    _, _, img, ref = _decode_both(QwenVaeConfig(), 23, 1, 128, 128, mode)
    assert img.shape == (1, 3, 1024, 1024) and torch.isfinite(img).all()
    err = (img - ref).abs()
    print(f"qwen vae {mode} @1024^2: image err mean {err.mean().item():.3e} max {err.max().item():.3e}")
    if mode == "bf16x3":
        assert err.mean().item() < 2e-5 and err.max().item() < 1e-3
    else:
        assert err.mean().item() < 4e-3 and err.max().item() < 6e-2
