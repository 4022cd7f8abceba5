"""GPU parity of the VAE decoder path against the fp32 torch oracle (itself "parity unpinned" vs diffusers)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_conv3x3_implicit_gemm_matches_fp32_conv():
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    for (B, H, W, Ci, Co, up) in [(2, 16, 24, 64, 96, False), (1, 8, 8, 128, 64, True), (2, 32, 32, 128, 3, False)]:
        x = torch.randn(B, H, W, Ci, device="cuda", generator=g).to(torch.bfloat16)
        wt = (torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) / (3 * Ci ** 0.5)).to(torch.bfloat16)
        bias = torch.randn(Co, device="cuda", generator=g).to(torch.bfloat16)
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        res = torch.randn(B, Ho, Wo, Co, device="cuda", generator=g).to(torch.bfloat16)
        y = ops.conv3x3(x, wt.permute(0, 2, 3, 1).reshape(Co, -1).contiguous(), bias=bias, upsample=up, residual=res,
                        out_dtype=torch.float32)
        xin = x.float().permute(0, 3, 1, 2)
        if up:
            xin = torch.nn.functional.interpolate(xin, scale_factor=2.0, mode="nearest")
        ref = torch.nn.functional.conv2d(xin, wt.float(), bias.float(), padding=1).permute(0, 2, 3, 1) + res.float()
        assert (y - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-3


def test_groupnorm_silu_and_softmax_rows():
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(2)
    for C in (128, 256, 512):
        x = (torch.randn(2, 24, 24, C, device="cuda", generator=g) * 2 + 0.5).to(torch.bfloat16)
        w = torch.randn(C, device="cuda", generator=g).to(torch.bfloat16)
        b = torch.randn(C, device="cuda", generator=g).to(torch.bfloat16)
        y = ops.groupnorm_nhwc(x, w, b, 32, 1e-6, True)
        ref = torch.nn.functional.silu(torch.nn.functional.group_norm(x.float().permute(0, 3, 1, 2), 32, w.float(),
                                                                       b.float(), 1e-6)).permute(0, 2, 3, 1)
        # bf16 output rounding (2^-8 relative) + f32 statistics
        assert ((y.float() - ref).abs() <= 2 ** -8 * ref.abs() + 5e-3).all()
    s = (torch.randn(37, 4096, device="cuda", generator=g) * 3).to(torch.bfloat16)
    ref = s.float().softmax(-1)
    ops.softmax_rows_(s)
    assert (s.float() - ref).abs().max().item() < 1e-3
    assert ((s.float().sum(-1) - 1).abs() < 2e-2).all()


@pytest.mark.parametrize("B,hw", [(2, 16), (1, 64)])
def test_vae_decode_vs_fp32_oracle(B, hw):
    """Full SD3 VAE decoder shapes (conv widths 512/512/256/128, one mid attention).  hw=64 is the C2
    latent size (512^2 image).  Tolerance (bf16 operands vs the reference's fp32): mean abs error on the
    [0,1] image <= 4e-3 (about one 8-bit level), max abs <= 6e-2."""
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    from oracle import vae as o
    cfg = o.VaeConfig()
    W = synthetic.vae_decoder_weights(cfg, 99)
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    g = torch.Generator().manual_seed(hw)
    lat = torch.randn(B, 16, hw, hw, generator=g).to(torch.bfloat16)
    dec = AutoencoderKLDecoder(Wb, cfg, "cuda")
    img = dec.decode_to_image(lat.cuda())
    W32 = {k: v.float().cuda() for k, v in Wb.items()}
    z = lat.float().cuda() / cfg.scaling_factor + cfg.shift_factor
    ref = o.postprocess(o.vae_decode(W32, cfg, z))
    assert img.shape == (B, 3, 8 * hw, 8 * hw) and img.dtype == torch.float32
    err = (img - ref).abs()
    print("vae image err mean", err.mean().item(), "max", err.max().item(), "ref std", ref.std().item())
    assert err.mean().item() < 4e-3 and err.max().item() < 6e-2
