"""CPU checks of the Qwen-Image VAE decoder restatement (oracle/qwen_vae.py, PARITY UNPINNED -- diffusers is absent from the image,
so what can be checked here are the properties the published architecture implies, and the identities the HIP path relies on)."""
import torch
import torch.nn.functional as F

from adv_grpo_amd import synthetic
from adv_grpo_amd.model_configs import QwenVaeConfig
from oracle import qwen_vae as o


def _small():
    return o.QwenVaeConfig(base_dim=8, dim_mult=(1, 2, 4, 4), num_res_blocks=1)


def test_widths_follow_the_published_decoder():
    cfg = o.QwenVaeConfig()
    assert cfg.dims == [384, 384, 384, 192, 96]
    assert [cfg.up_block_io(i) for i in range(4)] == [(384, 384, True), (192, 384, True), (192, 192, True), (96, 96, False)]
    assert QwenVaeConfig().dims == cfg.dims and QwenVaeConfig().latents_std == cfg.latents_std
    W = synthetic.qwen_vae_decoder_weights(QwenVaeConfig())
    assert W["decoder.conv_in.weight"].shape == (384, 16, 3, 3, 3)
    assert W["decoder.up_blocks.1.resnets.0.conv_shortcut.weight"].shape == (384, 192, 1, 1, 1)
    assert W["decoder.up_blocks.2.upsamplers.0.resample.1.weight"].shape == (96, 192, 3, 3)
    assert W["decoder.mid_block.attentions.0.to_qkv.weight"].shape == (1152, 384, 1, 1)
    assert W["decoder.mid_block.attentions.0.norm.gamma"].shape == (384, 1, 1)
    assert W["decoder.norm_out.gamma"].shape == (96, 1, 1, 1) and W["decoder.conv_out.weight"].shape == (3, 96, 3, 3, 3)
    assert not any("time_conv" in k for k in W)


def test_one_frame_causal_conv_is_the_last_temporal_tap():
    g = torch.Generator().manual_seed(0)
    W = {"c.weight": torch.randn(5, 4, 3, 3, 3, generator=g), "c.bias": torch.randn(5, generator=g)}
    x = torch.randn(2, 4, 1, 6, 7, generator=g)
    y3 = o.causal_conv3d(W, "c", x)
    y2 = F.conv2d(x[:, :, 0], W["c.weight"][:, :, -1], W["c.bias"], padding=1)
    assert y3.shape == (2, 5, 1, 6, 7)
    assert torch.allclose(y3[:, :, 0], y2, atol=1e-5)
    # and it IS causal: with two frames, frame 0's output does not see frame 1
    x2 = torch.cat([x, torch.randn(2, 4, 1, 6, 7, generator=g)], dim=2)
    assert torch.allclose(o.causal_conv3d(W, "c", x2)[:, :, 0], y3[:, :, 0], atol=1e-5)


def test_rms_norm_is_unit_rms_times_gamma():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 12, 1, 3, 3, generator=g) * 7
    gamma = torch.rand(12, 1, 1, 1, generator=g) + 0.5
    y = o.rms_norm(x, gamma)
    ref = x / x.pow(2).mean(1, keepdim=True).sqrt() * gamma
    assert torch.allclose(y, ref, atol=1e-5)


def test_decode_shapes_range_and_sample_independence():
    cfg = _small()
    W = synthetic.qwen_vae_decoder_weights(cfg, seed=5)
    z = torch.randn(2, 16, 3, 5, generator=torch.Generator().manual_seed(2))
    img = o.decode_to_image(W, cfg, z)
    assert img.shape == (2, 3, 24, 40) and img.min() >= 0 and img.max() <= 1
    assert torch.allclose(o.decode_to_image(W, cfg, z[1:]), img[1:], atol=1e-5)
    # de-normalisation: z * std + mean per channel
    d = o.denormalise(cfg, z[:, :, None])
    assert torch.allclose(d[0, 3, 0], z[0, 3] * cfg.latents_std[3] + cfg.latents_mean[3], atol=1e-6)


def test_flops_decode_counts_the_2d_taps_of_every_weight():
    """qwen_vae.flops_decode (what bench.py --config c5 credits per decoded image) against a count made from the synthetic state dict
    itself: every convolution contributes 2 x (spatial taps of ONE temporal slice) x Cin x Cout per output pixel at its resolution."""
    from adv_grpo_amd.qwen_vae import flops_decode
    cfg = QwenVaeConfig()
    W = synthetic.qwen_vae_decoder_weights(cfg)
    h = w = 16
    px = {"mid": h * w}
    total = 0.0
    res = h * w
    for name, t in W.items():
        if not name.endswith(".weight"):
            continue
        if name.startswith("decoder.up_blocks."):
            i = int(name.split(".")[2])
            n = h * w * 4 ** i
            if ".upsamplers." in name:
                n *= 4                                   # the upsampler's convolution runs at the doubled resolution
        elif name.startswith("decoder.conv_out"):
            n = h * w * 64
        else:
            n = h * w
        taps = t.shape[-1] * t.shape[-2]
        total += 2.0 * taps * t.shape[0] * t.shape[1] * n
    total += 4.0 * (h * w) ** 2 * cfg.dims[0]            # the mid attention's two products
    assert abs(flops_decode(cfg, h, w) - total) / total < 1e-9
