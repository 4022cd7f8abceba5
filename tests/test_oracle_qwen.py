"""CPU checks of the Qwen-Image MMDiT restatement (oracle/qwen_mmdit.py, PARITY UNPINNED -- diffusers is absent from the
image, so what can be checked here are the properties the published architecture implies, not diffusers' outputs)."""
import math

import torch

from adv_grpo_amd import synthetic
from oracle import qwen_mmdit as o


def _cfg():
    return o.QwenMMDiTConfig(num_layers=2, num_heads=2, head_dim=32, joint_attention_dim=48, axes_dims_rope=(8, 12, 12))


def test_pack_unpack_are_inverse_and_channel_major():
    lat = torch.arange(2 * 3 * 4 * 6, dtype=torch.float32).view(2, 3, 4, 6)
    tok = o.pack_latents(lat)
    assert tok.shape == (2, 6, 12)
    # token (y, x) = (1, 2): column c * 4 + py * 2 + px
    assert tok[0, 1 * 3 + 2, 1 * 4 + 1 * 2 + 0] == lat[0, 1, 3, 4]
    assert torch.equal(o.unpack_latents(tok, 4, 6), lat)


def test_rope_tables_are_unit_rotations_with_centred_axes():
    cfg = _cfg()
    vid, txt = o.rope_freqs(cfg, 1, 6, 4, 5)
    assert vid.shape == (24, 16) and txt.shape == (5, 16)
    assert torch.allclose(vid.abs(), torch.ones(24, 16)) and torch.allclose(txt.abs(), torch.ones(5, 16))
    # frame axis: position 0 everywhere -> angle 0; height axis of row y is index y - 3 (centred), width x - 2
    assert torch.allclose(vid[:, :4].angle(), torch.zeros(24, 4))
    f_h = 1.0 / torch.pow(torch.tensor(cfg.rope_theta), torch.arange(0, 12, 2).float() / 12)
    for y in range(6):
        for x in range(4):
            e = torch.polar(torch.ones(6), (y - 3) * f_h)
            assert torch.allclose(vid[y * 4 + x, 4:10], e, atol=1e-6)
    # text positions start after the largest half-extent (3) on every axis
    assert torch.allclose(txt[0, :4], torch.polar(torch.ones(4), 3 / torch.pow(torch.tensor(cfg.rope_theta), torch.arange(0, 8, 2).float() / 8)),
                          atol=1e-6)


def test_apply_rope_rotates_adjacent_pairs():
    x = torch.randn(1, 3, 2, 8)
    ang = torch.rand(3, 4)
    y = o.apply_rope(x, torch.polar(torch.ones(3, 4), ang))
    xe, xo = x[..., 0::2], x[..., 1::2]
    c, s = torch.cos(ang)[None, :, None], torch.sin(ang)[None, :, None]
    assert torch.allclose(y[..., 0::2], xe * c - xo * s, atol=1e-6) and torch.allclose(y[..., 1::2], xe * s + xo * c, atol=1e-6)
    assert torch.allclose(y.norm(dim=-1), x.norm(dim=-1), atol=1e-5)


def test_forward_shapes_and_timestep_convention():
    cfg = _cfg()
    W = synthetic.qwen_mmdit_weights(cfg, 3)
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(2, 16, 8, 12, generator=g)
    ctx = torch.randn(2, 7, cfg.joint_attention_dim, generator=g)
    sig = torch.tensor([0.9133, 0.9133])
    out, inter = o.qwen_forward(W, cfg, lat, sig, ctx, return_intermediates=True)
    assert out.shape == lat.shape and torch.isfinite(out).all()
    assert inter["x2"].shape == (2, 24, cfg.dim) and inter["c2"].shape == (2, 7, cfg.dim)
    # the sinusoid is taken at sigma * 1000 (Timesteps(scale=1000))
    e = o.timestep_sinusoid(sig[:1])
    assert math.isclose(e[0, 0].item(), math.cos(913.3), rel_tol=1e-4)
    # text tokens of one sample do not see another sample; samples are independent
    out1 = o.qwen_forward(W, cfg, lat[:1], sig[:1], ctx[:1])
    assert torch.allclose(out1, out[:1], atol=1e-5)
