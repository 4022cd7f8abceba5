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
