"""GPU parity of the fused attention kernel against a plain PyTorch fp32 reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, H, causal=False):
    B, Sq, HD = q.shape
    D = HD // H
    qf, kf, vf = (x.float().view(B, -1, H, D).transpose(1, 2) for x in (q, k, v))
    s = qf @ kf.transpose(-1, -2) * D ** -0.5
    if causal:
        s = s.masked_fill(torch.ones(Sq, k.shape[1], device=q.device).triu(1).bool(), float("-inf"))
    return (s.softmax(-1) @ vf).transpose(1, 2).reshape(B, Sq, HD)


@pytest.mark.parametrize("B,H,Sq,Skv", [(2, 3, 128, 64), (1, 2, 1229, 1229), (2, 24, 333, 333), (1, 12, 1370, 1370),
                                       (3, 4, 77, 77), (1, 1, 16, 200)])
def test_attention_matches_fp32_reference(B, H, Sq, Skv):
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(Sq + H)
    D = 64
    # packed QKV buffer: q, k, v are strided views, as produced by the fused QKV GEMM
    S = max(Sq, Skv)
    qkv = torch.randn(B, S, 3 * H * D, device="cuda", generator=g).to(torch.bfloat16)
    q, k, v = qkv[:, :Sq, :H * D], qkv[:, :Skv, H * D:2 * H * D], qkv[:, :Skv, 2 * H * D:]
    out = ops.attention(q, k, v, H)
    ref = _ref(q, k, v, H)
    # bf16 probabilities (2^-9 relative) and bf16 output rounding: |err| <= ~1e-2 on O(1) values
    err = (out.float() - ref).abs().max().item()
    assert err < 2e-2, err
    assert (out.float() - ref).abs().mean().item() < 2e-3


@pytest.mark.parametrize("B,H,Sq,Skv", [(8, 16, 257, 257), (2, 16, 50, 50), (1, 4, 300, 320), (3, 2, 512, 65), (2, 3, 1, 257),
                                       (1, 2, 513, 257), (1, 2, 257, 321), (2, 16, 577, 577)])
def test_attention_head_dim_80_matches_fp32_reference(B, H, Sq, Skv):
    """CLIP ViT-H's vision tower (16 heads x 80): the LDS-resident kernel (Skv <= 320, Sq <= 512: 257 tokens at 224^2) and the tiled
    one behind it (the last three shapes), ragged tails in both directions, the log-sum-exp output."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(Sq * 7 + Skv)
    D, S = 80, max(Sq, Skv)
    qkv = torch.randn(B, S, 3 * H * D, device="cuda", generator=g).to(torch.bfloat16)
    q, k, v = qkv[:, :Sq, :H * D], qkv[:, :Skv, H * D:2 * H * D], qkv[:, :Skv, 2 * H * D:]
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device="cuda")
    out = ops.attention(q, k, v, H, lse=lse)
    ref = _ref(q, k, v, H)
    assert (out.float() - ref).abs().max().item() < 2e-2
    assert (out.float() - ref).abs().mean().item() < 2e-3
    qf, kf = (x.float().view(B, -1, H, D).transpose(1, 2) for x in (q, k))
    ref_lse = torch.logsumexp(qf @ kf.transpose(-1, -2) * D ** -0.5, -1) * 1.4426950408889634      # base 2
    assert (lse - ref_lse).abs().max().item() < 2e-2


def test_attention_online_softmax_rescale_is_exercised():
    """Spike one key per query late in the sequence so the running max jumps in a later tile."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    B, H, S, D = 1, 2, 512, 64
    q = torch.randn(B, S, H * D, device="cuda", generator=g)
    k = torch.randn(B, S, H * D, device="cuda", generator=g)
    v = torch.randn(B, S, H * D, device="cuda", generator=g)
    k[:, 400:, :] *= 6.0      # later tiles carry much larger logits
    q, k, v = (x.to(torch.bfloat16) for x in (q, k, v))
    out = ops.attention(q, k, v, H)
    ref = _ref(q, k, v, H)
    assert (out.float() - ref).abs().max().item() < 3e-2


def test_attention_causal():
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    B, H, S, D = 2, 16, 77, 64
    q, k, v = (torch.randn(B, S, H * D, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3))
    out = ops.attention(q, k, v, H, causal=True)
    assert (out.float() - _ref(q, k, v, H, causal=True)).abs().max().item() < 2e-2


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 128, 128), (2, 3, 333, 333), (1, 24, 1229, 1229), (2, 2, 64, 200)])
def test_attention_backward_matches_autograd(B, H, Sq, Skv):
    """dq, dk, dv against torch autograd of the fp32 reference; bf16 probabilities/gradients => ~1e-2 relative."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + Sq)
    D = 64
    S = max(Sq, Skv)
    qkv = (torch.randn(B, S, 3 * H * D, device="cuda", generator=g)).to(torch.bfloat16)
    q, k, v = qkv[:, :Sq, :H * D], qkv[:, :Skv, H * D:2 * H * D], qkv[:, :Skv, 2 * H * D:]
    d_o = torch.randn(B, Sq, H * D, device="cuda", generator=g).to(torch.bfloat16)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device="cuda")
    o = ops.attention(q, k, v, H, lse=lse)
    dqkv = torch.zeros(B, S, 3 * H * D, dtype=torch.bfloat16, device="cuda")
    dq, dk, dv = dqkv[:, :Sq, :H * D], dqkv[:, :Skv, H * D:2 * H * D], dqkv[:, :Skv, 2 * H * D:]
    ops.attention_bwd(q, k, v, o, d_o, lse, H, dq, dk, dv)
    qf, kf, vf = (x.float().clone().requires_grad_(True) for x in (q, k, v))
    ref = _ref(qf, kf, vf, H)
    ref.backward(d_o.float())
    # lse check (base 2)
    sc = (qf.view(B, Sq, H, D).transpose(1, 2) @ kf.view(B, Skv, H, D).transpose(1, 2).transpose(-1, -2)) * D ** -0.5
    assert (lse - torch.logsumexp(sc, -1) * 1.4426950408889634).abs().max().item() < 2e-3
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        rel = ((got.float() - want).norm() / want.norm()).item()
        assert rel < 2e-2, (name, rel)


@pytest.mark.parametrize("where", ["late", "early"])
def test_attention_scores_far_outside_the_first_tiles_window(where):
    """The pipelined head-dim-64 kernel takes every probability relative to the row maximum of the FIRST 64 keys and never
    rescales on the way; a row sum that over- or underflows sends the workgroup through its running-maximum fallback.  Keys
    scaled by 200 put logits hundreds of octaves above (late) or the first tile hundreds above the rest (early)."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    B, H, S, D = 1, 2, 640, 64
    q = torch.randn(B, S, H * D, device="cuda", generator=g)
    k = torch.randn(B, S, H * D, device="cuda", generator=g)
    v = torch.randn(B, S, H * D, device="cuda", generator=g)
    if where == "late":
        k[:, 400:, :] *= 200.0
    else:
        k[:, :64, :] *= 200.0
    q, k, v = (x.to(torch.bfloat16) for x in (q, k, v))
    lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
    out = ops.attention(q, k, v, H, lse=lse)
    ref = _ref(q, k, v, H)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    assert (out.float() - ref).abs().max().item() < 3e-2
    sc = (q.float().view(B, S, H, D).transpose(1, 2) @ k.float().view(B, S, H, D).transpose(1, 2).transpose(-1, -2)) * D ** -0.5
    want = torch.logsumexp(sc, -1) * 1.4426950408889634
    assert ((lse - want).abs() / want.abs().clamp_min(1.0)).max().item() < 1e-3


@pytest.mark.parametrize("B,H,S", [(8, 4, 77), (8, 4, 64), (2, 24, 1229)])
def test_attention_is_bitwise_reproducible(B, H, S):
    """Two launches on the same data give the same bits, with and without the log-sum-exp output (the training forward and the
    rollout forward of the MMDiT must be identical for ratio = 1 at update 0)."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(S + B)
    qkv = torch.randn(B, S, 3 * H * 64, device="cuda", generator=g).to(torch.bfloat16)
    q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
    first = ops.attention(q, k, v, H).clone()
    for i in range(5):
        lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda") if i % 2 else None
        assert torch.equal(ops.attention(q, k, v, H, lse=lse), first)
