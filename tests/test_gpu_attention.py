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
