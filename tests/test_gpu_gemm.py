"""GPU parity of the bf16 MFMA GEMM (through the C ABI) against a plain PyTorch fp32 reference of the
same op (computed on the bf16-rounded operands)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, w, bias=None, act=None, alpha=1.0, gate=None, gate_rows=0, residual=None):
    y = alpha * (a.float() @ w.float().T)
    if bias is not None:
        y = y + bias.float()
    if act == "gelu_tanh":
        y = torch.nn.functional.gelu(y, approximate="tanh")
    elif act == "gelu":
        y = torch.nn.functional.gelu(y)
    elif act == "silu":
        y = torch.nn.functional.silu(y)
    if gate is not None:
        idx = torch.arange(a.shape[0], device=a.device) // gate_rows
        y = y * gate.float()[idx]
    if residual is not None:
        y = y + residual.float()
    return y


def _check(out, ref, K):
    # bf16 output: half-ulp 2^-9 relative to the result, plus f32 accumulation noise ~ sqrt(K) * 2^-24 * |a||w|
    err = (out.float() - ref).abs()
    tol = 2 ** -8 * ref.abs() + 1e-3 * ref.abs().mean()
    assert (err <= tol).all(), f"max err {err.max().item()} (ref scale {ref.abs().mean().item()})"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (1229, 1536, 1536), (3280, 4608, 1536),
                                   (16384, 1536, 1536), (100, 64, 1536), (77, 200, 192), (16, 13824, 1536),
                                   (2057, 3072, 768)])
def test_gemm_plain(M, N, K):
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    # asymmetric, non-identity operands (transposes / fragment swaps would show)
    w = (torch.randn(N, K, device="cuda", generator=g) * (1 + torch.arange(N, device="cuda")[:, None] / N)
         ).to(torch.bfloat16)
    out = ops.gemm(a, w)
    _check(out, _ref(a, w), K)
    out32 = ops.gemm(a, w, out_dtype=torch.float32)
    ref = _ref(a, w)
    assert (out32 - ref).abs().max() <= 2e-3 * ref.abs().max()


@pytest.mark.parametrize("act", [None, "gelu_tanh", "gelu", "silu"])
def test_gemm_epilogue(act):
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    B, S, K, N = 4, 333, 256, 520
    M = B * S
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    gate = torch.randn(B, N, device="cuda", generator=g).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    out = ops.gemm(a, w, bias=bias, act=act, alpha=0.5, gate=gate, gate_rows=S, residual=res)
    _check(out, _ref(a, w, bias, act, 0.5, gate, S, res), K)


def test_gemm_row_segments_scatter_into_joint_buffer():
    """image tokens -> rows [b*S, b*S+Ni), text tokens -> rows [b*S+Ni, (b+1)*S) of one buffer."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(4)
    B, Ni, Nt, K, N = 3, 64, 21, 128, 192
    S = Ni + Nt
    xi = torch.randn(B * Ni, K, device="cuda", generator=g).to(torch.bfloat16)
    xt = torch.randn(B * Nt, K, device="cuda", generator=g).to(torch.bfloat16)
    wi = torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16)
    wt = torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16)
    joint = torch.zeros(B * S, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm(xi, wi, out=joint, seg=(Ni, S, 0))
    ops.gemm(xt, wt, out=joint, seg=(Nt, S, Ni))
    ref = torch.cat([(xi.float() @ wi.float().T).view(B, Ni, N), (xt.float() @ wt.float().T).view(B, Nt, N)], 1)
    _check(joint.view(B, S, N), ref, K)


def test_bmm_nt():
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(5, 300, 512, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn(5, 260, 512, device="cuda", generator=g).to(torch.bfloat16)
    out = ops.bmm_nt(a, w, alpha=0.25)
    _check(out, 0.25 * torch.einsum("bmk,bnk->bmn", a.float(), w.float()), 512)


@pytest.mark.parametrize("shape", [(3, 64, 21, 128, 384), (16, 1024, 205, 256, 4608)])
def test_gemm_grouped_pair_with_fused_qk_norm(shape):
    """One launch for the image- and text-stream QKV projections with the per-head RMSNorm in the epilogue ==
    two launches + the standalone rmsnorm_heads kernel (bit for bit), and == an fp32 torch reference."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(6)
    B, Ni, Nt, K, N = shape
    S, H = Ni + Nt, N // 3 // 64
    xi = torch.randn(B * Ni, K, device="cuda", generator=g).to(torch.bfloat16)
    xt = torch.randn(B * Nt, K, device="cuda", generator=g).to(torch.bfloat16)
    wi = (torch.randn(N, K, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    wt = (torch.randn(N, K, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    bi = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    bt = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    rw_i = (1 + 0.1 * torch.randn(2, 64, device="cuda", generator=g)).to(torch.bfloat16)
    rw_t = (1 + 0.1 * torch.randn(2, 64, device="cuda", generator=g)).to(torch.bfloat16)
    # unfused: two launches + two norm kernels
    ref = torch.zeros(B * S, N, dtype=torch.bfloat16, device="cuda")
    rs_ref = torch.zeros(B * S, 2 * H, dtype=torch.float32, device="cuda")
    ops.gemm(xi, wi, bias=bi, out=ref, seg=(Ni, S, 0))
    ops.gemm(xt, wt, bias=bt, out=ref, seg=(Nt, S, Ni))
    plain = ref.clone()
    ops.rmsnorm_heads(ref, 0, 2 * H, rw_i, H, seg=(Ni, S, 0), M=B * Ni, rs_out=rs_ref)
    ops.rmsnorm_heads(ref, 0, 2 * H, rw_t, H, seg=(Nt, S, Ni), M=B * Nt, rs_out=rs_ref)
    # fused + grouped
    out = torch.zeros(B * S, N, dtype=torch.bfloat16, device="cuda")
    rs = torch.zeros(B * S, 2 * H, dtype=torch.float32, device="cuda")
    ops.gemm_grouped([ops.gemm_desc(xi, wi, bias=bi, out=out, seg=(Ni, S, 0), rms=(rw_i, 2 * H, H, 1e-6, rs)),
                      ops.gemm_desc(xt, wt, bias=bt, out=out, seg=(Nt, S, Ni), rms=(rw_t, 2 * H, H, 1e-6, rs))])
    assert torch.equal(out, ref)
    assert torch.equal(rs, rs_ref)
    # fp32 reference of the norm on the bf16 projection
    q = plain.float().view(B, S, 3, H, 64)
    wsel = torch.cat([rw_i.float().view(1, 1, 2, 1, 64).expand(B, Ni, 2, H, 64),
                      rw_t.float().view(1, 1, 2, 1, 64).expand(B, Nt, 2, H, 64)], 1)
    qk = q[:, :, :2]
    normed = qk * torch.rsqrt(qk.pow(2).mean(-1, keepdim=True) + 1e-6)
    want = torch.cat([normed.to(torch.bfloat16).float() * wsel, q[:, :, 2:]], 2).reshape(B * S, N)
    assert (out.float() - want).abs().max().item() <= 2e-2 * want.abs().max().item()


def test_gemm_grouped_pair_gate_residual_matches_two_launches():
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(7)
    B, Ni, Nt, K, N = 16, 1024, 205, 512, 1536
    xi = torch.randn(B * Ni, K, device="cuda", generator=g).to(torch.bfloat16)
    xt = torch.randn(B * Nt, K, device="cuda", generator=g).to(torch.bfloat16)
    wi = (torch.randn(N, K, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    wt = (torch.randn(N, K, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    bi = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    gi = torch.randn(B, N, device="cuda", generator=g).to(torch.bfloat16)
    gt = torch.randn(B, N, device="cuda", generator=g).to(torch.bfloat16)
    ri = torch.randn(B * Ni, N, device="cuda", generator=g).to(torch.bfloat16)
    rt = torch.randn(B * Nt, N, device="cuda", generator=g).to(torch.bfloat16)
    a = ops.gemm(xi, wi, bias=bi, gate=gi, gate_rows=Ni, residual=ri)
    b = ops.gemm(xt, wt, act="gelu_tanh", gate=gt, gate_rows=Nt, residual=rt)
    oa, ob = ops.gemm_grouped([ops.gemm_desc(xi, wi, bias=bi, gate=gi, gate_rows=Ni, residual=ri),
                               ops.gemm_desc(xt, wt, act="gelu_tanh", gate=gt, gate_rows=Nt, residual=rt)])
    assert torch.equal(oa, a) and torch.equal(ob, b)
