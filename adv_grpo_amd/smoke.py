"""smoke(): one tiny pass of the hot path on cuda:0, checked against the CPU oracle.
(Allowed oracle import: __graft_entry__.smoke() is one of the three legal checker sites.)"""
import math

import numpy as np
import torch


def run():
    from adv_grpo_amd import losses, stat_tracking
    from adv_grpo_amd.diffusers_patch.sd3_sde_with_logprob import sde_step_cfg
    from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from oracle import grouping as o_grouping
    from oracle import losses as o_losses
    from oracle import sde as o_sde
    from oracle.scheduler import FlowMatchEulerScheduler

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    B, shape = 2, (2, 16, 32, 32)
    vu = torch.randn(shape, generator=g).to(torch.bfloat16)
    vt = torch.randn(shape, generator=g).to(torch.bfloat16)
    x = torch.randn(shape, generator=g).to(torch.bfloat16)
    eps = torch.randn(shape, generator=g)
    sch = FlowMatchEulerDiscreteScheduler(device=dev); sch.set_timesteps(10)
    osch = FlowMatchEulerScheduler(); osch.set_timesteps(10)
    nxt, cast, lp, mean, std = sde_step_cfg(sch, vu.to(dev), vt.to(dev), 4.5, None, x.to(dev), 0.8,
                                            noise=eps.to(dev), out_dtype=torch.bfloat16, step_index=1)
    v = o_sde.cfg_combine(vu, vt, 4.5)
    o_nxt, o_lp, o_mean, o_std = o_sde.sde_step_with_logprob(osch, v.float(), osch.timesteps[1:2], x.float(), 0.8,
                                                             noise=eps)
    assert torch.equal(mean.cpu(), o_mean), "sde mean not bit-exact"
    assert torch.equal(nxt.cpu(), o_nxt), "sde next not bit-exact"
    assert torch.equal(cast.cpu(), o_nxt.to(torch.bfloat16))
    assert torch.allclose(lp.cpu(), o_lp, rtol=1e-5, atol=0)
    # group advantage + GRPO loss
    ids = torch.tensor(np.repeat(np.arange(6), 4).astype(np.int32))
    r = torch.rand(24, 2, generator=g)
    adv = stat_tracking.group_advantage(r.to(dev), ids.to(dev), True).cpu().numpy()
    assert np.array_equal(adv, o_grouping.group_advantages(ids.numpy(), r.numpy(), True))
    old = lp.cpu() + 1e-5
    scal, grad = losses.grpo_loss(lp, old.to(dev), torch.tensor(adv[:B, 0], dtype=torch.float32, device=dev), 5, 1e-5)
    o_loss, _ = o_losses.grpo_loss(lp.cpu(), old, torch.tensor(adv[:B, 0], dtype=torch.float32), 5, 1e-5)
    assert math.isclose(scal[0].item(), o_loss.item(), rel_tol=1e-5, abs_tol=1e-7)
    _matrix_unit(dev, g)
    torch.cuda.synchronize()


def _matrix_unit(dev, g):
    """The kernels that carry the FLOPs, one small launch each, against fp32 torch arithmetic of the same operands (the
    oracle's formulas): the eight-phase 256 x 256 GEMM as a PAIRED launch with the bias + QK-norm epilogue, the pipelined
    head-dim-64 attention forward and backward, the head-dim-128 attention, and the split-bf16 3 x 3 convolution.  A wrong MFMA operand order, a
    broken LDS swizzle or a mis-counted DMA wait fails here."""
    from adv_grpo_amd import _lib, ops
    from oracle import mmdit as o_m
    bf16 = torch.bfloat16
    rnd = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k)
    # ---- eight-phase GEMM, fused QKV shape in small: image rows + text rows in one launch, 16 q | k heads normalised, 8 v heads not
    M, Mt, K, H = 8192, 512, 256, 8
    N = 3 * H * 64
    lib = _lib.load()
    assert lib.advgrpo_gemm_variant(M, N, K, 1, 0) == 30, "the wide-Linear shape no longer dispatches gemm8p_kernel"
    a, at = rnd(M, K).to(bf16), rnd(Mt, K).to(bf16)
    w, wt = rnd(N, K, k=K ** -0.5).to(bf16), rnd(N, K, k=K ** -0.5).to(bf16)
    b, bt = rnd(N, k=0.1).to(bf16), rnd(N, k=0.1).to(bf16)
    rw, rwt = (1 + 0.1 * rnd(2, 64)).to(bf16), (1 + 0.1 * rnd(2, 64)).to(bf16)
    S = M + Mt
    out = torch.zeros(S, N, dtype=bf16, device=dev)
    d = lambda t: t.to(dev)
    # (a descriptor holds raw pointers: the device copies must outlive the launch)
    da, dw, db, drw, dat, dwt, dbt, drwt = (d(t) for t in (a, w, b, rw, at, wt, bt, rwt))
    ops.gemm_grouped([ops.gemm_desc(da, dw, bias=db, out=out, seg=(M, S, 0), rms=(drw, 2 * H, H, 1e-6, None)),
                      ops.gemm_desc(dat, dwt, bias=dbt, out=out, seg=(Mt, S, M), rms=(drwt, 2 * H, H, 1e-6, None))])

    def ref_rows(x, wm, bias, rmsw):
        y = (x.float() @ wm.float().t() + bias.float()).to(bf16)
        y = y.view(-1, 3, H, 64)
        for part in range(2):
            y[:, part] = o_m._rms(y[:, part], rmsw[part])
        return y.reshape(-1, N)
    ref = torch.cat([ref_rows(a, w, b, rw), ref_rows(at, wt, bt, rwt)])
    err = (out.cpu().float() - ref.float()).abs().max().item()
    assert err < 6e-2, f"gemm8p + QK-norm epilogue: max error {err}"
    # ---- attention: head dim 64 (pipelined kernel) and 128 (8-wave kernel), ragged lengths
    for hd, S_att in ((64, 300), (128, 200)):
        Hh = 2
        qkv = rnd(1, S_att, 3 * Hh * hd).to(bf16)
        q, k, v = (qkv[..., i * Hh * hd:(i + 1) * Hh * hd] for i in range(3))
        dq = d(qkv)
        o = ops.attention(dq[..., :Hh * hd], dq[..., Hh * hd:2 * Hh * hd], dq[..., 2 * Hh * hd:], Hh)
        heads = lambda t: t.float().view(1, S_att, Hh, hd).transpose(1, 2)
        r = torch.nn.functional.scaled_dot_product_attention(heads(q), heads(k), heads(v)).transpose(1, 2).reshape(1, S_att, Hh * hd)
        err = (o.cpu().float() - r).abs().max().item()
        assert err < 2e-2, f"attention head dim {hd}: max error {err}"
        if True:          # the backward kernels (dQ and dK / dV; head dim 64: pipelined, 128: attention_bwd_d128.hip) against autograd of the same fp32 attention
            lse = torch.empty(1, Hh, S_att, dtype=torch.float32, device=dev)
            o = ops.attention(dq[..., :Hh * hd], dq[..., Hh * hd:2 * Hh * hd], dq[..., 2 * Hh * hd:], Hh, lse=lse)
            d_o = rnd(1, S_att, Hh * hd).to(bf16)
            dd_o, grads = d(d_o), torch.zeros(1, S_att, 3 * Hh * hd, dtype=bf16, device=dev)
            ops.attention_bwd(dq[..., :Hh * hd], dq[..., Hh * hd:2 * Hh * hd], dq[..., 2 * Hh * hd:], o, dd_o, lse, Hh,
                              grads[..., :Hh * hd], grads[..., Hh * hd:2 * Hh * hd], grads[..., 2 * Hh * hd:])
            leaf = [t.float().clone().requires_grad_(True) for t in (q, k, v)]
            with torch.enable_grad():
                r = torch.nn.functional.scaled_dot_product_attention(*(heads(t) for t in leaf)).transpose(1, 2).reshape(1, S_att, Hh * hd)
                r.backward(d_o.float())
            want = torch.cat([t.grad for t in leaf], dim=-1)
            rel = ((grads.cpu().float() - want).norm() / want.norm()).item()
            assert rel < 2e-2, f"attention backward (head dim {hd}): relative error {rel}"
    # ---- split-bf16 ("fp32-equivalent") 3 x 3 convolution, one 16 x 24 image of 64 -> 128 channels
    x = rnd(1, 16, 24, 64)
    wc, bc = rnd(128, 64, 3, 3, k=(64 * 9) ** -0.5), rnd(128, k=0.1)
    w3 = ops.split_x3(d(wc).permute(0, 2, 3, 1).contiguous(), order=1).reshape(128, -1)
    x3, dbc = ops.split_x3(d(x)), d(bc)
    y = ops.conv3x3_x3(x3, w3, bias=dbc)
    r = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wc.double(), bc.double(), padding=1).permute(0, 2, 3, 1)
    err = (y.cpu().double() - r).abs().max().item()
    assert err < 2e-4, f"conv3x3_x3: max error {err}"
    # ---- the Qwen-Image decoder's pieces: per-pixel RMS norm -> [hi | . | lo] rows -> two-product convolution on one-piece bf16 weights
    gam = 1 + 0.1 * rnd(64)
    wb = wc.to(bf16)
    a2 = ops.rmsnorm_nhwc(d(x), d(gam), 8.0, silu=True, out="x3pair")
    y2 = ops.conv3x3_f16x2(a2, d(wb).permute(0, 2, 3, 1).reshape(128, -1).contiguous(), bias=dbc, bf16_pieces=True)
    xn = torch.nn.functional.silu(torch.nn.functional.normalize(x, dim=-1) * 8.0 * gam)
    r2 = torch.nn.functional.conv2d(xn.permute(0, 3, 1, 2).double(), wb.double(), bc.double(), padding=1).permute(0, 2, 3, 1)
    err = (y2.cpu().double() - r2).abs().max().item()
    assert err < 2e-4, f"rmsnorm_nhwc + conv3x3 bf16x2: max error {err}"
