"""GPU parity (through the C ABI) of the leaf kernels: fused CFG+SDE+log-prob (forward, replay,
backward, Philox), group advantage, GRPO loss -- against the oracle and the reference goldens."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _groups(npz):
    names = sorted({k.split("/")[0] for k in npz.files})
    return {n: {k.split("/", 1)[1]: npz[k] for k in npz.files if k.startswith(n + "/")} for n in names}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("case", ["s10_a", "s10_b", "s10_c", "s4_a", "s4_b", "s10_last"])
def test_sde_step_vs_reference_golden(dev, case):
    from adv_grpo_amd.diffusers_patch.sd3_sde_with_logprob import sde_step_with_logprob
    from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
    g = _groups(np.load(os.path.join(G, "sde_step.npz")))[case]
    nsteps, idx, nl = int(g["meta"][0]), int(g["meta"][1]), float(g["meta"][2])
    sch = FlowMatchEulerDiscreteScheduler(device=dev); sch.set_timesteps(nsteps)
    v, x, eps = (torch.from_numpy(g[k]).to(dev) for k in ("v", "x", "eps"))
    t = sch.timesteps[idx].unsqueeze(0)
    nxt, lp, mean, std = sde_step_with_logprob(sch, v, t, x, noise_level=nl, noise=eps)
    # elementwise results: bit-exact with the reference (same f32 op sequence, no FMA)
    assert np.array_equal(mean.cpu().numpy(), g["mean"])
    assert np.array_equal(nxt.cpu().numpy(), g["next"])
    assert np.array_equal(std.reshape(-1).cpu().numpy(), np.broadcast_to(g["std"], (v.shape[0],)))
    # log_prob: summation order differs from torch's -> relative 2e-6 (n <= 16384 f32 terms)
    np.testing.assert_allclose(lp.cpu().numpy(), g["log_prob"], rtol=2e-6, atol=1e-12)
    # replay with the bf16-rounded next latents and per-sample timesteps (TP:258-265)
    tb = sch.timesteps[idx].repeat(v.shape[0])
    prev = torch.from_numpy(g["next"]).to(torch.bfloat16).to(dev)
    nxt_r, lp_r, mean_r, std_r = sde_step_with_logprob(sch, v, tb, x, noise_level=nl, prev_sample=prev)
    assert np.array_equal(mean_r.cpu().numpy(), g["mean"])
    np.testing.assert_allclose(lp_r.cpu().numpy(), g["log_prob_replay"], rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("B,shape", [(8, (16, 64, 64)), (3, (16, 8, 8)), (16, (16, 128, 128))])
def test_sde_step_cfg_bf16_full_size(dev, B, shape):
    """C2-sized inputs (8 x 16x64x64, bf16 v and x, CFG 4.5): vs the oracle with the reference's
    bf16 CFG roundings; includes the cast back to bf16 (PF:654-655)."""
    from adv_grpo_amd.diffusers_patch.sd3_sde_with_logprob import sde_step_cfg
    from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from oracle import sde as o_sde
    from oracle.scheduler import FlowMatchEulerScheduler
    g = torch.Generator().manual_seed(B)
    full = (B,) + shape
    vu = torch.randn(full, generator=g).to(torch.bfloat16)
    vt = torch.randn(full, generator=g).to(torch.bfloat16)
    x = torch.randn(full, generator=g).to(torch.bfloat16)
    eps = torch.randn(full, generator=g)
    sch = FlowMatchEulerDiscreteScheduler(device=dev); sch.set_timesteps(10)
    osch = FlowMatchEulerScheduler(); osch.set_timesteps(10)
    for step in (0, 1, 9):
        nxt, cast, lp, mean, std = sde_step_cfg(sch, vu.to(dev), vt.to(dev), 4.5, None, x.to(dev), 0.8,
                                                noise=eps.to(dev), out_dtype=torch.bfloat16, step_index=step)
        v = o_sde.cfg_combine(vu, vt, 4.5)
        o_nxt, o_lp, o_mean, _ = o_sde.sde_step_with_logprob(osch, v.float(), osch.timesteps[step:step + 1],
                                                             x.float(), 0.8, noise=eps)
        assert torch.equal(mean.cpu(), o_mean)
        assert torch.equal(nxt.cpu(), o_nxt)
        assert torch.equal(cast.cpu(), o_nxt.to(torch.bfloat16))
        np.testing.assert_allclose(lp.cpu().numpy(), o_lp.numpy(), rtol=5e-6, atol=1e-12)
        # size-independent property: with epsilon injected, log_prob == -std^2 * mean(eps^2)
        np.testing.assert_allclose(lp.cpu().numpy(),
                                   -(std.reshape(-1).cpu().numpy() ** 2) * (eps ** 2).mean(dim=(1, 2, 3)).numpy(),
                                   rtol=2e-4)


def test_sde_step_backward_matches_autograd(dev):
    from adv_grpo_amd import _lib
    from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from oracle import sde as o_sde
    from oracle.scheduler import FlowMatchEulerScheduler
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    B, shape = 4, (4, 16, 16, 16)
    for dt in (torch.float32, torch.bfloat16):
        vu = torch.randn(shape, generator=g).to(dt)
        vt = torch.randn(shape, generator=g).to(dt)
        x = torch.randn(shape, generator=g).to(dt)
        prev = (x.float() + 0.1 * torch.randn(shape, generator=g)).to(dt)
        glp = torch.randn(B, generator=g)
        osch = FlowMatchEulerScheduler(); osch.set_timesteps(10)
        vu_r = vu.clone().requires_grad_(True); vt_r = vt.clone().requires_grad_(True)
        v = o_sde.cfg_combine(vu_r, vt_r, 4.5)
        _, lp, _, _ = o_sde.sde_step_with_logprob(osch, v.float(), osch.timesteps[1].repeat(B), x.float(), 0.8,
                                                  prev_sample=prev.float())
        (lp * glp).sum().backward()
        sch = FlowMatchEulerDiscreteScheduler(device=dev); sch.set_timesteps(10)
        gu = torch.empty(shape, dtype=dt, device=dev); gt = torch.empty(shape, dtype=dt, device=dev)
        n = vu[0].numel()
        a = [t.to(dev).contiguous() for t in (vu, vt, x, prev, glp)]
        _lib.check(lib.advgrpo_sde_step_bwd(_lib.ptr(a[0]), _lib.ptr(a[1]), _lib.dtype_code(dt), 4.5, _lib.ptr(a[2]),
                                            _lib.dtype_code(dt), _lib.ptr(sch.sigmas[1:2]), _lib.ptr(sch.sigmas[2:3]),
                                            0, float(math.sin(0.8 * math.pi / 2)), _lib.ptr(a[3]), _lib.dtype_code(dt),
                                            _lib.ptr(a[4]), _lib.ptr(gu), _lib.ptr(gt), B, n, _lib.stream_ptr()))
        # f32: 1e-5 relative to the gradient scale; bf16: autograd itself rounds grads to bf16 (2^-8)
        tol = 1e-5 if dt == torch.float32 else 2e-2
        scale = vt_r.grad.float().abs().max().item()
        assert (gt.cpu().float() - vt_r.grad.float()).abs().max().item() <= tol * scale
        assert (gu.cpu().float() - vu_r.grad.float()).abs().max().item() <= tol * scale


def test_sde_step_backward_with_kl_term_matches_autograd(dev):
    """config.train.beta > 0 (TP:1105-1108,1126-1130): loss = sum_b glp_b log_prob_b + w * mean_b mean_elem (mean - mean_ref)^2;
    gradient w.r.t. both CFG halves and the per-sample KL value against torch autograd on the oracle's SDE step."""
    from adv_grpo_amd import _lib
    from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from oracle import losses as o_loss
    from oracle import sde as o_sde
    from oracle.scheduler import FlowMatchEulerScheduler
    lib = _lib.load()
    g = torch.Generator().manual_seed(6)
    B, shape, w = 4, (4, 16, 16, 16), 0.37
    for dt in (torch.float32, torch.bfloat16):
        vu, vt, ru, rt, x = (torch.randn(shape, generator=g).to(dt) for _ in range(5))
        prev = (x.float() + 0.1 * torch.randn(shape, generator=g)).to(dt)
        glp = torch.randn(B, generator=g) * 1e-3          # (policy and KL gradients of the same order)
        osch = FlowMatchEulerScheduler(); osch.set_timesteps(10)
        step = lambda a, b: o_sde.sde_step_with_logprob(osch, o_sde.cfg_combine(a, b, 4.5).float(), osch.timesteps[1].repeat(B),
                                                        x.float(), 0.8, prev_sample=prev.float())
        with torch.no_grad():
            _, _, mean_ref, _ = step(ru, rt)
        vu_r = vu.clone().requires_grad_(True); vt_r = vt.clone().requires_grad_(True)
        _, lp, mean, _ = step(vu_r, vt_r)
        kl_ref = ((mean - mean_ref) ** 2).mean(dim=(1, 2, 3))
        ((lp * glp).sum() + w * o_loss.kl_loss(mean, mean_ref)).backward()
        sch = FlowMatchEulerDiscreteScheduler(device=dev); sch.set_timesteps(10)
        gu = torch.empty(shape, dtype=dt, device=dev); gt = torch.empty(shape, dtype=dt, device=dev)
        kl = torch.empty(B, dtype=torch.float32, device=dev)
        n = vu[0].numel()
        ws = torch.empty(max(1, lib.advgrpo_sde_step_workspace_bytes(B, n) // 4), dtype=torch.float32, device=dev)
        a = [t_.to(dev).contiguous() for t_ in (vu, vt, x, prev, glp, mean_ref.float())]
        _lib.check(lib.advgrpo_sde_step_bwd_kl(_lib.ptr(a[0]), _lib.ptr(a[1]), _lib.dtype_code(dt), 4.5, _lib.ptr(a[2]),
                                               _lib.dtype_code(dt), _lib.ptr(sch.sigmas[1:2]), _lib.ptr(sch.sigmas[2:3]), 0,
                                               float(math.sin(0.8 * math.pi / 2)), _lib.ptr(a[3]), _lib.dtype_code(dt),
                                               _lib.ptr(a[4]), _lib.ptr(a[5]), w, _lib.ptr(gu), _lib.ptr(gt), _lib.ptr(kl),
                                               _lib.ptr(ws), B, n, _lib.stream_ptr()))
        tol = 1e-5 if dt == torch.float32 else 2e-2     # bf16: autograd itself rounds the summed gradient to bf16
        scale = vt_r.grad.float().abs().max().item()
        assert (gt.cpu().float() - vt_r.grad.float()).abs().max().item() <= tol * scale
        assert (gu.cpu().float() - vu_r.grad.float()).abs().max().item() <= tol * scale
        assert torch.allclose(kl.cpu(), kl_ref.detach(), rtol=2e-5 if dt == torch.float32 else 2e-2)
        # and with weight 0 the policy-only entry point is reproduced bit for bit
        gu0 = torch.empty_like(gu); gt0 = torch.empty_like(gt)
        _lib.check(lib.advgrpo_sde_step_bwd(_lib.ptr(a[0]), _lib.ptr(a[1]), _lib.dtype_code(dt), 4.5, _lib.ptr(a[2]),
                                            _lib.dtype_code(dt), _lib.ptr(sch.sigmas[1:2]), _lib.ptr(sch.sigmas[2:3]), 0,
                                            float(math.sin(0.8 * math.pi / 2)), _lib.ptr(a[3]), _lib.dtype_code(dt),
                                            _lib.ptr(a[4]), _lib.ptr(gu0), _lib.ptr(gt0), B, n, _lib.stream_ptr()))
        _lib.check(lib.advgrpo_sde_step_bwd_kl(_lib.ptr(a[0]), _lib.ptr(a[1]), _lib.dtype_code(dt), 4.5, _lib.ptr(a[2]),
                                               _lib.dtype_code(dt), _lib.ptr(sch.sigmas[1:2]), _lib.ptr(sch.sigmas[2:3]), 0,
                                               float(math.sin(0.8 * math.pi / 2)), _lib.ptr(a[3]), _lib.dtype_code(dt),
                                               _lib.ptr(a[4]), _lib.ptr(a[5]), 0.0, _lib.ptr(gu), _lib.ptr(gt), _lib.ptr(kl),
                                               _lib.ptr(ws), B, n, _lib.stream_ptr()))
        assert torch.equal(gu, gu0) and torch.equal(gt, gt0)


def test_philox_noise_statistics_and_reproducibility(dev):
    from adv_grpo_amd import _lib
    lib = _lib.load()
    n = 1 << 22
    a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.empty_like(a); c = torch.empty_like(a)
    _lib.check(lib.advgrpo_randn(_lib.ptr(a), _lib.F32, n, 1234, 0, _lib.stream_ptr()))
    _lib.check(lib.advgrpo_randn(_lib.ptr(b), _lib.F32, n, 1234, 0, _lib.stream_ptr()))
    _lib.check(lib.advgrpo_randn(_lib.ptr(c), _lib.F32, n, 1235, 0, _lib.stream_ptr()))
    assert torch.equal(a, b) and not torch.equal(a, c)
    x = a.double().cpu()
    assert abs(x.mean().item()) < 4 / math.sqrt(n)
    assert abs(x.var().item() - 1) < 0.005
    assert abs((x ** 3).mean().item()) < 0.01 and abs((x ** 4).mean().item() - 3) < 0.03
    # offset continues the same stream
    d = torch.empty(n - 1024, dtype=torch.float32, device=dev)
    _lib.check(lib.advgrpo_randn(_lib.ptr(d), _lib.F32, n - 1024, 1234, 256, _lib.stream_ptr()))
    assert torch.equal(d, a[1024:])


def test_sde_step_philox_mode_is_a_gaussian_step(dev):
    from adv_grpo_amd.diffusers_patch.sd3_sde_with_logprob import sde_step_cfg
    from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
    sch = FlowMatchEulerDiscreteScheduler(device=dev); sch.set_timesteps(10)
    g = torch.Generator().manual_seed(2)
    shape = (8, 16, 64, 64)
    v = torch.randn(shape, generator=g).to(dev); x = torch.randn(shape, generator=g).to(dev)
    nxt, _, lp, mean, std = sde_step_cfg(sch, v, None, 1.0, None, x, 0.8, seed=77, step_index=0)
    z = ((nxt - mean) / std).double()
    assert abs(z.mean().item()) < 0.01 and abs(z.var().item() - 1) < 0.01
    np.testing.assert_allclose(lp.cpu().numpy(), -(std.reshape(-1).cpu().numpy() ** 2), rtol=0.02)
    nxt2, _, lp2, _, _ = sde_step_cfg(sch, v, None, 1.0, None, x, 0.8, seed=77, step_index=0)
    assert torch.equal(nxt, nxt2) and torch.equal(lp, lp2)


@pytest.mark.parametrize("case", ["toy", "epoch", "zero_std"])
def test_group_advantage_vs_reference_golden(dev, case):
    from adv_grpo_amd.stat_tracking import group_advantage
    g = _groups(np.load(os.path.join(G, "stat_tracker.npz")))[case]
    r = torch.from_numpy(g["rewards"]).to(dev)
    ids = torch.from_numpy(g["group_ids"]).to(dev)
    for gs in (0, 1):
        adv = group_advantage(r, ids, bool(gs)).cpu().numpy()
        if g["rewards"].ndim == 1:
            # the reference's 1-D toy input goes through numpy's pairwise path; T==1 here too
            assert np.array_equal(adv, g[f"adv_global{gs}"])
        else:
            assert np.array_equal(adv, g[f"adv_global{gs}"]), f"{case} global_std={gs}: not bit-exact"


@pytest.mark.parametrize("N,T,ngroups", [(768, 2, 48), (768, 1, 48), (1000, 3, 7), (96, 2, 96), (300, 1, 1)])
def test_group_advantage_vs_oracle_random(dev, N, T, ngroups):
    from adv_grpo_amd.stat_tracking import group_advantage
    from oracle import grouping
    rng = np.random.RandomState(N + T)
    ids = rng.randint(0, ngroups, size=N).astype(np.int32) * 7 - 3   # arbitrary (also negative) keys
    r = (rng.randn(N, T) * 0.05 + 0.8).astype(np.float32)
    r[ids == ids[0]] = 0.5                                          # a zero-std group
    for gs in (True, False):
        for rr in (r, r.astype(np.float64) + 1e-9):
            rin = rr[:, 0] if T == 1 else rr
            adv = group_advantage(torch.from_numpy(rin).to(dev), torch.from_numpy(ids).to(dev), gs).cpu().numpy()
            ref = grouping.group_advantages(ids, rin, gs)
            assert adv.shape == ref.shape
            assert np.array_equal(adv, ref), f"N={N} T={T} gs={gs} max|d|={np.abs(adv - ref).max()}"


def test_stat_tracker_class_with_prompt_strings(dev):
    from adv_grpo_amd.stat_tracking import PerPromptStatTracker
    tr = PerPromptStatTracker(global_std=True)
    adv = tr.update(["a", "b", "a", "c", "b", "a"], np.array([1, 2, 3, 4, 5, 6], dtype=np.float64))
    np.testing.assert_allclose(adv.cpu().numpy(), [-1.36618011, -0.87825864, -0.19516859, 0, 0.87825864, 1.56134869],
                               atol=1e-8)
    assert tr.get_stats() == (2.0, 3)
    tr.clear(); assert tr.stats == {}


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_grpo_loss_vs_reference_golden(dev, case):
    from adv_grpo_amd.losses import INFO_KEYS, grpo_loss
    g = _groups(np.load(os.path.join(G, "losses.npz")))[f"grpo_{case}"]
    scal, grad = grpo_loss(torch.from_numpy(g["log_prob"]).to(dev), torch.from_numpy(g["old"]).to(dev),
                           torch.from_numpy(g["adv"]).to(dev), 5, float(g["clip"]))
    scal = scal.cpu().numpy()
    for i, k in enumerate(INFO_KEYS):
        # loss = mean of signed terms of magnitude |A|*ratio <= 5: f32 summation order + device expf (<= 2 ulp)
        # give an absolute error of a few 1e-7; the clip fractions are exact counts
        np.testing.assert_allclose(scal[i], g[k], rtol=2e-6, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(grad.cpu().numpy(), g["grad"], rtol=2e-6, atol=1e-9)


def test_grpo_loss_takes_strided_views():
    """losses.grpo_loss on column views of [G, T] tensors (what the G-step passes: log_probs[:, j], advantages[:, j]) equals the
    call on contiguous copies.  Regression: the f32 / contiguous temporaries were created inside the argument list of the ctypes
    call, freed as soon as their pointer had been taken, and the next temporary re-used the block -- the kernel then read the
    advantages where the old log-probs should have been."""
    from adv_grpo_amd import losses
    g = torch.Generator(device="cuda").manual_seed(3)
    lp = torch.randn(8, 2, device="cuda", generator=g) * 0.01 - 0.8
    old = lp + torch.randn(8, 2, device="cuda", generator=g) * 1e-3
    adv = torch.randn(8, 2, device="cuda", generator=g)
    for j in range(2):
        a, ga = losses.grpo_loss(lp[:, j].contiguous(), old[:, j].contiguous(), adv[:, j].contiguous(), 5.0, 1e-4)
        b, gb = losses.grpo_loss(lp[:, j].contiguous(), old[:, j], adv[:, j], 5.0, 1e-4)
        assert torch.equal(a, b) and torch.equal(ga, gb)
        r = torch.exp(lp[:, j] - old[:, j])
        want = torch.maximum(-adv[:, j] * r, -adv[:, j] * r.clamp(1 - 1e-4, 1 + 1e-4)).mean()
        assert abs(a[0].item() - want.item()) < 1e-6 and abs(a[1].item() - 0.5 * ((lp[:, j] - old[:, j]) ** 2).mean().item()) < 1e-9


@pytest.mark.parametrize("fp8", [False, True])
def test_layernorm_mod_pair_equals_two_launches(fp8):
    """advgrpo_layernorm_mod_pair: the image-stream and text-stream norms of an MMDiT block in one launch, bit for bit the two
    single launches (second output on the first problem, different rows per batch and modulation rows on the second)."""
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    B, Ni, Nt, D = 3, 1024, 154, 1536
    x = torch.randn(B * Ni, D, device="cuda", generator=g).to(torch.bfloat16)
    c = (torch.randn(B * Nt, D, device="cuda", generator=g) * 3 + 1).to(torch.bfloat16)
    mods = (torch.randn(B, 6 * D, device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    m = lambda k: mods[:, k * D:(k + 1) * D]
    def args(first):
        if first:
            kw = dict(x=x, scale=m(1), shift=m(0), scale2=m(3), shift2=m(2), rows_per_batch=Ni)
            M = B * Ni
        else:
            kw = dict(x=c, scale=m(5), shift=m(4), rows_per_batch=Nt)
            M = B * Nt
        if fp8:
            mk = lambda: ops.Fp8Rows(torch.empty(M, D, dtype=torch.uint8, device="cuda"), torch.empty(M, dtype=torch.float32, device="cuda"))
            kw["q"] = mk()
            if first:
                kw["q2"] = mk()
        return kw
    a1, b1 = args(True), args(False)
    ra, rb = ops.layernorm_mod_pair(a1, b1)
    a2, b2 = args(True), args(False)
    if fp8:
        ops.layernorm_mod_fp8(a2.pop("x"), a2.pop("q"), **a2)
        ops.layernorm_mod_fp8(b2.pop("x"), b2.pop("q"), **b2)
        a2, b2 = args(True), args(False)        # fresh destination buffers for the reference launches
        ops.layernorm_mod_fp8(x, a2["q"], q2=a2["q2"], scale=m(1), shift=m(0), scale2=m(3), shift2=m(2), rows_per_batch=Ni)
        ops.layernorm_mod_fp8(c, b2["q"], scale=m(5), shift=m(4), rows_per_batch=Nt)
        for got, want in ((a1["q"], a2["q"]), (a1["q2"], a2["q2"]), (b1["q"], b2["q"])):
            assert torch.equal(got.q, want.q) and torch.equal(got.scale, want.scale)
    else:
        wa = ops.layernorm_mod(x, scale=m(1), shift=m(0), scale2=m(3), shift2=m(2), rows_per_batch=Ni)
        wb = ops.layernorm_mod(c, scale=m(5), shift=m(4), rows_per_batch=Nt)
        assert torch.equal(ra[0], wa[0]) and torch.equal(ra[1], wa[1]) and torch.equal(rb, wb)
