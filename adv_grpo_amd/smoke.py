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
    torch.cuda.synchronize()
