"""CPU: the oracle restatement against the golden vectors produced by the reference's own code
(tests/golden/make_golden.py).  Tolerances are written next to each check."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import grouping, losses, rewards, rollout, sde
from oracle.scheduler import FlowMatchEulerScheduler
from oracle.standin import StandinVelocity, standin_vae_decode

G = os.path.join(os.path.dirname(__file__), "golden")


def _groups(npz):
    names = sorted({k.split("/")[0] for k in npz.files})
    return {n: {k.split("/", 1)[1]: npz[k] for k in npz.files if k.startswith(n + "/")} for n in names}


def test_scheduler_known_sigmas():
    # SURVEY Appendix A.1 expected schedules (parity unpinned vs diffusers; formula self-check)
    s = FlowMatchEulerScheduler(); s.set_timesteps(10)
    exp = [1.0, 0.960129, 0.913349, 0.857692, 0.790368, 0.707278, 0.602151, 0.464876, 0.278049, 0.008929, 0]
    np.testing.assert_allclose(s.sigmas.numpy(), exp, atol=2e-6)
    s.set_timesteps(4)
    np.testing.assert_allclose(s.sigmas.numpy(), [1.0, 0.857692, 0.602151, 0.008929, 0], atol=2e-6)
    assert s.index_for_timestep(s.timesteps[2]) == 2


@pytest.mark.parametrize("case", ["s10_a", "s10_b", "s10_c", "s4_a", "s4_b", "s10_last"])
def test_sde_step_bit_exact(case):
    g = _groups(np.load(os.path.join(G, "sde_step.npz")))[case]
    nsteps, idx, nl = int(g["meta"][0]), int(g["meta"][1]), float(g["meta"][2])
    sch = FlowMatchEulerScheduler(); sch.set_timesteps(nsteps)
    assert np.array_equal(sch.sigmas.numpy(), g["sigmas"])
    v, x, eps = (torch.from_numpy(g[k]) for k in ("v", "x", "eps"))
    nxt, lp, mean, std = sde.sde_step_with_logprob(sch, v, sch.timesteps[idx].unsqueeze(0), x,
                                                   noise_level=nl, noise=eps)
    # same torch ops in the same order => bit-exact on CPU
    assert np.array_equal(nxt.numpy(), g["next"])
    assert np.array_equal(mean.numpy(), g["mean"])
    assert np.array_equal(lp.numpy(), g["log_prob"])
    assert np.array_equal(std.reshape(-1).numpy(), g["std"])
    tb = sch.timesteps[idx].repeat(v.shape[0])
    _, lp_r, mean_r, std_r = sde.sde_step_with_logprob(sch, v, tb, x, noise_level=nl,
                                                       prev_sample=nxt.to(torch.bfloat16))
    assert np.array_equal(lp_r.numpy(), g["log_prob_replay"])
    assert np.array_equal(std_r.reshape(-1).numpy(), g["std_replay"])


def test_sampler_bit_exact():
    for c in json.load(open(os.path.join(G, "sampler.json"))):
        got = grouping.k_repeat_indices(c["dataset_len"], c["b"], c["k"], c["n"], c["seed"], c["epoch"])
        assert got == c["per_rank"], c
    # SURVEY 8(a1) known answer
    got = grouping.k_repeat_indices(25432, 1, 2, 8, 42, 0)
    assert got == [[20793], [12678], [12678], [17208], [5139], [5139], [17208], [20793]]


@pytest.mark.parametrize("case", ["toy", "epoch", "zero_std"])
def test_group_advantages_bit_exact(case):
    g = _groups(np.load(os.path.join(G, "stat_tracker.npz")))[case]
    for gs in (0, 1):
        adv = grouping.group_advantages(g["group_ids"], g["rewards"], bool(gs))
        assert adv.dtype == np.float64
        assert np.array_equal(adv, g[f"adv_global{gs}"]), case
    ori = g["rewards"] if g["rewards"].ndim == 1 else g["rewards"][:, 0]
    z = grouping.zero_std_ratio(g["group_ids"], ori)
    assert np.array_equal(np.array(z), g["zero_std"])
    if case == "toy":  # SURVEY 8(a13)/(a14) known answers
        np.testing.assert_allclose(g["adv_global1"], [-1.36618011, -0.87825864, -0.19516859, 0, 0.87825864,
                                                      1.56134869], atol=1e-8)
        np.testing.assert_allclose(g["zero_std"], [1 / 3, 1.1849348892187752], rtol=1e-15)


def test_ungather():
    a = np.arange(24.0).reshape(12, 2)
    assert np.array_equal(grouping.ungather(a, 3, 1), a[4:8])


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_grpo_loss(case):
    g = _groups(np.load(os.path.join(G, "losses.npz")))[f"grpo_{case}"]
    lp = torch.from_numpy(g["log_prob"]).requires_grad_(True)
    loss, info = losses.grpo_loss(lp, torch.from_numpy(g["old"]), torch.from_numpy(g["adv"]), 5, float(g["clip"]))
    loss.backward()
    assert np.array_equal(lp.grad.numpy(), g["grad"])
    for k in ("approx_kl", "clipfrac", "clipfrac_gt_one", "clipfrac_lt_one", "policy_loss", "loss"):
        assert np.array_equal(info[k].detach().numpy(), g[k]), k


def test_grpo_loss_with_kl_term():
    """config.train.beta > 0: the golden was made by exec'ing the trainer's own loss block (TP:1111-1162) with a KL reference
    mean; pins oracle.losses.kl_loss, the composition loss = policy_loss + beta * kl_loss and both gradients."""
    g = _groups(np.load(os.path.join(G, "losses.npz")))["grpo_kl"]
    lp = torch.from_numpy(g["log_prob"]).requires_grad_(True)
    mean = torch.from_numpy(g["mean"]).requires_grad_(True)
    pl, info = losses.grpo_loss(lp, torch.from_numpy(g["old"]), torch.from_numpy(g["adv"]), 5, float(g["clip"]))
    kl = losses.kl_loss(mean, torch.from_numpy(g["mean_ref"]))
    loss = pl + float(g["beta"]) * kl
    loss.backward()
    assert np.array_equal(kl.detach().numpy(), g["kl_loss"]) and np.array_equal(pl.detach().numpy(), g["policy_loss"])
    assert np.array_equal(loss.detach().numpy(), g["loss"])
    assert np.array_equal(lp.grad.numpy(), g["grad_log_prob"]) and np.array_equal(mean.grad.numpy(), g["grad_mean"])
    for k in ("approx_kl", "clipfrac", "clipfrac_gt_one", "clipfrac_lt_one"):
        assert np.array_equal(info[k].detach().numpy(), g[k]), k


def test_clip_pair_loss():
    g = _groups(np.load(os.path.join(G, "losses.npz")))["clip"]
    t, i0, i1 = (torch.from_numpy(g[k]) for k in ("text", "img0", "img1"))
    loss = losses.clip_pair_loss(t, i0, i1, torch.tensor(100.0))
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=1e-6)
    # closed form noted in SURVEY 3.4
    sp = torch.nn.functional.softplus(100.0 * ((t * i1).sum(-1) - (t * i0).sum(-1))).mean()
    np.testing.assert_allclose(sp.numpy(), g["loss"], rtol=1e-5)


def _head_from_flat(flat, D, H):
    head = losses.DinoHead(D, H)
    off = 0
    with torch.no_grad():
        for p in head.parameters():
            n = p.numel(); p.copy_(torch.from_numpy(flat[off:off + n]).reshape(p.shape)); off += n
    return head


def test_dino_hinge_loss_and_adam_step():
    g = _groups(np.load(os.path.join(G, "losses.npz")))["dino"]
    head = _head_from_flat(g["head"], 24, 16)
    opt = torch.optim.Adam(head.parameters(), lr=1e-3, betas=(0.5, 0.999))
    loss, acc = losses.dino_hinge_loss(head, torch.from_numpy(g["feats_real"]), torch.from_numpy(g["feats_fake"]),
                                       torch.from_numpy(g["idx_real"]), torch.from_numpy(g["idx_fake"]))
    np.testing.assert_allclose(loss.item(), g["d_loss"], rtol=1e-6)
    assert acc == float(g["acc"])
    opt.zero_grad(); loss.backward(); opt.step()
    after = np.concatenate([p.detach().numpy().ravel() for p in head.parameters()])
    np.testing.assert_allclose(after, g["head_after"], rtol=1e-5, atol=1e-7)


def test_ema():
    g = _groups(np.load(os.path.join(G, "losses.npz")))["ema"]
    p = [torch.arange(6.0).reshape(2, 3)]
    e = [p[0].clone()]
    for step in range(40):
        p[0] += 0.5
        losses.ema_step(e, p, step, decay=0.9, update_step_interval=8)
        assert losses.ema_decay(step) == g["decay"][step]
        assert np.array_equal(e[0].numpy(), g["params"][step])


def test_rewards_epilogues():
    g = _groups(np.load(os.path.join(G, "rewards.npz")))
    d = g["dino_patch"]
    head = _head_from_flat(d["head"], 24, 16).to(torch.bfloat16)
    feats = torch.from_numpy(d["feats"]).to(torch.bfloat16)
    hyb, cls_s, patch_s = rewards.dino_patch_score(feats, head, torch.from_numpy(d["idx"]))
    # bf16 pipeline, same torch ops: exact
    assert np.array_equal(hyb.float().detach().numpy(), d["scores"])
    pre = rewards.dino_preprocess(torch.from_numpy(d["images"]).to(torch.bfloat16))
    assert np.array_equal(pre.float().numpy()[:, :, ::37, ::37], d["preprocessed"])
    det = rewards.weighted_sum({"a": 0.3, "b": 0.7}, {"a": torch.tensor([0.25, 0.5, 1.0]),
                                                        "b": np.array([3.0, 2.0, 1.0])})
    np.testing.assert_allclose([float(x) for x in det["avg"]], g["multi"]["avg"], rtol=1e-7)


@pytest.mark.parametrize("case,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
def test_rollout_and_replay(case, dtype):
    g = _groups(np.load(os.path.join(G, "rollout.npz")))[case]
    steps, T, Gn, hw = (int(v) for v in g["meta"])
    net = StandinVelocity(); sch = FlowMatchEulerScheduler()
    tt = lambda k: torch.from_numpy(g[k]).to(dtype)
    fn = lambda x, t, c, p: net(x, t, c, p)[0]
    with torch.no_grad():
        image, lats, lps, tss = rollout.rollout(
            fn, standin_vae_decode, sch, prompt_embeds=tt("pe"), pooled_prompt_embeds=tt("ppe"),
            negative_prompt_embeds=tt("npe"), negative_pooled_prompt_embeds=tt("nppe"),
            num_inference_steps=steps, guidance_scale=4.5, height=hw, width=hw, noise_level=0.8,
            mini_num_image_per_prompt=Gn, train_num_steps=T, process_index=0, sample_num_steps=steps,
            random_timestep=0, latents=torch.from_numpy(g["lat0"]), noises=list(torch.from_numpy(g["noises"])))
    lat = torch.stack(lats, 1); lp = torch.stack(lps, 1); ts = torch.stack(tss, 1)
    assert lat.dtype == dtype and lp.dtype == torch.float32
    assert np.array_equal(lat.float().numpy(), g["latents"])
    assert np.array_equal(lp.numpy(), g["log_probs"])
    assert np.array_equal(ts.float().numpy(), g["timesteps"])
    assert np.array_equal(image.float().numpy(), g["image"])
    sample = {"latents": lat[:, :-1], "next_latents": lat[:, 1:], "timesteps": ts}
    embeds = torch.cat([tt("npe").repeat(Gn, 1, 1), tt("pe").repeat(Gn, 1, 1)])
    pooled = torch.cat([tt("nppe").repeat(Gn, 1), tt("ppe").repeat(Gn, 1)])
    for j in range(T):
        with torch.no_grad():
            _, lpj, mj, _ = rollout.compute_log_prob(fn, sch, sample, j, embeds, pooled,
                                                     guidance_scale=4.5, noise_level=0.8)
        assert np.array_equal(lpj.numpy(), g["replay_log_probs"][:, j])
        assert np.array_equal(mj.numpy(), g["replay_mean"][:, j])
