"""GPU: the PRODUCT entry points (not the oracle) against the goldens made from the reference's own functions
(tests/golden/make_golden.py): scorer plugin surface (a8), EMA (a22), group statistics (a14)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _groups(npz):
    names = sorted({k.split("/")[0] for k in npz.files})
    return {n: {k.split("/", 1)[1]: npz[k] for k in npz.files if k.startswith(n + "/")} for n in names}


class _GoldenBackbone:
    """Stand-in for vit.DinoV2 with the product interface: returns the golden's backbone features (the reference golden
    was made the same way, with a stand-in ``scorer`` -- adv_grpo/rewards.py:393-398 only calls forward_features)."""

    def __init__(self, feats):
        self.feats = feats

    def forward_features(self, images=None):
        assert images.is_cuda and images.shape[0] == self.feats.shape[0]
        return self.feats


def test_dino_patch_cotrain_score_product_vs_reference_golden():
    """rewards.multi_score({'dino_patch_cotrain': 1}) -> dino_patch_cotrain_score._fn -> vit.DinoHead.patch_score (HIP
    gather + L2-norm, head GEMM + exact GELU, combine) against the scores the reference's _fn produced (RW:375-434) for
    the same features, head and patch indices.  The reference runs the head in bf16 on the CPU; the kernels keep f32
    accumulators, hence the bf16-resolution tolerance (scores are O(0.1-1))."""
    from adv_grpo_amd import rewards, vit
    d = _groups(np.load(os.path.join(G, "rewards.npz")))["dino_patch"]
    B, T, D = d["feats"].shape
    Hd = 16
    flat = d["head"]
    w1 = torch.from_numpy(flat[:Hd * D]).reshape(Hd, D); b1 = torch.from_numpy(flat[Hd * D:Hd * D + Hd])
    w2 = torch.from_numpy(flat[Hd * D + Hd:Hd * D + 2 * Hd]).reshape(1, Hd); b2 = torch.from_numpy(flat[Hd * D + 2 * Hd:])
    # the MFMA GEMM wants K % 64 == 0: zero columns change neither the L2 norms nor the products
    pad = 64 - D
    feats = torch.nn.functional.pad(torch.from_numpy(d["feats"]).to(torch.bfloat16), (0, pad)).cuda()
    head = vit.DinoHead({"layers.0.weight": torch.nn.functional.pad(w1, (0, pad)), "layers.0.bias": b1,
                         "layers.2.weight": w2, "layers.2.bias": b2}, "cuda")
    fn = rewards.multi_score("cuda", {"dino_patch_cotrain": 1.0})
    images = torch.from_numpy(d["images"]).to(torch.bfloat16).cuda()
    # the reference draws the patch indices from torch's global generator (RW:405); the golden kept them
    idx = torch.from_numpy(d["idx"]).cuda()
    scorer_fn = rewards.dino_patch_cotrain_score("cuda")
    scores, aux = scorer_fn(_GoldenBackbone(feats), head, images, ["p"] * B, [{}] * B, idx=idx)
    np.testing.assert_allclose(scores.cpu().numpy(), d["scores"], atol=2e-2, rtol=2e-2)
    assert aux["patch_scores"].shape == (B, 64) and torch.equal(aux["patch_indices"], idx)
    # through the aggregator: same numbers under 'dino_patch_cotrain' and, with weight 1, under 'avg' (RW:1084-1092)
    torch.manual_seed(55)
    details, _ = fn(images, ["p"] * B, [{}] * B, scorer=_GoldenBackbone(feats), head=head)
    assert set(details) == {"dino_patch_cotrain", "avg"}
    assert torch.equal(torch.as_tensor(details["avg"]).float().cpu(), details["dino_patch_cotrain"].float().cpu())
    # (indices come from the device generator here: only the statistics of the score are comparable)
    assert torch.isfinite(details["dino_patch_cotrain"]).all()


def test_multi_score_weighted_sum_product_vs_reference_golden():
    """rewards.multi_score aggregation (RW:1043-1093) with two plain scorers registered through the plugin registry: a
    factory that takes ``device`` and one that does not (RW:1040), tensor and ndarray scores (RW:1084-1092)."""
    from adv_grpo_amd import rewards
    g = _groups(np.load(os.path.join(G, "rewards.npz")))["multi"]
    seen = {}

    def toy_a(device):
        seen["a"] = device
        return lambda images, prompts, metadata: (torch.tensor([0.25, 0.5, 1.0], device=device), {})

    def toy_b():
        return lambda images, prompts, metadata: (np.array([3.0, 2.0, 1.0]), {})
    rewards.register_scorer("toy_a", toy_a)
    rewards.register_scorer("toy_b", toy_b)
    try:
        det, meta = rewards.multi_score("cuda", {"toy_a": 0.3, "toy_b": 0.7})(None, ["p"] * 3, [{}] * 3)
    finally:
        rewards.score_functions.pop("toy_a"); rewards.score_functions.pop("toy_b")
    assert seen["a"] == "cuda" and meta == {}
    assert list(det) == ["toy_a", "toy_b", "avg"]
    np.testing.assert_allclose([float(x) for x in det["avg"]], g["avg"], rtol=1e-6)
    with pytest.raises(KeyError):
        rewards.multi_score("cuda", {"deqa": 1.0})          # outside the accelerated path: loud, not silent


def test_ema_wrapper_vs_reference_golden():
    """adv_grpo_amd.ema.EMAModuleWrapper (advgrpo_ema_step) through the 40-step sequence recorded from the reference's
    EMAModuleWrapper (adv_grpo/ema.py:33-52; decay warm-up min((1+s)/(10+s), 0.9), update iff (s+1) % 8 == 0): bit-exact."""
    from adv_grpo_amd.ema import EMAModuleWrapper
    g = _groups(np.load(os.path.join(G, "losses.npz")))["ema"]
    p = [torch.arange(6.0).reshape(2, 3).cuda()]
    ema = EMAModuleWrapper(p, decay=0.9, update_step_interval=8, device="cuda")
    assert torch.equal(ema.ema_parameters[0], p[0]) and ema.ema_parameters[0].data_ptr() != p[0].data_ptr()
    for step in range(40):
        p[0] += 0.5
        ema.step(p, step)
        assert ema.get_current_decay(step) == g["decay"][step]
        assert np.array_equal(ema.ema_parameters[0].cpu().numpy(), g["params"][step]), step
    live = p[0].clone()
    ema.copy_ema_to(p, store_temp=True)
    assert np.array_equal(p[0].cpu().numpy(), g["params"][-1])
    ema.copy_temp_to(p)
    assert torch.equal(p[0], live) and ema.temp_stored_parameters is None


def test_lora_model_builds_its_ema_at_construction():
    """TP:528: the EMA starts from the INITIAL trainable parameters (not from the weights after the first update), and
    loading adapters resets it (PeftModel.from_pretrained precedes the wrapper upstream, TP:506-528)."""
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
    from adv_grpo_amd.model_configs import MMDiTConfig
    cfg = MMDiTConfig(num_layers=2, num_heads=4, joint_attention_dim=128, pooled_projection_dim=64, pos_embed_max_size=16,
                      dual_attention_layers=(0,))
    model = SD3TransformerLoRA(synthetic.mmdit_weights(cfg, 3), cfg, "cuda", seed=1)
    assert model.ema is not None and torch.equal(model.ema, model.params) and model.ema.data_ptr() != model.params.data_ptr()
    before = model.ema.clone()
    model.grads.normal_()
    model.optimizer_step(lr=1e-2)
    assert not torch.equal(model.params, before) and torch.equal(model.ema, before)       # untouched until ema_step
    model.ema_step(7)                                                                      # (7 + 1) % 8 == 0 -> update
    d = min(8 / 17, 0.9)
    torch.testing.assert_close(model.ema, before + (1 - d) * (model.params - before), rtol=1e-6, atol=1e-7)
    state = {k: v + 1 for k, v in model.lora_state_dict().items()}
    model.load_lora_state(state)
    assert torch.equal(model.ema, model.params)


@pytest.mark.parametrize("case", ["toy", "epoch", "zero_std"])
def test_zero_std_ratio_vs_reference_golden(case):
    """calculate_zero_std_ratio (TP:195-229) from the group-advantage launch: (zero_std_ratio, reward_std_mean) bit-exact
    with the reference function, float32 rewards in float32 arithmetic like numpy."""
    from adv_grpo_amd.stat_tracking import PerPromptStatTracker, calculate_zero_std_ratio, group_advantage
    g = _groups(np.load(os.path.join(G, "stat_tracker.npz")))[case]
    ori = g["rewards"] if g["rewards"].ndim == 1 else g["rewards"][:, 0]
    ids = torch.from_numpy(g["group_ids"]).cuda()
    stats = calculate_zero_std_ratio(ids, {"ori_avg": torch.from_numpy(ori).cuda()})
    assert np.array_equal(stats.cpu().numpy(), g["zero_std"]), (stats.cpu().numpy(), g["zero_std"])
    # the same numbers ride along with the advantages of the [N, T] launch the trainer makes
    r = torch.from_numpy(g["rewards"]).cuda()
    for gs in (False, True):
        adv, st = group_advantage(r, ids, gs, return_stats=True)
        assert np.array_equal(adv.cpu().numpy(), g[f"adv_global{int(gs)}"])
        assert np.array_equal(st.cpu().numpy(), g["zero_std"])
    tr = PerPromptStatTracker(True)
    tr.update(ids, r)
    assert np.array_equal(tr.last_group_stats.cpu().numpy(), g["zero_std"])
    assert tr.get_stats() == tuple(g["stats1"])
    if case == "toy":      # SURVEY 8(a14) known answer; prompt strings sorted like np.unique
        st = calculate_zero_std_ratio(["a", "b", "a", "c", "b", "a"], np.array([1, 2, 3, 4, 5, 6], dtype=np.float64))
        np.testing.assert_allclose(st.cpu().numpy(), [1 / 3, 1.1849348892187752], rtol=1e-15)
