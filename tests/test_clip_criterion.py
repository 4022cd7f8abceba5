"""The reference's ``CLIPCriterion(CLIPCriterionConfig())(model, batch)`` face (adv_grpo/pick_score_training.py:76-87, 89-224; the call is
scripts/train_sd3_fast_pickscore.py:177).  CPU: config fields, patch-row plumbing, refusals.  GPU: the criterion kernel with the batch's
labels against goldens made by the reference's own ``forward`` / ``calc_loss`` (tests/golden/make_golden.py, ``clip/*`` and ``clipl/*``),
and the whole face -- forward, ``zero_grad(); loss.backward(); step`` order -- against the D-step it wraps."""
import dataclasses
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _groups(npz, prefix):
    return {k[len(prefix) + 1:]: npz[k] for k in npz.files if k.startswith(prefix + "/")}


def test_config_fields_are_the_references():
    from adv_grpo_amd.pick_score_training import CLIPCriterionConfig
    f = {x.name: x.default for x in dataclasses.fields(CLIPCriterionConfig)}
    assert f == {"_target_": "trainer.criterions.clip_criterion.CLIPCriterion", "is_distributed": False,
                 "label_0_column_name": "label_0", "label_1_column_name": "label_1", "input_ids_column_name": "input_ids",
                 "pixels_0_column_name": "pixels_0", "pixels_1_column_name": "pixels_1",
                 "num_examples_per_prompt_column_name": "num_examples_per_prompt", "in_batch_negatives": False}


def test_unsupported_modes_are_refused_by_name():
    from adv_grpo_amd.pick_score_training import CLIPCriterion, CLIPCriterionConfig
    with pytest.raises(NotImplementedError, match="in_batch_negatives"):
        CLIPCriterion(CLIPCriterionConfig(in_batch_negatives=True))
    with pytest.raises(NotImplementedError, match="is_distributed"):
        CLIPCriterion(CLIPCriterionConfig(is_distributed=True))


def test_patch_rows_are_the_conv_weights_flatten_order():
    """rows @ conv_weight.reshape(D, -1).T == Conv2d(stride = kernel = 14) of the pixel values."""
    from adv_grpo_amd.pick_score_training import patch_rows
    g = torch.Generator().manual_seed(0)
    px = torch.randn(3, 3, 56, 56, generator=g).to(torch.bfloat16).float()
    w = torch.randn(8, 3, 14, 14, generator=g)
    rows = patch_rows(px)
    assert rows.shape == (3 * 16, 640) and rows.dtype == torch.bfloat16 and (rows[:, 588:] == 0).all()
    ref = torch.nn.functional.conv2d(px, w, stride=14).flatten(2).transpose(1, 2).reshape(3 * 16, 8)
    torch.testing.assert_close(rows[:, :588].float() @ w.reshape(8, -1).T, ref, rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError):
        patch_rows(torch.zeros(1, 3, 50, 56))


@pytest.mark.gpu
def test_criterion_kernel_with_labels_vs_reference_goldens():
    """advgrpo_clip_pair_loss_labels against CLIPCriterion.forward run by the reference on bf16-exact features (``clipl/*``: labels
    (1, 0), (0, 1), a tie, per-example) -- f32 arithmetic on both sides, rtol 2e-5 -- and its gradient against autograd of the oracle's
    restatement of calc_loss generalised to labels; the original (1, 0) entry point on the same data gives the same bits."""
    from adv_grpo_amd import _lib
    lib = _lib.load()
    g = _groups(np.load(os.path.join(G, "losses.npz")), "clipl")
    t, i0, i1 = (torch.from_numpy(g[k]) for k in ("text", "img0", "img1"))
    B, P = t.shape
    s = float(g["logit_scale_exp"])
    e = torch.cat([i0, i1]).to(torch.bfloat16).cuda()
    tt = t.to(torch.bfloat16).cuda()
    assert torch.equal(e.float().cpu(), torch.cat([i0, i1])) and torch.equal(tt.float().cpu(), t)      # bf16-exact by construction

    def run(l0, l1):
        loss = torch.empty(1, dtype=torch.float32, device="cuda")
        de = torch.empty_like(e)
        _lib.check(lib.advgrpo_clip_pair_loss_labels(e.data_ptr(), tt.data_ptr(), B, P, s, _lib.ptr(l0), _lib.ptr(l1), loss.data_ptr(),
                                                     de.data_ptr(), _lib.stream_ptr()))
        return loss.item(), de.float().cpu()

    def autograd(l0, l1):
        ee = torch.cat([i0, i1]).double().requires_grad_(True)
        n = lambda x: x / x.norm(dim=-1, keepdim=True)
        z = s * ((n(t.double()) * n(ee[B:])).sum(-1) - (n(t.double()) * n(ee[:B])).sum(-1))
        sp = torch.nn.functional.softplus
        loss = (l0.double() * sp(z) + l1.double() * sp(-z) + (l0 == l1).double() * np.log(0.5)).mean()
        loss.backward()
        return loss.item(), ee.grad.float()

    for name in ("real_fake", "fake_real", "tie", "mixed"):
        l0 = torch.from_numpy(np.broadcast_to(g[f"{name}/label_0"], (B,)).copy()).float()
        l1 = torch.from_numpy(np.broadcast_to(g[f"{name}/label_1"], (B,)).copy()).float()
        loss, de = run(l0.cuda(), l1.cuda())
        np.testing.assert_allclose(loss, float(g[f"{name}/loss"]), rtol=2e-5, atol=1e-6, err_msg=name)
        ref_loss, ref_de = autograd(l0, l1)
        np.testing.assert_allclose(loss, ref_loss, rtol=2e-5, atol=1e-6)
        assert (de - ref_de).norm() <= 6e-3 * ref_de.norm() + 1e-7, name            # de is rounded to bf16
    a, da = run(None, None)
    loss0 = torch.empty(1, dtype=torch.float32, device="cuda")
    de0 = torch.empty_like(e)
    _lib.check(lib.advgrpo_clip_pair_loss(e.data_ptr(), tt.data_ptr(), B, P, s, loss0.data_ptr(), de0.data_ptr(), _lib.stream_ptr()))
    ones, zeros = torch.ones(B, device="cuda"), torch.zeros(B, device="cuda")
    b, db = run(ones, zeros)
    # (the loss is a sum of per-pair atomics: equal up to the order of B float additions; the gradient rows are written once each)
    assert abs(a - loss0.item()) <= 1e-6 * abs(a) and abs(a - b) <= 1e-6 * abs(a)
    assert torch.equal(da, de0.float().cpu()) and torch.equal(da, db)
    # the f32 normalised golden of the reference's calc_loss (``clip/*``; features not bf16-exact: tolerance of the operand rounding)
    c = _groups(np.load(os.path.join(G, "losses.npz")), "clip")
    e2 = torch.cat([torch.from_numpy(c["img0"]), torch.from_numpy(c["img1"])]).to(torch.bfloat16).cuda()
    t2 = torch.from_numpy(c["text"]).to(torch.bfloat16).cuda()
    loss = torch.empty(1, dtype=torch.float32, device="cuda")
    de2 = torch.empty_like(e2)
    _lib.check(lib.advgrpo_clip_pair_loss_labels(e2.data_ptr(), t2.data_ptr(), 6, 32, 100.0, None, None, loss.data_ptr(), de2.data_ptr(),
                                                 _lib.stream_ptr()))
    assert abs(loss.item() - float(c["loss"])) < 0.05 * max(1.0, float(c["loss"]))


def _toy():
    from adv_grpo_amd import synthetic, vit
    from oracle import vit as o
    cfg = o.ClipConfig(v_hidden=320, v_layers=3, v_heads=4, v_mlp=640, image_size=56, t_hidden=128, t_layers=2, t_heads=2,
                       t_mlp=256, vocab=1000, proj=128, eos_token_id=999)
    W = {k: v.to(torch.bfloat16) for k, v in synthetic.clip_weights(cfg, 12).items()}
    g = torch.Generator().manual_seed(5)
    B = 6
    px = torch.randn(2 * B, 3, 56, 56, generator=g).to(torch.bfloat16)
    ids = torch.randint(1, 990, (B, 77), generator=g); ids[:, 20] = 999
    return cfg, W, px, ids, B, lambda: vit.CLIPModel(W, cfg, "cuda")


@pytest.mark.gpu
def test_criterion_face_is_the_d_step_behind_the_references_call():
    """criterion(scorer.model, batch) -> zero_grad -> loss.backward() -> step, with the reference's batch keys (TP:164-177): the loss and
    the parameters after the step are those of the D-step the trainer runs (d_step_pickscore.loss_and_grads + adam_step) on the
    same pixels up to the order of the D-step's own atomic sums (bias / LayerNorm / split-K gradients); gradients only reach the accumulator at backward(); a second backward() and a backward() on a view-less model raise."""
    from adv_grpo_amd import _lib
    from adv_grpo_amd.d_step_pickscore import ClipLastLayerTrainable, ClipLayersTrainable
    from adv_grpo_amd.pick_score_training import CLIPCriterion, CLIPCriterionConfig, patch_rows
    cfg, W, px, ids, B, make = _toy()
    crit = CLIPCriterion(CLIPCriterionConfig())
    for view in (ClipLastLayerTrainable, lambda m: ClipLayersTrainable(m, -2)):
        m_ref = make(); tr_ref = view(m_ref)
        loss_ref = tr_ref.loss_and_grads(patch_rows(px.cuda()), ids)
        g_ref = tr_ref.grads.clone()
        tr_ref.adam_step(1e-3)
        model = make(); tr = view(model)
        batch = {"input_ids": ids.cuda(), "pixels_0": px[:B].float().cuda(), "pixels_1": px[B:].float().cuda(),
                 "label_0": torch.tensor(1.0).cuda(), "label_1": torch.tensor(0.0).cuda(), "num_examples_per_prompt": torch.tensor(1.0).cuda()}
        loss = crit(model, batch)                       # handed scorer.model, as the reference does
        assert (tr.grads == 0).all()                    # nothing reaches the accumulator before backward()
        tr.grads.zero_()                                # optimizer.zero_grad()
        loss.backward()
        assert abs(loss.item() - loss_ref.item()) <= 1e-6 * abs(loss_ref.item())
        assert (tr.grads - g_ref).norm() <= 1e-5 * g_ref.norm()
        with pytest.raises(_lib.AdvGrpoError, match="twice"):
            loss.backward()
        tr.adam_step(1e-3)
        d = (tr.params - tr_ref.params).abs()                                 # one Adam step of lr 1e-3 moves an entry by <= lr: only entries whose
        assert d.max() <= 2.5e-3 and (d > 1e-5).float().mean() < 1e-2        # gradient is ~0 (its sign is rounding noise) may differ visibly
        assert abs(crit(tr, batch).item() - crit(model, batch).item()) <= 1e-6 * abs(loss.item())      # the view itself is accepted too
    # labels other than (1, 0) go through the whole face: a tie's loss is label-symmetric and its gradient differs from (1, 0)'s
    batch["label_0"], batch["label_1"] = torch.tensor(0.5), torch.tensor(0.5)
    tie = crit(model, batch)
    assert np.isfinite(tie.item()) and tie.item() != loss.item()
    # forward only on a model without a trainable view
    plain = make()
    batch["label_0"], batch["label_1"] = torch.tensor(1.0), torch.tensor(0.0)
    fo = crit(plain, batch)
    assert abs(fo.item() - loss_ref.item()) < 0.05 * max(1.0, abs(loss_ref.item()))
    with pytest.raises(_lib.AdvGrpoError, match="trainable view"):
        fo.backward()
    with pytest.raises(KeyError):
        crit(model, {k: v for k, v in batch.items() if k != "num_examples_per_prompt"})
