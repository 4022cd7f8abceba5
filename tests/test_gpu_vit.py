"""GPU parity of the reward towers and their pre/post-processing against the oracle
(oracle/vit.py is pinned against transformers in tests/test_oracle_vit.py)."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def _pil_clip_preprocess(images_u8):
    """Reference CPU path: PIL BICUBIC resize to 224, /255, CLIP normalise (CLIPProcessor, square inputs)."""
    from adv_grpo_amd.preprocess import CLIP_MEAN, CLIP_STD
    out = []
    for im in images_u8:
        r = np.asarray(Image.fromarray(im).resize((224, 224), Image.BICUBIC)).astype(np.float32) * np.float32(1 / 255)
        out.append((r - np.array(CLIP_MEAN, dtype=np.float32)) / np.array(CLIP_STD, dtype=np.float32))
    return torch.from_numpy(np.stack(out)).permute(0, 3, 1, 2).contiguous()


def _unpatch(patches, B, size):
    g = size // 14
    x = patches.float().view(B, g, g, 640)[..., :588].reshape(B, g, g, 3, 14, 14)
    return x.permute(0, 3, 1, 4, 2, 5).reshape(B, 3, size, size)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_clip_preprocess_is_pil_exact(dt):
    from adv_grpo_amd import preprocess
    from oracle.rewards import to_uint8
    g = torch.Generator().manual_seed(0)
    img = torch.rand(3, 3, 512, 512, generator=g).to(dt)
    patches = preprocess.clip_patches(img.cuda(), 224)
    u8 = to_uint8(img).permute(0, 2, 3, 1).numpy()        # rewards.py:567-569 (same dtype as the reference path)
    ref = _pil_clip_preprocess(u8)
    got = _unpatch(patches, 3, 224).cpu()
    # identical uint8 resize => only the final bf16 rounding of the normalised pixel differs
    assert torch.equal(got, ref.to(torch.bfloat16).float())


def test_dino_preprocess_matches_torch_bicubic():
    from adv_grpo_amd import preprocess
    from oracle.rewards import dino_preprocess
    g = torch.Generator().manual_seed(1)
    img = torch.rand(2, 3, 512, 512, generator=g).to(torch.bfloat16)
    got = _unpatch(preprocess.dino_patches(img.cuda(), 518), 2, 518).cpu()
    ref = dino_preprocess(img, cuda_semantics=True).float()
    err = (got - ref).abs()
    # the 16-tap bicubic sum is accumulated in a different order than torch's: where the f32 sums straddle
    # a bf16 rounding boundary the interpolated pixel flips by one bf16 ulp (<= 2^-8), which /std and the
    # final bf16 rounding turn into <= 0.06 on the normalised pixel; it must stay a rare event.
    frac = (err > 0).float().mean().item()
    print("dino preprocess mismatch fraction", frac, "max", err.max().item())
    assert err.max().item() <= 0.06 and frac < 0.005 and err.mean().item() < 1e-4


def test_clip_towers_and_pickscore_vs_oracle():
    from adv_grpo_amd import synthetic, vit
    from oracle import rewards as o_rw
    from oracle import vit as o
    cfg = o.ClipConfig(v_layers=3, t_layers=3)                  # full ViT-H widths (1280/80-dim heads, 1024), 3 layers
    W = synthetic.clip_weights(cfg, 5)
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    model = vit.CLIPModel(Wb, cfg, "cuda")
    g = torch.Generator().manual_seed(2)
    img = torch.rand(4, 3, 512, 512, generator=g).to(torch.bfloat16)
    ids = synthetic.clip_input_ids(4, 3)
    ie = model.get_image_features(images=img.cuda())
    te = model.get_text_features(ids)
    W32 = {k: v.float().cuda() for k, v in Wb.items()}
    px = _pil_clip_preprocess(o_rw.to_uint8(img).permute(0, 2, 3, 1).numpy()).to(torch.bfloat16).float().cuda()
    ri = o.clip_image_features(W32, cfg, px)
    rt = o.clip_text_features(W32, cfg, ids.cuda())
    assert _rel(ie, ri) < 2e-2 and _rel(te, rt) < 2e-2, (_rel(ie, ri), _rel(te, rt))
    s = vit.pickscore_scores(ie, te, Wb["logit_scale"].float())
    rs = o_rw.pickscore_from_embeddings(ri, rt, W32["logit_scale"])
    # <= 1e-3 abs on the /26 score scale would need ~3e-4 cosine accuracy; bf16 towers give ~1e-2 relative
    assert (s.cpu() - rs.cpu()).abs().max().item() < 2e-2 * max(1.0, rs.abs().max().item())


def test_clip_text_pooling_with_legacy_eos_token_id_2():
    """with the released PickScore_v1 config (text_config.eos_token_id = 2) the text towers pool at argmax(input_ids) like
    transformers does, not at the first id-2 token -- otherwise every prompt would pool at position 0 and share one embedding."""
    import dataclasses
    from adv_grpo_amd import synthetic, vit, vit_x3
    from oracle import vit as o
    cfg = o.ClipConfig(v_layers=1, t_layers=2)
    W = synthetic.clip_weights(cfg, 6)
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    ids = synthetic.clip_input_ids(4, 9)                # eos 49407 at a random position, pad after
    legacy = dataclasses.replace(cfg, eos_token_id=2)
    W32 = {k: v.float().cuda() for k, v in Wb.items()}
    ref = o.clip_text_features(W32, legacy, ids.cuda())
    assert torch.allclose(ref, o.clip_text_features(W32, cfg, ids.cuda()))          # same position either way for these ids
    te = vit.CLIPModel(Wb, legacy, "cuda").get_text_features(ids)
    assert _rel(te, ref) < 2e-2
    assert (te[0] - te[1]).abs().max().item() > 1e-3                               # prompts differ: not all pooled at BOS
    te3 = vit_x3.CLIPModelX3(W, legacy, "cuda").get_text_features(ids)               # the fp32-equivalent scorer (unrounded f32 weights)
    assert _rel(te3, o.clip_text_features({k: v.float().cuda() for k, v in W.items()}, legacy, ids.cuda())) < 5e-5


def test_clip_towers_fp32_equivalent_vs_oracle():
    """The fp32 scorer of rewards.py:561-574 (PickScoreScorer(dtype=torch.float32)) on split-bf16 products (vit_x3.py): full
    ViT-H / text widths, 3 layers, fp32 weights, against the fp32 oracle on PIL-preprocessed pixels.  Tolerance: each of the
    ~25 matrix products between pixels and features is good to ~2^-16; features are held to 5e-5 (measured 1e-5), the score (logit scale
    100 on a cosine, / 26) to 1e-4 (measured 3e-6) -- a tenth of SURVEY 7's 1e-3."""
    from adv_grpo_amd import synthetic, vit_x3
    from adv_grpo_amd.pickscore_scorer import PickScoreScorer
    from oracle import rewards as o_rw
    from oracle import vit as o
    cfg = o.ClipConfig(v_layers=3, t_layers=3)
    W = synthetic.clip_weights(cfg, 5)                                    # f32 weights, used unrounded
    scorer = PickScoreScorer("cuda", dtype=torch.float32, model_sd=W, clip_cfg=cfg)
    assert scorer.compute_dtype == "bf16x3" and isinstance(scorer.model, vit_x3.CLIPModelX3)
    g = torch.Generator().manual_seed(2)
    img = torch.rand(4, 3, 512, 512, generator=g)
    ids = synthetic.clip_input_ids(4, 3)
    ie = scorer.model.get_image_features(images=img.cuda())
    te = scorer.model.get_text_features(ids)
    assert ie.dtype == torch.float32 and te.dtype == torch.float32
    W32 = {k: v.float().cuda() for k, v in W.items()}
    px = _pil_clip_preprocess(o_rw.to_uint8(img).permute(0, 2, 3, 1).numpy()).cuda()
    ri = o.clip_image_features(W32, cfg, px)
    rt = o.clip_text_features(W32, cfg, ids.cuda())
    print("bf16x3 CLIP towers vs fp32 oracle: image", _rel(ie, ri), "text", _rel(te, rt))
    assert _rel(ie, ri) < 5e-5 and _rel(te, rt) < 5e-5
    s = scorer(ids, img.cuda())
    rs = o_rw.pickscore_from_embeddings(ri, rt, W32["logit_scale"])
    print("scores", s.tolist(), "oracle", rs.tolist())
    assert (s.cpu() - rs.cpu()).abs().max().item() < 1e-4
    # and the bf16 scorer of the co-training loop (TP:514) is the other arithmetic
    assert PickScoreScorer("cuda", dtype=torch.bfloat16, model_sd=W, clip_cfg=cfg).compute_dtype == "bf16"


def test_clip_vit_h_full_depth_fp32_equivalent_image_tower():
    """All 32 layers of the ViT-H image tower in the fp32-equivalent arithmetic: the error must not compound past 1e-4 (measured 1.1e-5)."""
    from adv_grpo_amd import synthetic, vit_x3
    from oracle import vit as o
    cfg = o.ClipConfig(t_layers=1)
    W = synthetic.clip_weights(cfg, 6)
    model = vit_x3.CLIPModelX3(W, cfg, "cuda")
    g = torch.Generator().manual_seed(3)
    img = torch.rand(2, 3, 224, 224, generator=g)
    from oracle import rewards as o_rw
    px = _pil_clip_preprocess(o_rw.to_uint8(img).permute(0, 2, 3, 1).numpy()).cuda()
    ie = model.get_image_features(images=img.cuda())
    ri = o.clip_image_features({k: v.float().cuda() for k, v in W.items()}, cfg, px)
    print("ViT-H 32 layers, bf16x3 vs fp32 oracle:", _rel(ie, ri))
    assert _rel(ie, ri) < 1e-4


def test_clip_vit_h_full_depth_image_tower():
    from adv_grpo_amd import synthetic, vit
    from oracle import vit as o
    cfg = o.ClipConfig(t_layers=1)
    W = synthetic.clip_weights(cfg, 6)
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    model = vit.CLIPModel(Wb, cfg, "cuda")
    g = torch.Generator().manual_seed(3)
    px = torch.randn(2, 3, 224, 224, generator=g).to(torch.bfloat16)
    gsz = 16
    patches = torch.zeros(2 * 256, 640, dtype=torch.bfloat16)
    patches[:, :588] = px.view(2, 3, gsz, 14, gsz, 14).permute(0, 2, 4, 1, 3, 5).reshape(2 * 256, 588)
    ie = model.get_image_features(pixel_patches=patches.cuda())
    ri = o.clip_image_features({k: v.float().cuda() for k, v in Wb.items()}, cfg, px.float().cuda())
    assert _rel(ie, ri) < 3e-2, _rel(ie, ri)


def test_dinov2_fp32_equivalent_tower_and_image_similarity_vs_oracle():
    """image_similarity_score's tower in the reference's fp32 arithmetic (rewards.py:147-203): full ViT-B/14 @ 518 (12 layers,
    1370 tokens, LayerScale) on split-bf16 products against the fp32 oracle fed by torch's own f32 bicubic resize; then the
    scorer end to end (max over the reference images of the CLS cosine)."""
    import torch.nn.functional as F
    from adv_grpo_amd import rewards, synthetic, vit_x3
    from adv_grpo_amd.preprocess import IMAGENET_MEAN, IMAGENET_STD
    from oracle import vit as o
    cfg = o.DinoConfig()
    W = synthetic.dino_weights(cfg, 7)
    model = vit_x3.DinoV2X3(W, cfg, "cuda")
    g = torch.Generator().manual_seed(4)
    img, ref_img = torch.rand(3, 3, 512, 512, generator=g), torch.rand(2, 3, 384, 384, generator=g)

    def pre(x):                                                 # RW:151-170
        x = F.interpolate(x.cuda(), size=(518, 518), mode="bicubic", align_corners=False)
        m = torch.tensor(IMAGENET_MEAN, device="cuda")[None, :, None, None]
        s = torch.tensor(IMAGENET_STD, device="cuda")[None, :, None, None]
        return (x - m) / s
    W32 = {k: v.float().cuda() for k, v in W.items()}
    feats = model.forward_features(images=img.cuda())
    want = o.dino_forward_features(W32, cfg, pre(img))
    assert feats.shape == (3, 1370, 768) and feats.dtype == torch.float32
    print("DINOv2 bf16x3 vs fp32 oracle:", _rel(feats, want))
    assert _rel(feats, want) < 1e-4
    rewards.configure_dino(model)
    try:
        s, info = rewards.image_similarity_score("cuda")(img, ref_img)
    finally:
        rewards.configure_dino(None)
    a = want[:, 0] / want[:, 0].norm(dim=-1, keepdim=True)
    b = o.dino_forward_features(W32, cfg, pre(ref_img))[:, 0]
    b = b / b.norm(dim=-1, keepdim=True)
    pair = a @ b.T
    assert (info["pairwise"] - pair).abs().max().item() < 1e-4 and (s - pair.max(dim=1).values).abs().max().item() < 1e-4


def test_dinov2_features_and_patch_head_vs_oracle():
    from adv_grpo_amd import synthetic, vit
    from oracle import losses as o_l
    from oracle import rewards as o_rw
    from oracle import vit as o
    cfg = o.DinoConfig()                                        # full ViT-B/14 @ 518: 1370 tokens, 12 layers
    W = synthetic.dino_weights(cfg, 7)
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    model = vit.DinoV2(Wb, cfg, "cuda")
    g = torch.Generator().manual_seed(4)
    img = torch.rand(2, 3, 512, 512, generator=g).to(torch.bfloat16)
    feats = model.forward_features(images=img.cuda())
    ref = o.dino_forward_features({k: v.float().cuda() for k, v in Wb.items()}, cfg, o_rw.dino_preprocess(img, cuda_semantics=True).float().cuda())
    assert feats.shape == (2, 1370, 768)
    assert _rel(feats, ref) < 2e-2, _rel(feats, ref)
    # head epilogue on identical (bf16) features and indices
    hw = synthetic.dino_head_weights(768, 512, 8)
    head = vit.DinoHead(hw, "cuda")
    idx = torch.randint(0, 1369, (2, 64), generator=g)
    hyb, cls, pat = head.patch_score(feats, idx.cuda())
    oh = o_l.DinoHead(768, 512)
    oh.load_state_dict({k: v for k, v in hw.items()})
    oh = oh.to(torch.bfloat16)
    rh, rc, rp = o_rw.dino_patch_score(feats.cpu(), oh, idx)
    assert (cls.cpu() - rc.float()).abs().max().item() < 3e-2 * max(1.0, rc.float().abs().max().item())
    assert (hyb.cpu() - rh.float()).abs().max().item() < 3e-2 * max(1.0, rh.float().abs().max().item())
    assert (pat.cpu() - rp.float()).abs().max().item() < 5e-2 * max(1.0, rp.float().abs().max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_pickscore_text_tower_runs_once_per_distinct_prompt(dtype):
    """A GRPO group scores G images of ONE prompt: PickScoreScorer finds equal prompt STRINGS on the host (rewards.PromptBatch)
    and runs the text tower on the distinct ones only -- scores bit-identical to the full batch, in both arithmetics, for one
    repeated prompt and for a mixed batch (a, b, a, c, b, a)."""
    from adv_grpo_amd import rewards, synthetic
    from adv_grpo_amd.model_configs import ClipConfig
    from adv_grpo_amd.pickscore_scorer import PickScoreScorer
    cfg = ClipConfig(v_layers=2, t_layers=3)
    W = synthetic.clip_weights(cfg, 5)
    scorer = PickScoreScorer("cuda", dtype=dtype, model_sd=W, clip_cfg=cfg)
    calls = []
    full = scorer.model.get_text_features
    scorer.model.get_text_features = lambda ids: (calls.append(ids.shape[0]), full(ids))[1]
    imgs = torch.rand(6, 3, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()
    uniq_ids = synthetic.clip_input_ids(3, 9).cuda()
    for texts, pick in ((["a"] * 6, [0] * 6), (["a", "b", "a", "c", "b", "a"], [0, 1, 0, 2, 1, 0])):
        ids = uniq_ids[pick]
        calls.clear()
        s_dedup = scorer(rewards.PromptBatch(texts, clip_ids=ids), imgs)
        assert calls == [len(set(texts))]
        calls.clear()
        s_full = scorer(ids, imgs)                       # bare ids: nothing to compare on the host, the whole batch runs
        assert calls == [6] and torch.equal(s_dedup, s_full)


def test_vit_stack_through_the_c_entry_is_bit_identical_to_the_python_sequencing():
    """advgrpo_vit_forward (csrc/vit_encoder.cpp) runs the same launches in the same order as vit._Encoder's Python loop: CLIP ViT-H
    width (head dim 80), the causal CLIP text tower (head dim 64) and DINOv2 (LayerScale as the gate operand), bit for bit."""
    from adv_grpo_amd import synthetic, vit
    from adv_grpo_amd.model_configs import DinoConfig
    from oracle import vit as o
    cfg = o.ClipConfig(v_layers=3, t_layers=2)
    clip = vit.CLIPModel({k: v.to(torch.bfloat16) for k, v in synthetic.clip_weights(cfg, 5).items()}, cfg, "cuda")
    g = torch.Generator().manual_seed(4)
    img = torch.rand(3, 3, 512, 512, generator=g).to(torch.bfloat16).cuda()
    ids = synthetic.clip_input_ids(3, 7)
    dcfg = DinoConfig(layers=2)
    dino = vit.DinoV2(synthetic.dino_weights(dcfg, 8), dcfg, "cuda")
    outs = {}
    for c_stack in (True, False):
        vit._Encoder.c_stack = c_stack
        try:
            outs[c_stack] = (clip.get_image_features(images=img), clip.get_text_features(ids), dino.forward_features(images=img[:2]))
        finally:
            vit._Encoder.c_stack = True
    for a, b in zip(outs[True], outs[False]):
        assert a.shape == b.shape and torch.equal(a, b)
    assert torch.isfinite(outs[True][2].float()).all()


def test_vit_c_entry_refuses_what_it_does_not_implement():
    import ctypes
    from adv_grpo_amd import _lib
    lib = _lib.load()
    d = _lib.VitDesc()
    d.B, d.S, d.D, d.H, d.mlp, d.n_layers, d.act, d.causal, d.eps = 1, 16, 96, 1, 128, 0, 2, 0, 1e-5      # head dim 96
    x = torch.zeros(16, 96, dtype=torch.bfloat16, device="cuda")
    ws = torch.empty(int(lib.advgrpo_vit_workspace_bytes(1, 16, 96, 128)), dtype=torch.uint8, device="cuda")
    d.x = x.data_ptr()
    assert lib.advgrpo_vit_forward(ctypes.byref(d), ws.data_ptr(), ws.numel(), None) != 0
    assert b"head dim" in lib.advgrpo_last_error()
