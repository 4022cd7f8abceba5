"""GPU: the rollout function (fused SDE kernel, same control flow / cast order / draw order) against the
golden trajectory produced by the REFERENCE's pipeline_with_logprob_random + compute_log_prob on a stand-in
velocity network (tests/golden/rollout.npz), and the end-to-end path on a reduced-depth SD3 stack."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

E2E_SCORE_TOL = 1.5e-3   # end-to-end reward tolerance on the score scale: measured 6.0e-4 (printed by the test) x 2.5
G = os.path.join(os.path.dirname(__file__), "golden")


def _groups(npz):
    names = sorted({k.split("/")[0] for k in npz.files})
    return {n: {k.split("/", 1)[1]: npz[k] for k in npz.files if k.startswith(n + "/")} for n in names}


class _StandinVae:
    """Same arithmetic as the stand-in used for the golden: rescale, stand-in decode, postprocess."""

    def decode_to_image(self, latents):
        from oracle.standin import standin_vae_decode
        z = (latents / 1.5305) + 0.0609
        return (standin_vae_decode(z.to(torch.float32)) / 2 + 0.5).clamp(0, 1)


@pytest.mark.parametrize("case,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
def test_rollout_matches_reference_trajectory(case, dtype):
    from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
    from adv_grpo_amd.diffusers_patch.sd3_sde_with_logprob import sde_step_cfg
    from adv_grpo_amd.pipeline import SD3Pipeline
    from oracle.standin import StandinVelocity
    g = _groups(np.load(os.path.join(G, "rollout.npz")))[case]
    steps, T, Gn, hw = (int(v) for v in g["meta"])
    net = StandinVelocity().cuda()
    net.config = type("C", (), {"in_channels": 16})()
    pipe = SD3Pipeline(net, _StandinVae(), "cuda")
    tt = lambda k: torch.from_numpy(g[k]).to(dtype).cuda()
    image, lats, lps, tss = pipeline_with_logprob_random(
        pipe, prompt_embeds=tt("pe"), pooled_prompt_embeds=tt("ppe"), negative_prompt_embeds=tt("npe"),
        negative_pooled_prompt_embeds=tt("nppe"), num_inference_steps=steps, guidance_scale=4.5, output_type="pt",
        height=hw, width=hw, noise_level=0.8, mini_num_image_per_prompt=Gn, train_num_steps=T, process_index=0,
        sample_num_steps=steps, random_timestep=0, latents=torch.from_numpy(g["lat0"]).cuda(),
        noises=list(torch.from_numpy(g["noises"]).cuda()))
    lat = torch.stack(lats, 1); lp = torch.stack(lps, 1); ts = torch.stack(tss, 1)
    assert lat.dtype == dtype and lp.dtype == torch.float32 and lat.shape == g["latents"].shape
    assert np.array_equal(ts.float().cpu().numpy(), g["timesteps"])
    # the stand-in network runs through torch-on-GPU (tanh/einsum differ from the CPU in the last ulp), so the
    # trajectory is compared at one ulp of the storage dtype; the SDE arithmetic itself is bit-exact (test_gpu_leaf_kernels)
    tol = 2e-5 if dtype == torch.float32 else 2 ** -7
    assert np.allclose(lat.float().cpu().numpy(), g["latents"], atol=tol * 4, rtol=tol)
    assert np.allclose(lp.cpu().numpy(), g["log_probs"], rtol=5e-3 if dtype == torch.bfloat16 else 1e-4)
    assert np.allclose(image.float().cpu().numpy(), g["image"], atol=2e-2 if dtype == torch.bfloat16 else 1e-4)
    # replay (compute_log_prob, TP:233-267) from the REFERENCE's stored trajectory: ratio/clipfrac inputs
    ref_lat = torch.from_numpy(g["latents"]).to(dtype).cuda()
    embeds = torch.cat([tt("npe").repeat(Gn, 1, 1), tt("pe").repeat(Gn, 1, 1)])
    pooled = torch.cat([tt("nppe").repeat(Gn, 1), tt("ppe").repeat(Gn, 1)])
    for j in range(T):
        x = ref_lat[:, j]
        v = net(torch.cat([x] * 2), torch.from_numpy(g["timesteps"][:, j]).cuda().repeat(2), embeds, pooled)[0]
        _, _, lpj, mean, _ = sde_step_cfg(pipe.scheduler, v[:Gn], v[Gn:], 4.5, None, x, 0.8,
                                         prev_sample=ref_lat[:, j + 1], step_index=j)
        assert np.allclose(mean.cpu().numpy(), g["replay_mean"][:, j], atol=tol * 8, rtol=tol)
        assert np.allclose(lpj.cpu().numpy(), g["replay_log_probs"][:, j], rtol=2e-2 if dtype == torch.bfloat16 else 1e-4)


@pytest.mark.parametrize("qk_scale", [1.0, 3.0])
def test_end_to_end_sample_decode_score_small_stack(qk_scale):
    """Whole hot path (MMDiT rollout -> VAE decode -> PickScore) on a reduced-depth stack, HIP vs the fp32
    oracle driven with the same injected noise: log-probs, image and rewards.  qk_scale = 3: the q / k RMSNorm weights of every
    attention x 3, i.e. attention logits with a standard deviation of ~9 instead of the ~1 random weights give -- the peaked
    softmax rows of a trained MMDiT -- under the SAME tolerances."""
    from adv_grpo_amd import synthetic, vit
    from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
    from adv_grpo_amd.mmdit import SD3Transformer2DModel
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    from oracle import mmdit as o_m
    from oracle import rewards as o_rw
    from oracle import rollout as o_r
    from oracle import vae as o_v
    from oracle import vit as o_t
    from oracle.scheduler import FlowMatchEulerScheduler
    from tests.test_gpu_vit import _pil_clip_preprocess
    mcfg = o_m.MMDiTConfig(num_layers=3, num_heads=6, joint_attention_dim=256, pooled_projection_dim=128,
                           pos_embed_max_size=96, dual_attention_layers=(0, 1))
    vcfg = o_v.VaeConfig()
    ccfg = o_t.ClipConfig(v_layers=2, t_layers=2)
    Wm = {k: ((v * qk_scale) if (".norm_q." in k or ".norm_k." in k or ".norm_added_" in k) else v).to(torch.bfloat16)
          for k, v in synthetic.mmdit_weights(mcfg, 21).items()}
    Wv = {k: v.to(torch.bfloat16) for k, v in synthetic.vae_decoder_weights(vcfg, 22).items()}
    Wc = {k: v.to(torch.bfloat16) for k, v in synthetic.clip_weights(ccfg, 23).items()}
    pipe = SD3Pipeline(SD3Transformer2DModel(Wm, mcfg, "cuda"), AutoencoderKLDecoder(Wv, vcfg, "cuda"), "cuda")
    clip = vit.CLIPModel(Wc, ccfg, "cuda")
    gen = torch.Generator().manual_seed(9)
    Gn, hw, steps, T = 2, 256, 4, 2                                        # BASELINE config C1 sizes
    pe = torch.randn(1, 21, 256, generator=gen).to(torch.bfloat16); ppe = torch.randn(1, 128, generator=gen).to(torch.bfloat16)
    npe = torch.randn(1, 21, 256, generator=gen).to(torch.bfloat16); nppe = torch.randn(1, 128, generator=gen).to(torch.bfloat16)
    lat0 = torch.randn(Gn, 16, hw // 8, hw // 8, generator=gen)
    noises = [torch.randn(Gn, 16, hw // 8, hw // 8, generator=gen) for _ in range(steps)]
    ids = synthetic.clip_input_ids(Gn, 4)
    kw = dict(num_inference_steps=steps, guidance_scale=4.5, height=hw, width=hw, noise_level=0.8,
              mini_num_image_per_prompt=Gn, train_num_steps=T, process_index=0, sample_num_steps=steps, random_timestep=0)
    image, lats, lps, _ = pipeline_with_logprob_random(
        pipe, prompt_embeds=pe.cuda(), pooled_prompt_embeds=ppe.cuda(), negative_prompt_embeds=npe.cuda(),
        negative_pooled_prompt_embeds=nppe.cuda(), latents=lat0.cuda(), noises=[n.cuda() for n in noises],
        output_type="pt", **kw)
    scores = vit.pickscore_scores(clip.get_image_features(images=image.to(torch.bfloat16)), clip.get_text_features(ids),
                                  Wc["logit_scale"].float())
    # oracle: fp32 math on the same bf16-rounded weights, on the GPU box's... GPU via torch (plain fp32 reference)
    W32 = {k: v.float().cuda() for k, v in Wm.items()}
    V32 = {k: v.float().cuda() for k, v in Wv.items()}
    C32 = {k: v.float().cuda() for k, v in Wc.items()}
    sch = FlowMatchEulerScheduler()
    sch.device = "cuda"
    tr = lambda x, t, c, p: o_m.mmdit_forward(W32, mcfg, x.float(), t, c.float(), p.float()).to(x.dtype)
    o_img, o_lats, o_lps, _ = o_r.rollout(
        tr, lambda z: o_v.vae_decode(V32, vcfg, z), sch, prompt_embeds=pe.cuda(), pooled_prompt_embeds=ppe.cuda(),
        negative_prompt_embeds=npe.cuda(), negative_pooled_prompt_embeds=nppe.cuda(), latents=lat0.cuda(),
        noises=[n.cuda() for n in noises], **kw)
    # per-step log-prob with injected noise is -std^2 * mean(eps^2) up to the bf16 cast of the sample: tight
    for a, b in zip(lps, o_lps):
        assert torch.allclose(a, b, rtol=2e-3), (a, b)
    # latents after 1 and 2 stochastic steps: bf16 transformer vs fp32 oracle
    for a, b in zip(lats[1:], o_lats[1:]):
        assert ((a.float() - b.float()).norm() / b.float().norm()).item() < 3e-2
    err = (image - o_img).abs()
    assert err.mean().item() < 1.5e-2, err.mean().item()
    px = _pil_clip_preprocess(o_rw.to_uint8(o_img.to(torch.bfloat16).cpu()).permute(0, 2, 3, 1).numpy()).to(torch.bfloat16).float().cuda()
    o_scores = o_rw.pickscore_from_embeddings(o_t.clip_image_features(C32, ccfg, px), o_t.clip_text_features(C32, ccfg, ids.cuda()),
                                              C32["logit_scale"])
    print(f"C1 end-to-end PickScore: product {scores.tolist()} oracle {o_scores.tolist()} max |diff| {(scores - o_scores).abs().max().item():.3e}")
    assert (scores - o_scores).abs().max().item() < E2E_SCORE_TOL * max(1.0, o_scores.abs().max().item()), (scores, o_scores)


def test_cfg_halves_on_two_streams_give_the_same_rollout():
    """pipeline.cfg_two_streams: the unconditional and conditional halves of the CFG batch as two forwards on two HIP streams instead of one
    forward of twice the batch (bench.py's `cfg_two_streams` leg).  Every row of every kernel of the transformer is independent of the other
    rows, so the latents, log-probs and the decoded image of a rollout are bit-identical -- at a reduced depth and at config 2's full size
    (where the halves take differently tiled launches: 8192 + 1640 rows instead of 16384 + 3280)."""
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
    from adv_grpo_amd.mmdit import SD3Transformer2DModel
    from adv_grpo_amd.model_configs import MMDiTConfig, VaeConfig
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    with synthetic.on_device("cuda"):
        vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 22, fp16_checkpoint=True), VaeConfig(), "cuda")
    for mcfg, G, hw, steps in ((MMDiTConfig(num_layers=3, num_heads=6, pos_embed_max_size=96, dual_attention_layers=(0, 1)), 3, 256, 4), (MMDiTConfig(), 8, 512, 3)):
        with synthetic.on_device("cuda"):
            pipe = SD3Pipeline(SD3Transformer2DModel(synthetic.mmdit_weights(mcfg, 21), mcfg, "cuda"), vae, "cuda")
        pe, ppe, npe, nppe = (t.cuda().to(torch.bfloat16) for t in synthetic.prompt_embeddings(5))
        kw = dict(prompt_embeds=pe, pooled_prompt_embeds=ppe, negative_prompt_embeds=npe, negative_pooled_prompt_embeds=nppe, num_inference_steps=steps,
                  guidance_scale=4.5, height=hw, width=hw, noise_level=0.8, mini_num_image_per_prompt=G, train_num_steps=2, process_index=0,
                  sample_num_steps=steps, random_timestep=0, seed=77, output_type="pt")
        img_a, lats_a, lps_a, _ = pipeline_with_logprob_random(pipe, **kw)
        pipe.cfg_two_streams = True
        img_b, lats_b, lps_b, _ = pipeline_with_logprob_random(pipe, **kw)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(lats_a, lats_b)) and all(torch.equal(a, b) for a, b in zip(lps_a, lps_b))
        assert torch.equal(img_a, img_b) and torch.isfinite(img_a).all()
        del pipe
