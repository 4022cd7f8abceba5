"""GPU: the epoch loop end to end on a reduced-depth stack -- D-step epoch then G-step epoch (DINO variant), and a
G-step epoch with the PickScore reward; checks the plumbing (shapes, gate, optimizer step, parameter movement)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(reward, **over):
    from adv_grpo_amd import synthetic, vit
    from adv_grpo_amd.config.experiments import get_config
    from adv_grpo_amd.d_step import DinoHeadTrainable
    from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
    from adv_grpo_amd.model_configs import ClipConfig, DinoConfig, MMDiTConfig, VaeConfig
    from adv_grpo_amd.pickscore_scorer import PickScoreScorer
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.trainer import SyntheticData, Trainer
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    cfg = get_config("dino_cotrain_sd3_patch_fast" if reward == "dino" else "pickscore_cotrain_sd3_fast", gpu_number=1)
    cfg.resolution = 256
    cfg.sample.num_steps = 4
    cfg.sample.num_image_per_prompt = cfg.sample.mini_num_image_per_prompt = 2
    cfg.sample.num_batches_per_epoch = 2
    cfg.train.gradient_accumulation_steps = 1
    cfg.update(over)
    mcfg = MMDiTConfig(num_layers=2, num_heads=4, joint_attention_dim=256, pooled_projection_dim=128,
                       pos_embed_max_size=96, dual_attention_layers=(0,))
    tr = SD3TransformerLoRA({k: v.to(torch.bfloat16) for k, v in synthetic.mmdit_weights(mcfg, 1).items()}, mcfg, "cuda")
    vae = AutoencoderKLDecoder({k: v.to(torch.bfloat16) for k, v in synthetic.vae_decoder_weights(VaeConfig(), 2).items()},
                               VaeConfig(), "cuda")
    head = None
    if reward == "dino":
        dc = DinoConfig(layers=2)
        scorer = vit.DinoV2({k: v.to(torch.bfloat16) for k, v in synthetic.dino_weights(dc, 3).items()}, dc, "cuda")
        head = DinoHeadTrainable(device="cuda", seed=0)
    else:
        cc = ClipConfig(v_layers=2, t_layers=2)
        scorer = PickScoreScorer("cuda", dtype=torch.bfloat16, model_sd=synthetic.clip_weights(cc, 4), clip_cfg=cc)
    data = SyntheticData(n_prompts=100, n_tokens=21, ctx_dim=256, pooled_dim=128, resolution=256)
    return Trainer(cfg, SD3Pipeline(tr, vae, "cuda"), data, scorer, head), tr, head


def test_dino_variant_d_then_g():
    tr_, model, head = _build("dino", d_times=2)
    h0, p0 = head.params.clone(), model.params.clone()
    a = tr_.run_epoch()                       # epoch 0: (0+1) % 2 != 0 -> D-step
    assert a["phase"] == "D" and torch.isfinite(torch.tensor(a["train/d_loss"]))
    assert not torch.equal(head.params, h0) and torch.equal(model.params, p0)
    b = tr_.run_epoch()                       # epoch 1: (1+1) % 2 == 0 -> G-step
    assert b["phase"] == "G" and tr_.global_step >= 2
    assert not torch.equal(model.params, p0) and torch.isfinite(model.params).all()
    assert (model.grads == 0).all()


def test_pickscore_variant_g_step():
    tr_, model, _ = _build("pickscore", train_d=False)
    p0 = model.params.clone()
    b = tr_.run_epoch()
    assert b["phase"] == "G" and not torch.equal(model.params, p0) and torch.isfinite(model.params).all()


def test_g_step_epoch_with_kl_term_logs_kl_loss():
    """config.train.beta > 0 through the epoch loop (TP:1105-1108,1126-1130,1158-1160): a second epoch, after the adapters
    have moved, logs a positive kl_loss and loss = policy_loss + beta * kl_loss."""
    tr_, model, _ = _build("pickscore", train_d=False)
    tr_.cfg.train.beta = 0.04
    recs = []
    log = tr_.logger.log
    tr_.logger.log = lambda rec, step=None: (recs.append(dict(rec)), log(rec, step))[1]
    tr_.run_epoch()
    tr_.run_epoch()
    steps = [r for r in recs if "kl_loss" in r]
    assert steps, recs
    first, last = steps[0], steps[-1]
    assert float(first["kl_loss"]) < 1e-12                   # B = 0 at initialisation: the policy IS the reference
    assert float(last["kl_loss"]) > 0.0 and torch.isfinite(torch.tensor(float(last["loss"])))
    assert abs(float(last["loss"]) - float(last["policy_loss"]) - 0.04 * float(last["kl_loss"])) < 1e-5


def test_pickscore_variant_d_step_gate():
    """mean(reference reward) < mean(generated reward) -> D-step on the CLIP scorer's last layer (TP:1025-1037)."""
    tr_, model, _ = _build("pickscore", train_d=True)
    w0 = tr_.scorer.model.v_enc.layers[-1]["fc1.w"].clone()
    p0 = model.params.clone()
    phases = [tr_.run_epoch()["phase"] for _ in range(2)]
    assert set(phases) <= {"D", "G"}
    if "D" in phases:
        assert not torch.equal(tr_.scorer.model.v_enc.layers[-1]["fc1.w"], w0)
    if "G" in phases:
        assert not torch.equal(model.params, p0)


def test_pickscore_variant_d_step_with_tune_layer_minus_2():
    """config.tune_layer = -2 through the Trainer (TP:1016-1020): the discriminator update reaches the last TWO vision layers; a
    non-negative or tuple value is refused with the reason."""
    from adv_grpo_amd.d_step_pickscore import ClipLayersTrainable
    tr_, model, _ = _build("pickscore", train_d=True, tune_layer=-2)
    assert isinstance(tr_.clip_trainable, ClipLayersTrainable) and tr_.clip_trainable.k == 2
    layers = tr_.scorer.model.v_enc.layers
    w0 = [L["fc1.w"].clone() for L in layers[-2:]]
    phases = [tr_.run_epoch()["phase"] for _ in range(3)]
    if "D" in phases:
        assert all(not torch.equal(L["fc1.w"], w) for L, w in zip(layers[-2:], w0))
    with pytest.raises(NotImplementedError):
        _build("pickscore", train_d=True, tune_layer=(11,))


def test_eval_loop_is_deterministic_swaps_ema_and_checkpoint_round_trips(tmp_path):
    """eval() (TP:269-382) + save_ckpt (TP:389-398) + lora_path reload (TP:506-509) on the reduced stack."""
    from adv_grpo_amd import checkpoint
    tr_, model, _ = _build("pickscore", train_d=False, save_dir=str(tmp_path))
    tr_.cfg.sample.eval_num_steps = 6
    tr_.cfg.sample.test_batch_size = 3
    tr_.run_epoch()                                   # moves the LoRA parameters and creates / updates the EMA copy
    model.ema_step(7)
    live = model.params.clone()
    a = tr_.evaluate(eval_reward_fn={"pickscore_cotrain": 1})
    b = tr_.evaluate(eval_reward_fn={"pickscore_cotrain": 1}, save_folder=str(tmp_path / "eval"))
    import json
    with open(tmp_path / "eval" / "prompt2img.json", encoding="utf-8") as f:          # scripts/eval.py:291-294
        p2i = json.load(f)
    assert len(p2i) == 3 and all((tmp_path / "eval" / v[0]).exists() and v[0].startswith("node0_rank0_00000_") for v in p2i.values())
    assert set(a) == set(b) == {"eval_reward_pickscore_cotrain", "eval_reward_avg"}
    # fixed seed-0 latents and noise 0: repeatable up to the order of the f64 atomic sums in the VAE's GroupNorm statistics
    assert all(v == v and abs(v - b[k]) <= 5e-3 * abs(v) for k, v in a.items())
    assert torch.equal(model.params, live)            # EMA swapped back (copy_temp_to)
    # checkpoint: PEFT layout, EMA weights written; reloading them reproduces the EMA model's prediction
    path = tr_.save_checkpoint()
    state, cfg = checkpoint.load_lora(path)
    assert cfg["r"] == 32 and cfg["lora_alpha"] == 64
    ema_state = {k: v for k, v in state.items()}
    key = "transformer_blocks.0.attn.to_q.lora_A.weight"
    from adv_grpo_amd.mmdit_train import RANK
    assert torch.equal(ema_state[key].cuda(), model.A_view(model.adapters[key[:-len(".lora_A.weight")]], model.ema)[:RANK])
    x = torch.randn(2, 16, 32, 32, device="cuda").to(torch.bfloat16)
    t = torch.tensor([500.0, 500.0], device="cuda")
    ctx = torch.randn(2, 21, 256, device="cuda").to(torch.bfloat16)
    pooled = torch.randn(2, 128, device="cuda").to(torch.bfloat16)
    model.load_lora_state(ema_state)
    y1 = model(x, t, ctx, pooled)[0].clone()
    model.load_lora_state({k: v.clone() for k, v in state.items()})
    assert torch.equal(model(x, t, ctx, pooled)[0], y1)


def test_eval_matches_oracle_eval_on_reduced_stack():
    """f1: Trainer.evaluate (TP:269-382) against the same loop through the fp32 oracle: `eval_num_steps` deterministic steps
    (noise_level = 0), latents of every test batch from a CPU generator seeded with 0 (TP:298-299), PickScore of the
    decoded images, mean over the prompts.  bf16 transformer / VAE / towers vs fp32: a few per cent on the score scale."""
    from adv_grpo_amd import synthetic
    from oracle import lora as o_lora
    from oracle import mmdit as o_m
    from oracle import rewards as o_rw
    from oracle import rollout as o_r
    from oracle import vae as o_v
    from oracle import vit as o_t
    from oracle.scheduler import FlowMatchEulerScheduler
    from tests.test_gpu_vit import _pil_clip_preprocess
    tr_, model, _ = _build("pickscore", train_d=False)
    c = tr_.cfg
    c.sample.eval_num_steps, c.sample.test_batch_size, c.train.ema = 6, 3, False
    got = tr_.evaluate(n_prompts=3, eval_reward_fn={"pickscore_cotrain": 1})
    # ---- the oracle's eval on identical weights / prompts
    mcfg = o_m.MMDiTConfig(num_layers=2, num_heads=4, joint_attention_dim=256, pooled_projection_dim=128,
                           pos_embed_max_size=96, dual_attention_layers=(0,))
    vcfg, ccfg = o_v.VaeConfig(), o_t.ClipConfig(v_layers=2, t_layers=2)
    W = {k: v.to(torch.bfloat16).float().cuda() for k, v in synthetic.mmdit_weights(mcfg, 1).items()}
    lora = {k: v.float().cuda() for k, v in model.lora_state_dict().items()}
    Weff = o_lora.effective_weights(W, lora)
    V32 = {k: v.to(torch.bfloat16).float().cuda() for k, v in synthetic.vae_decoder_weights(vcfg, 2).items()}
    C32 = {k: v.float().cuda() for k, v in synthetic.clip_weights(ccfg, 4).items()}
    sch = FlowMatchEulerScheduler(); sch.device = "cuda"
    data = tr_.data
    idxs = [0, 1, 2]
    pe = torch.cat([data.prompt(i)[0] for i in idxs]); ppe = torch.cat([data.prompt(i)[1] for i in idxs])
    npe, nppe = data.neg[0].repeat(3, 1, 1), data.neg[1].repeat(3, 1)
    lat0 = torch.randn(3, 16, c.resolution // 8, c.resolution // 8, generator=torch.Generator().manual_seed(0), dtype=pe.dtype)
    fn = lambda x, t, cc, p: o_m.mmdit_forward(Weff, mcfg, x.float(), t, cc.float(), p.float()).to(x.dtype)
    with torch.no_grad():
        img, _, _, _ = o_r.rollout(fn, lambda z: o_v.vae_decode(V32, vcfg, z), sch, prompt_embeds=pe, pooled_prompt_embeds=ppe,
                                   negative_prompt_embeds=npe, negative_pooled_prompt_embeds=nppe, num_inference_steps=6,
                                   guidance_scale=c.sample.guidance_scale, height=c.resolution, width=c.resolution, noise_level=0,
                                   mini_num_image_per_prompt=1, train_num_steps=c.sample.train_num_steps, process_index=0,
                                   sample_num_steps=c.sample.num_steps, random_timestep=c.sample.random_timestep,
                                   latents=lat0.cuda(), noises=[torch.zeros_like(lat0).cuda() for _ in range(6)])
        px = _pil_clip_preprocess(o_rw.to_uint8(img.to(torch.bfloat16).cpu()).permute(0, 2, 3, 1).numpy()).to(torch.bfloat16).float().cuda()
        ids = torch.cat([data.clip_ids(i, 1) for i in idxs])
        ref = o_rw.pickscore_from_embeddings(o_t.clip_image_features(C32, ccfg, px), o_t.clip_text_features(C32, ccfg, ids),
                                             C32["logit_scale"]).mean().item()
    print(f"eval reward: product {got['eval_reward_pickscore_cotrain']:.5f} oracle {ref:.5f} |diff| {abs(got['eval_reward_pickscore_cotrain'] - ref):.3e}")
    assert abs(got["eval_reward_pickscore_cotrain"] - ref) < 1.5e-3 * max(1.0, abs(ref)), (got, ref)       # measured 6.3e-4
    assert got["eval_reward_avg"] == got["eval_reward_pickscore_cotrain"]


def test_reward_futures_from_worker_threads_match_serial_scoring():
    """a11 / include/advgrpo.h "callable from worker threads, one stream each": two scorer calls issued from two worker
    threads on their own streams WHILE the main thread runs a rollout must return exactly what serial calls return."""
    import threading
    from adv_grpo_amd import rewards
    from adv_grpo_amd.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random
    tr_, model, _ = _build("pickscore", train_d=False)
    g = torch.Generator().manual_seed(3)
    imgs = [torch.rand(2, 3, 256, 256, generator=g).cuda().to(torch.bfloat16) for _ in range(2)]
    ids = [tr_.data.clip_ids(i, 2) for i in range(2)]
    fn = rewards.multi_score("cuda", {"pickscore_cotrain": 1})
    serial = [fn(imgs[k], ids[k], [{}] * 2, scorer=tr_.scorer)[0]["avg"].clone() for k in range(2)]
    torch.cuda.synchronize()
    out, errs = [None, None], []
    lock = threading.Lock()          # the scorer's towers keep per-model workspaces: calls on ONE model are serialised

    def work(k):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(3):
                    with lock:
                        r = fn(imgs[k], ids[k], [{}] * 2, scorer=tr_.scorer)[0]["avg"]
                        st.synchronize()
                out[k] = r.clone()
            st.synchronize()
        except Exception as e:       # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    pe, ppe = tr_.data.prompt(5)
    c = tr_.cfg
    img, _, lps, _ = pipeline_with_logprob_random(                 # concurrently, on the main thread's stream
        tr_.pipe, prompt_embeds=pe, pooled_prompt_embeds=ppe, negative_prompt_embeds=tr_.data.neg[0],
        negative_pooled_prompt_embeds=tr_.data.neg[1], num_inference_steps=4, guidance_scale=4.5, output_type="pt", height=256,
        width=256, noise_level=0.8, mini_num_image_per_prompt=2, train_num_steps=2, process_index=0, sample_num_steps=4,
        random_timestep=0, seed=11)
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for k in range(2):
        assert torch.equal(out[k], serial[k]), (out[k], serial[k])
    img2, _, lps2, _ = pipeline_with_logprob_random(               # and the rollout was not disturbed either
        tr_.pipe, prompt_embeds=pe, pooled_prompt_embeds=ppe, negative_prompt_embeds=tr_.data.neg[0],
        negative_pooled_prompt_embeds=tr_.data.neg[1], num_inference_steps=4, guidance_scale=4.5, output_type="pt", height=256,
        width=256, noise_level=0.8, mini_num_image_per_prompt=2, train_num_steps=2, process_index=0, sample_num_steps=4,
        random_timestep=0, seed=11)
    assert torch.equal(torch.stack(lps), torch.stack(lps2))
    # the trainer's own futures: async and inline scoring give the same rewards for the same epoch
    a = tr_.sample_epoch()
    tr_.async_reward = False
    b = tr_.sample_epoch()
    assert torch.equal(a["log_probs"], b["log_probs"])
    # (the decoded images repeat only up to the order of the f64 atomic sums in the VAE's GroupNorm statistics)
    assert torch.allclose(a["rewards"], b["rewards"], rtol=1e-2, atol=2e-3), (a["rewards"], b["rewards"])


def test_no_discriminator_multi_reward_config_runs():
    """BASELINE config 4's preset (`pickscore_sd3_fast`: reward_fn = {pickscore: 0.5, ocr: 0.5}, no discriminator keys,
    config/grpo.py:379-427): the loop keeps the D/G gate at G, scores no reference images, and the ocr scorer runs as a
    host plugin on the prompt strings with a stand-in recogniser (PaddleOCR is not installed)."""
    from adv_grpo_amd import rewards, synthetic
    from adv_grpo_amd.config.experiments import get_config
    from adv_grpo_amd.model_configs import ClipConfig
    tr0, model, _ = _build("pickscore", train_d=False)
    cfg = get_config("pickscore_sd3_fast", gpu_number=1)
    assert cfg.get("train_d") is None and dict(cfg.reward_fn.items()) == {"pickscore": 0.5, "ocr": 0.5}
    cfg.resolution = 256
    cfg.sample.num_steps = 4
    cfg.sample.num_image_per_prompt = cfg.sample.mini_num_image_per_prompt = 2
    cfg.sample.num_batches_per_epoch = 2
    cfg.train.gradient_accumulation_steps = 1
    cc = ClipConfig(v_layers=2, t_layers=2)
    rewards.configure_pickscore(synthetic.clip_weights(cc, 4), cc)
    seen = []

    def recognizer(img):
        assert img.dtype.name == "uint8" and img.shape == (256, 256, 3)
        seen.append(1)
        return "prompt 1" if len(seen) % 2 else ""
    rewards.configure_ocr(recognizer)
    from adv_grpo_amd.trainer import Trainer
    tr_ = Trainer(cfg, tr0.pipe, tr0.data, None, None)
    assert not tr_.needs_reference
    p0 = model.params.clone()
    s = tr_.sample_epoch()
    assert "reference_rewards" not in s and s["rewards"].shape == (4,) and len(seen) == 4
    info = tr_.run_epoch()
    assert info["phase"] == "G" and not torch.equal(model.params, p0)
    assert "zero_std_ratio" in tr_.last_metrics and "reward_std_mean" in tr_.last_metrics


def test_image_similarity_scorer_on_the_kernels():
    """rewards.py:147-203 (eval reward): max cosine similarity of DINOv2 CLS embeddings, kernels vs the fp32 oracle tower."""
    from adv_grpo_amd import rewards, synthetic, vit
    from adv_grpo_amd.model_configs import DinoConfig
    from oracle import rewards as o_rw
    from oracle import vit as o
    dc = DinoConfig(layers=2)
    W = {k: v.to(torch.bfloat16) for k, v in synthetic.dino_weights(dc, 3).items()}
    rewards.configure_dino(vit.DinoV2(W, dc, "cuda"))
    g = torch.Generator().manual_seed(1)
    a, b = torch.rand(3, 3, 128, 128, generator=g).to(torch.bfloat16), torch.rand(4, 3, 128, 128, generator=g).to(torch.bfloat16)
    s, info = rewards.multi_score("cuda", {"image_similarity": 1.0})(a.cuda(), None, None, ref_images=b.cuda())[0]["image_similarity"], None
    W32 = {k: v.float().cuda() for k, v in W.items()}
    ea = o.dino_forward_features(W32, dc, o_rw.dino_preprocess(a, cuda_semantics=True).float().cuda())[:, 0]
    eb = o.dino_forward_features(W32, dc, o_rw.dino_preprocess(b, cuda_semantics=True).float().cuda())[:, 0]
    ea, eb = ea / ea.norm(dim=-1, keepdim=True), eb / eb.norm(dim=-1, keepdim=True)
    ref = (ea @ eb.T).max(dim=1).values
    print(f"image_similarity: max |diff| {(s - ref).abs().max().item():.3e}")
    assert s.shape == (3,) and (s - ref).abs().max().item() < 8e-3, (s, ref)       # measured 3.4e-3 (bf16 tower vs fp32)


def test_checkpoint_resume_restores_state_bit_exactly(tmp_path):
    """f2: the resume file next to the PEFT adapter restores live LoRA weights, Adam moments, step counts, EMA and the
    discriminator head (none of which upstream's save_ckpt writes, TP:389-398 / TD:592-603): a restored trainer takes the
    SAME optimizer step from the same samples, bit for bit."""
    from adv_grpo_amd import checkpoint
    tr_a, model_a, head_a = _build("dino", d_times=2, save_dir=str(tmp_path))
    tr_a.run_epoch()                       # D-step: moves the head and its Adam moments
    tr_a.run_epoch()                       # G-step: moves the LoRA weights, moments, EMA bookkeeping
    model_a.ema_step(7)
    path = tr_a.save_checkpoint()
    assert sorted(os.listdir(path)) == ["adapter_config.json", "adapter_model.safetensors", "trainer_state.json", "trainer_state.safetensors"]
    tr_b, model_b, head_b = _build("dino", d_times=2, save_dir=str(tmp_path))
    assert not torch.equal(model_b.params, model_a.params)
    assert tr_b.load_checkpoint(path) is True
    for x, y in ((model_a, model_b), (head_a, head_b)):
        for f in ("params", "exp_avg", "exp_avg_sq"):
            assert torch.equal(getattr(x, f), getattr(y, f)), f
        assert x.opt_step == y.opt_step
    assert torch.equal(model_a.ema, model_b.ema) and (tr_a.global_step, tr_a.epoch) == (tr_b.global_step, tr_b.epoch)
    # the adapter file alone (what upstream writes) still loads, with from_pretrained semantics
    state, _ = checkpoint.load_lora(path)
    assert torch.equal(state["transformer_blocks.0.attn.to_q.lora_A.weight"].cuda(),
                       model_a.A_view(model_a.adapters["transformer_blocks.0.attn.to_q"], model_a.ema)[:32])
    # and the restored trainer trains on: one more G-step from the same samples takes both models to the SAME bits (every
    # reduction on the path sums in a fixed order; the gradient-norm reduction behind clip_grad_norm_ used one atomicAdd per
    # block until round 3, and Adam turned its last-bit wobble into +-lr on near-zero gradients)
    samples = tr_a.sample_epoch()
    samples["advantages"] = torch.randn(samples["rewards"].shape[0], tr_a.cfg.sample.train_num_steps, device="cuda")
    before = model_b.params.clone()
    tr_a.g_step(samples)
    tr_b.g_step(samples)
    n_opt = tr_a.cfg.sample.num_batches_per_epoch
    for m in (model_a, model_b):
        assert torch.isfinite(m.params).all() and not torch.equal(m.params, before)
        assert (m.params - before).abs().max().item() <= 4 * n_opt * tr_a.cfg.train.learning_rate   # |Adam step| <= (1-b1)/sqrt(1-b2) lr
    assert tr_a.global_step == tr_b.global_step and model_a.opt_step == model_b.opt_step
    for f in ("params", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(model_a, f), getattr(model_b, f)), f


def test_groups_in_flight_do_not_change_the_samples():
    """sample_epoch with two prompt groups in flight (two HIP streams, two host threads) against one at a time: seeds depend on
    (config seed, batch index, rank) only and every kernel on the path sums in a fixed order (GroupNorm's statistics used to
    be accumulated with atomics: two decodes of the same latents then differed by up to 2e-2), so latents, log-probs, images
    and rewards are bit-identical."""
    tr_, _, _ = _build("pickscore", train_d=False)
    tr_.cfg.sample.num_batches_per_epoch = 3
    tr_.cfg.sample.groups_in_flight = 1
    a = tr_.sample_epoch()
    tr_.cfg.sample.groups_in_flight = 2
    b = tr_.sample_epoch()
    torch.cuda.synchronize()
    assert set(a) == set(b) and tr_._rollout_pool._max_workers == 2
    for k in a:
        if isinstance(a[k], torch.Tensor):
            assert torch.equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k
    assert a["latents"].shape[0] == 3 * tr_.cfg.sample.mini_num_image_per_prompt


def test_trainer_side_streams_are_its_rollout_streams():
    """Round 6: the streams a Trainer needs beside the launch stream exist from construction and are as few as possible -- the scoring stream and
    one stream per group in flight; the decoder's side stream for decodes on the launch stream and the G-step's adapter-gradient side stream ARE
    rollout streams (idle whenever those run).  A fifth live stream cost the in-flight schedule 2.5 %, a side stream chosen before the others
    came to life 27 % of a micro-step (LABNOTES 9)."""
    tr_, model, _ = _build("pickscore", train_d=False)
    main = torch.cuda.current_stream()
    roll = tr_._rollout_stream_list
    assert len(roll) == 2 and roll[0].cuda_stream != roll[1].cuda_stream and all(r.cuda_stream != main.cuda_stream for r in roll)
    assert [s.cuda_stream for s in tr_.pipe.vae._side[main.cuda_stream]] == [roll[0].cuda_stream]
    assert model._wgrad_stream.cuda_stream == roll[-1].cuda_stream
    a = tr_.run_epoch()                                            # the schedule still runs (two groups in flight, one decode stream each) ...
    assert a["phase"] == "G" and torch.isfinite(model.params).all()
    assert tr_.pipe.vae.two_streams is True                        # ... and hands the decoder's split back afterwards
    tr1, model1, _ = _build("pickscore", train_d=False)
    tr1.cfg.sample.groups_in_flight = 1
    b = tr1.run_epoch()
    assert b["phase"] == "G"


def test_bench_self_spawn_path_at_world_1():
    """`bench.py --spawn` takes the launcher path a plain `python bench.py --gpus N` takes for N > 1 (re-exec under
    torch.distributed.run, RCCL process group, all-reduce of ones) on this one-GPU box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--spawn", "--steps", "1", "--warmup", "1",
                        "--no-epoch", "--no-cpu-baseline", "--no-pricing"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["value"] > 0
    # the fields that make a SCALE run diagnosable from its one line (per-rank times, the two collectives, the solo leg)
    sd = line["scaling_diagnostics"]
    assert len(sd["ms_per_step_by_rank"]["all"]) == 1 and sd["ms_per_step_by_rank"]["max"] > 0
    assert sd["reward_all_gather_ms"] >= 0 and sd["lora_gradient_all_reduce_ms"] >= 0 and sd["lora_gradient_bytes"] == 18_776_064 * 4
    assert sd["value_at_n1_same_run"] > 0 and 0.8 < sd["value_over_n_times_n1"] < 1.25


@pytest.mark.parametrize("model,preset,extra", [
    ("Qwen/Qwen-Image", "dino_cotrain_sd3_patch_fast", ["--linear-dtype", "fp8", "--images-per-prompt", "8", "--no-train-d"]),
    ("stabilityai/stable-diffusion-3.5-large", "pickscore_sd3_fast", ["--images-per-prompt", "4"])])
def test_launcher_selects_the_model_by_pretrained_name(tmp_path, model, preset, extra):
    """scripts/train_sd3_fast.py --model overrides config.pretrained.model, the field by which the reference selects BASELINE configs 4
    and 5 (config/grpo.py:324,330): a reduced-depth run of each goes through sampling, scoring and one G epoch."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    log = tmp_path / "train.jsonl"
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "train_sd3_fast.py"), "--config",
                        os.path.join(root, "config", "grpo.py") + ":" + preset, "--model", model, "--resolution", "256", "--layers", "2",
                        "--batches", "1", "--epochs", "1", "--log", str(log)] + extra, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["phase"] == "G" and line["global_step"] == 1


def test_bench_world_2_end_to_end_on_one_gpu():
    """The N > 1 branches of bench.py and of the Trainer, executed: two ranks under torch.distributed.run, the rank partition of the
    sampler, the packed reward all-gather, the scaling diagnostics (per-rank times, both collectives timed, the solo leg) and the epoch
    leg at world 2 (state broadcast at construction, the 75 MB LoRA-gradient all-reduce before each optimizer step, phases as the maximum
    over ranks).  RCCL refuses two ranks on one device, so ADVGRPO_BENCH_SHARED_GPU=1 puts both ranks on cuda:0 over gloo with the device
    tensors staged through the host inside the collectives: an integration test of the code paths, not a measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ADVGRPO_BENCH_SHARED_GPU="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-pricing"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    sd = line["scaling_diagnostics"]
    assert len(sd["ms_per_step_by_rank"]["all"]) == 2 and len(sd["solo_ms_per_step_by_rank"]["all"]) == 2
    assert sd["reward_all_gather_ms"] > 0 and sd["lora_gradient_all_reduce_ms"] > 0 and sd["lora_gradient_bytes"] == 18_776_064 * 4
    ep = line["epoch"]
    assert ep["images"] == 32 and ep["phases_are"] == "max over ranks" and ep["g_step_inside"]["grad_all_reduce"]["n"] == 2
    assert ep["g_step_inside"]["grad_all_reduce"]["mean_ms"] > 0.05         # a real exchange happened (0.01 ms at world 1)


def test_full_size_epoch_config2(tmp_path):
    """One whole sample -> score -> gather -> advantage -> G-step epoch at BASELINE config 2's FULL size (SD3.5-medium, 24
    blocks, D = 1536, 512^2, 10 steps, CFG 4.5, G = 8, SDE window 2, fp32-equivalent VAE decode, full CLIP ViT-H PickScore)
    through trainer.Trainer -- the loop bench.py's `epoch` leg times.  Properties of TP:709-1191: the metrics the reference
    logs exist and are finite (TP:941-955,975-988,1132-1183), the gate takes the G branch with the discriminator off, every
    new log-prob equals the rollout's at update 0 up to the bf16 cast of the stored next latents (the rollout takes its log-prob
    from the f32 sample BEFORE the cast, PF:646-660, the replay from the stored bf16 one, TP:258: |d log p| ~ 5e-5, approx_kl
    ~ 1e-9 -- with clip_range = 1e-5 that already counts as clipped, in the reference too), the LoRA moved, stayed finite, and
    the gradient vector was zeroed.  (This test found the G-step differentiating exp(log_prob - ADVANTAGE): losses.grpo_loss
    passed pointers of temporaries that were freed and re-used inside its own argument list.)"""
    import json
    from adv_grpo_amd import synthetic
    from adv_grpo_amd.config.experiments import get_config
    from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
    from adv_grpo_amd.model_configs import ClipConfig, MMDiTConfig, VaeConfig
    from adv_grpo_amd.pickscore_scorer import PickScoreScorer
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.trainer import SyntheticData, Trainer
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    cfg = get_config("pickscore_cotrain_sd3_fast", gpu_number=1)
    cfg.sample.num_image_per_prompt = 8
    cfg.sample.num_batches_per_epoch = 1
    cfg.train.gradient_accumulation_steps = 1
    cfg.train_d = False
    mcfg = MMDiTConfig()
    with synthetic.on_device("cuda"):
        tr = SD3TransformerLoRA(synthetic.mmdit_weights(mcfg, 1234), mcfg, "cuda", seed=cfg.seed)
        vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 4321), VaeConfig(), "cuda")
        scorer = PickScoreScorer("cuda", dtype=torch.bfloat16, model_sd=synthetic.clip_weights(ClipConfig(), 777), clip_cfg=ClipConfig())
    log = tmp_path / "metrics.jsonl"
    trainer = Trainer(cfg, SD3Pipeline(tr, vae, "cuda"), SyntheticData(resolution=cfg.resolution, device="cuda"), scorer, None, 0, 1,
                      log_path=str(log))
    p0 = tr.params.clone()
    out = trainer.run_epoch()
    torch.cuda.synchronize()
    assert out["phase"] == "G" and trainer.global_step == 1 and trainer.epoch == 1
    assert torch.isfinite(tr.params).all() and not torch.equal(tr.params, p0) and (tr.grads == 0).all()
    recs = [json.loads(l) for l in open(log)]
    epoch_rec = next(r for r in recs if "reward_avg" in r)
    for k in ("reward_avg", "zero_std_ratio", "reward_std_mean", "group_size", "trained_prompt_num"):
        assert k in epoch_rec and torch.isfinite(torch.tensor(float(epoch_rec[k]))), (k, epoch_rec)
    assert epoch_rec["group_size"] == 8 and epoch_rec["trained_prompt_num"] == 1
    step_rec = next(r for r in recs if "approx_kl" in r)
    for k in ("loss", "policy_loss", "approx_kl", "clipfrac", "clipfrac_gt_one", "clipfrac_lt_one"):
        assert k in step_rec and torch.isfinite(torch.tensor(float(step_rec[k]))), (k, step_rec)
    assert 0.0 <= step_rec["approx_kl"] < 1e-8, step_rec                                 # ratio == 1 at update 0 up to the bf16 cast


def test_full_size_epoch_config4(tmp_path):
    """BASELINE config 4 at FULL size through the epoch loop: SD3.5-large (38 joint blocks, D = 2432, 8 B parameters) with LoRA,
    1024 x 1024, 10 steps, CFG 4.5, G = 4, the multi-reward preset `pickscore_sd3_fast` (config/grpo.py:379-427: PickScore +
    OCR at 0.5 each, no discriminator, random SDE window), fp32-equivalent VAE decode at 1024^2, full CLIP ViT-H PickScore; the
    OCR half is the host plugin with a stand-in recogniser (PaddleOCR is not installed).  Properties: gate stays at G, rewards =
    0.5 pick + 0.5 ocr per image, metrics finite, the replayed log-probs equal the rollout's up to the bf16 cast of the stored
    latents (approx_kl < 1e-8), the LoRA moves and stays finite."""
    import json
    from adv_grpo_amd import rewards, synthetic
    from adv_grpo_amd.config.experiments import get_config
    from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
    from adv_grpo_amd.model_configs import ClipConfig, MMDiTConfig, VaeConfig
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.trainer import SyntheticData, Trainer
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    cfg = get_config("pickscore_sd3_fast", gpu_number=1)
    cfg.resolution = 1024
    cfg.sample.num_image_per_prompt = cfg.sample.mini_num_image_per_prompt = 4
    cfg.sample.num_batches_per_epoch = 1
    cfg.train.gradient_accumulation_steps = 1
    mcfg = MMDiTConfig(num_layers=38, num_heads=38, dual_attention_layers=(), pos_embed_max_size=192)
    with synthetic.on_device("cuda"):
        tr = SD3TransformerLoRA(synthetic.mmdit_weights(mcfg, 1234), mcfg, "cuda", seed=cfg.seed)
        vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 4321), VaeConfig(), "cuda")
        rewards.configure_pickscore(synthetic.clip_weights(ClipConfig(), 777), ClipConfig())
    calls = []

    def recognizer(img):
        assert img.dtype.name == "uint8" and img.shape == (1024, 1024, 3)
        calls.append(1)
        return "prompt" if len(calls) % 2 else ""
    rewards.configure_ocr(recognizer)
    log = tmp_path / "metrics.jsonl"
    trainer = Trainer(cfg, SD3Pipeline(tr, vae, "cuda"), SyntheticData(resolution=1024, device="cuda"), None, None, 0, 1,
                      log_path=str(log))
    assert not trainer.needs_reference
    p0 = tr.params.clone()
    out = trainer.run_epoch()
    torch.cuda.synchronize()
    print(f"config 4 epoch: phases {trainer.timers}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    assert out["phase"] == "G" and len(calls) == 4
    assert torch.isfinite(tr.params).all() and not torch.equal(tr.params, p0) and (tr.grads == 0).all()
    recs = [json.loads(l) for l in open(log)]
    epoch_rec = next(r for r in recs if "reward_avg" in r)
    assert epoch_rec["group_size"] == 4 and all(torch.isfinite(torch.tensor(float(epoch_rec[k]))) for k in
                                                ("reward_avg", "zero_std_ratio", "reward_std_mean"))
    step_rec = next(r for r in recs if "approx_kl" in r)
    assert 0.0 <= step_rec["approx_kl"] < 1e-8 and torch.isfinite(torch.tensor(float(step_rec["loss"]))), step_rec


def test_full_size_epoch_config3(tmp_path):
    """BASELINE config 3 at FULL size on one rank: the adversarial loop the reference is named after (TD:156-232,1091-1115) --
    `dino_cotrain_sd3_patch_fast`, SD3.5-medium (24 blocks) 512^2, 10 steps, G = 8, reward = the co-trained DINOv2 ViT-B/14 @ 518
    patch discriminator + head (RW:375-434) -- one D epoch ((0 + 1) % d_times != 0: train_dino on 16 reference + 16 generated
    images, hinge on CLS + 0.3 x hinge on 64 patches, Adam on the head) then one G epoch ((1 + 1) % 2 == 0).  Properties: the
    property set of test_full_size_epoch_config2 for the G epoch; for the D epoch train/d_loss and train/acc are logged and finite,
    the head moved and the LoRA did not; generated AND reference images are scored every epoch (reference_reward_avg)."""
    import json
    from adv_grpo_amd import synthetic, vit
    from adv_grpo_amd.config.experiments import get_config
    from adv_grpo_amd.d_step import DinoHeadTrainable
    from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
    from adv_grpo_amd.model_configs import DinoConfig, MMDiTConfig, VaeConfig
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.trainer import SyntheticData, Trainer
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    cfg = get_config("dino_cotrain_sd3_patch_fast", gpu_number=1)
    cfg.sample.num_image_per_prompt = 8
    cfg.sample.num_batches_per_epoch = 2
    cfg.train.gradient_accumulation_steps = 1
    cfg.d_times = 2
    mcfg, dcfg = MMDiTConfig(), DinoConfig()
    with synthetic.on_device("cuda"):
        tr = SD3TransformerLoRA(synthetic.mmdit_weights(mcfg, 1234), mcfg, "cuda", seed=cfg.seed)
        vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 4321), VaeConfig(), "cuda")
        scorer = vit.DinoV2(synthetic.dino_weights(dcfg, 888), dcfg, "cuda")
    head = DinoHeadTrainable(device="cuda", seed=0)
    log = tmp_path / "metrics.jsonl"
    trainer = Trainer(cfg, SD3Pipeline(tr, vae, "cuda"), SyntheticData(resolution=cfg.resolution, device="cuda"), scorer, head, 0, 1,
                      log_path=str(log))
    assert trainer.needs_reference and trainer.variant == "dino"
    p0, h0 = tr.params.clone(), head.params.clone()
    a = trainer.run_epoch()
    torch.cuda.synchronize()
    assert a["phase"] == "D" and torch.isfinite(torch.tensor(float(a["train/d_loss"]))) and 0.0 <= float(a["train/acc"]) <= 1.0
    assert not torch.equal(head.params, h0) and torch.isfinite(head.params).all() and torch.equal(tr.params, p0)
    h1 = head.params.clone()
    b = trainer.run_epoch()
    torch.cuda.synchronize()
    print(f"config 3 epochs: phases {trainer.timers}, d_loss {float(a['train/d_loss']):.4f}, acc {float(a['train/acc']):.3f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    assert b["phase"] == "G" and trainer.epoch == 2
    assert torch.isfinite(tr.params).all() and not torch.equal(tr.params, p0) and (tr.grads == 0).all() and torch.equal(head.params, h1)
    recs = [json.loads(l) for l in open(log)]
    ep = [r for r in recs if "reward_avg" in r]
    assert len(ep) == 2
    for r in ep:
        for k in ("reward_avg", "reference_reward_avg", "zero_std_ratio", "reward_std_mean", "group_size", "trained_prompt_num"):
            assert k in r and torch.isfinite(torch.tensor(float(r[k]))), (k, r)
        assert r["group_size"] == 8
    assert [r["trained_prompt_num"] for r in ep] == [2, 4]            # distinct prompts seen so far (stat_tracking.py:73-76)
    d_rec = next(r for r in recs if "train/d_loss" in r)
    assert torch.isfinite(torch.tensor(float(d_rec["train/d_loss"]))) and "train/acc" in d_rec
    steps = [r for r in recs if "approx_kl" in r]
    assert len(steps) == 2
    assert 0.0 <= steps[0]["approx_kl"] < 1e-8, steps[0]                  # update 0: ratio == 1 up to the bf16 cast of the latents
    for k in ("loss", "policy_loss", "approx_kl", "clipfrac", "clipfrac_gt_one", "clipfrac_lt_one"):
        assert all(torch.isfinite(torch.tensor(float(s[k]))) for s in steps), k


def test_full_size_epoch_config5(tmp_path):
    """BASELINE config 5 at FULL size on one rank, through the Trainer: Qwen-Image MMDiT (60 blocks, 24 x 128, LoRA r = 32 on the eight
    attention projections of every block), 1024^2, 10 steps, G = 8, fp8 Linears in the rollout AND in the replay, Qwen-Image's own VAE
    decoder, reward = the co-trained DINOv2-B/14 patch discriminator + head (`dino_cotrain_sd3_patch_fast` with the model set swapped:
    the reference names the config, config/grpo.py:324,330, and ships no Qwen-Image code, README.md:75).  One group per epoch; a D epoch
    then a G epoch (forward with one activation checkpoint per block, backward with per-block recomputation).  The property set of
    test_full_size_epoch_config3, and the whole thing inside one GPU's memory."""
    import json
    from adv_grpo_amd import synthetic, vit
    from adv_grpo_amd.config.experiments import get_config
    from adv_grpo_amd.d_step import DinoHeadTrainable
    from adv_grpo_amd.model_configs import DinoConfig, QwenMMDiTConfig, QwenVaeConfig
    from adv_grpo_amd.pipeline import SD3Pipeline
    from adv_grpo_amd.qwen_mmdit_train import QwenImageTransformerLoRA
    from adv_grpo_amd.qwen_vae import AutoencoderKLQwenImageDecoder
    from adv_grpo_amd.trainer import SyntheticData, Trainer
    cfg = get_config("dino_cotrain_sd3_patch_fast", gpu_number=1)
    cfg.sample.num_image_per_prompt = 8
    cfg.sample.num_batches_per_epoch = 1
    cfg.train.gradient_accumulation_steps = 1
    cfg.d_times = 2
    cfg.resolution = 1024
    cfg.linear_dtype = "fp8"
    qcfg, vcfg, dcfg = QwenMMDiTConfig(), QwenVaeConfig(), DinoConfig()
    torch.cuda.reset_peak_memory_stats()
    with synthetic.on_device("cuda"):
        tr = QwenImageTransformerLoRA(synthetic.qwen_mmdit_weights(qcfg, 4242, dtype=torch.bfloat16), qcfg, "cuda", seed=cfg.seed)
        vae = AutoencoderKLQwenImageDecoder(synthetic.qwen_vae_decoder_weights(vcfg, 2468, dtype=torch.bfloat16), vcfg, "cuda")
        scorer = vit.DinoV2(synthetic.dino_weights(dcfg, 888), dcfg, "cuda")
    head = DinoHeadTrainable(device="cuda", seed=0)
    log = tmp_path / "metrics.jsonl"
    data = SyntheticData(n_tokens=128, ctx_dim=3584, pooled_dim=8, resolution=1024, device="cuda")
    trainer = Trainer(cfg, SD3Pipeline(tr, vae, "cuda"), data, scorer, head, 0, 1, log_path=str(log))
    assert trainer.needs_reference and trainer.variant == "dino" and tr.fp8 is not None
    p0, h0 = tr.params.clone(), head.params.clone()
    a = trainer.run_epoch()
    torch.cuda.synchronize()
    assert a["phase"] == "D" and torch.isfinite(torch.tensor(float(a["train/d_loss"]))) and 0.0 <= float(a["train/acc"]) <= 1.0
    assert not torch.equal(head.params, h0) and torch.isfinite(head.params).all() and torch.equal(tr.params, p0)
    h1 = head.params.clone()
    b = trainer.run_epoch()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"config 5 epochs: phases {trainer.timers}, d_loss {float(a['train/d_loss']):.4f}, acc {float(a['train/acc']):.3f}, peak memory {peak:.1f} GiB")
    assert b["phase"] == "G" and trainer.epoch == 2 and peak < 200
    assert torch.isfinite(tr.params).all() and not torch.equal(tr.params, p0) and (tr.grads == 0).all() and torch.equal(head.params, h1)
    recs = [json.loads(l) for l in open(log)]
    ep = [r for r in recs if "reward_avg" in r]
    assert len(ep) == 2
    for r in ep:
        for k in ("reward_avg", "reference_reward_avg", "zero_std_ratio", "reward_std_mean", "group_size", "trained_prompt_num"):
            assert k in r and torch.isfinite(torch.tensor(float(r[k]))), (k, r)
        assert r["group_size"] == 8
    steps = [r for r in recs if "approx_kl" in r]
    assert len(steps) == 1
    assert 0.0 <= steps[0]["approx_kl"] < 1e-7, steps[0]          # update 0: the fp8 replay repeats the fp8 rollout (up to the bf16 cast of the stored latents)
    for k in ("loss", "policy_loss", "approx_kl", "clipfrac", "clipfrac_gt_one", "clipfrac_lt_one"):
        assert torch.isfinite(torch.tensor(float(steps[0][k]))), k
