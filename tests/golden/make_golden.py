#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE'S OWN CODE.

Runs only in the build container (needs /root/reference; the GPU box never has it, and no
test imports this file).  Nothing from the reference is copied: its modules are imported
(or single functions are exec'd from its source text at run time) and only numeric
inputs/outputs are stored.

Import shims (the image lacks diffusers / timm / torchvision / absl / ml_collections):
  * diffusers.utils.torch_utils.randn_tensor -> torch.randn(generator=...)   (its documented
    behaviour for a same-device generator)
  * diffusers...FlowMatchEulerDiscreteScheduler -> annotation-only placeholder
  * diffusers...pipeline_stable_diffusion_3.retrieve_timesteps -> oracle.scheduler.retrieve_timesteps
  * timm -> empty module (rewards.py imports it at top level; the co-train scorers receive
    the backbone as an argument)
  * torchvision.transforms -> PIL-based Compose/Resize/ToTensor/Normalize (only reached by
    train_dino's image preprocessing, whose output the stand-in backbone ignores)
The scheduler object handed to the reference functions is oracle.scheduler (diffusers is
absent => schedule itself stays "parity unpinned"; everything computed FROM sigmas is pinned).

Usage:  python tests/golden/make_golden.py
"""
import ast
import json
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle.scheduler import FlowMatchEulerScheduler, retrieve_timesteps  # noqa: E402
from oracle.standin import StandinVelocity, standin_vae_decode  # noqa: E402


# ----------------------------------------------------------------------------- shims
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shims():
    # transformers probes for timm/torchvision at import time: import it before the shims exist
    import transformers  # noqa: F401
    from transformers import CLIPModel, CLIPProcessor  # noqa: F401

    def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
        return torch.randn(shape, generator=generator, device=device, dtype=dtype)

    _mod("diffusers")
    _mod("diffusers.utils")
    _mod("diffusers.utils.torch_utils", randn_tensor=randn_tensor)
    _mod("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_flow_match_euler_discrete",
         FlowMatchEulerDiscreteScheduler=FlowMatchEulerScheduler)
    _mod("diffusers.pipelines")
    _mod("diffusers.pipelines.stable_diffusion_3")
    _mod("diffusers.pipelines.stable_diffusion_3.pipeline_stable_diffusion_3",
         retrieve_timesteps=lambda sch, n, device=None, sigmas=None, **kw: retrieve_timesteps(sch, n, device))
    _mod("timm")
    from PIL import Image

    class Compose:
        def __init__(self, ts): self.ts = ts
        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Resize:
        def __init__(self, size, interpolation=None): self.size = size
        def __call__(self, img): return img.resize(self.size[::-1], Image.BICUBIC)

    class ToTensor:
        def __call__(self, img):
            return torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)

    class Normalize:
        def __init__(self, mean, std):
            self.m = torch.tensor(mean)[:, None, None]; self.s = torch.tensor(std)[:, None, None]
        def __call__(self, x): return (x - self.m) / self.s

    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms", Compose=Compose, Resize=Resize, ToTensor=ToTensor,
                         Normalize=Normalize, InterpolationMode=types.SimpleNamespace(BICUBIC=3))


def extract_defs(path, names, ns):
    """exec selected top-level class/function definitions of a reference script."""
    src = open(path).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    return ns


def extract_block(path, start_marker, end_marker):
    lines = open(path).read().split("\n")
    s = next(i for i, l in enumerate(lines) if start_marker in l)
    e = next(i for i, l in enumerate(lines) if end_marker in l and i > s)
    return textwrap.dedent("\n".join(lines[s:e + 1]))


TP = os.path.join(REF, "scripts/train_sd3_fast_pickscore.py")
TD = os.path.join(REF, "scripts/train_sd3_fast_dino_patch.py")


# ----------------------------------------------------------------------------- goldens
def gold_sde():
    from adv_grpo.diffusers_patch.sd3_sde_with_logprob import sde_step_with_logprob_new as ref_step
    out = {}
    cases = [("s10_a", 10, 0, 0.8, (2, 16, 8, 8)), ("s10_b", 10, 1, 0.8, (1, 16, 32, 32)),
             ("s10_c", 10, 5, 0.0, (2, 16, 8, 8)), ("s4_a", 4, 0, 0.8, (2, 16, 16, 16)),
             ("s4_b", 4, 3, 0.8, (3, 16, 8, 8)), ("s10_last", 10, 9, 0.8, (2, 16, 8, 8))]
    for name, nsteps, idx, nl, shape in cases:
        sch = FlowMatchEulerScheduler(); sch.set_timesteps(nsteps)
        g = torch.Generator().manual_seed(sum(map(ord, name)) + 11)
        v = torch.randn(shape, generator=g)
        x = torch.randn(shape, generator=g)
        t = sch.timesteps[idx].unsqueeze(0)
        # sampling mode; reproduce the epsilon the reference draws from the global RNG
        torch.manual_seed(1234)
        eps = torch.randn(shape)
        torch.manual_seed(1234)
        nxt, lp, mean, std = ref_step(sch, v, t, x, noise_level=nl)
        # replay mode with per-sample timesteps (training call site TP:258-265)
        tb = sch.timesteps[idx].repeat(shape[0])
        nxt_bf = nxt.to(torch.bfloat16)
        _, lp_r, mean_r, std_r = ref_step(sch, v, tb, x, noise_level=nl, prev_sample=nxt_bf)
        assert torch.equal(mean_r, mean)
        for k, a in dict(v=v, x=x, eps=eps, next=nxt, log_prob=lp, mean=mean, std=std.reshape(-1),
                         log_prob_replay=lp_r, std_replay=std_r.reshape(-1),
                         sigmas=sch.sigmas, timesteps=sch.timesteps).items():
            out[f"{name}/{k}"] = a.numpy()
        out[f"{name}/meta"] = np.array([nsteps, idx, nl], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "sde_step.npz"), **out)


def gold_sampler():
    ns = extract_defs(TP, {"DistributedKRepeatSampler"},
                      {"torch": torch, "Sampler": torch.utils.data.Sampler})
    cls = ns["DistributedKRepeatSampler"]
    out = []
    for (n, b, k, dlen) in [(1, 1, 1, 25432), (8, 1, 1, 25432), (8, 1, 2, 25432), (2, 3, 2, 97), (4, 2, 4, 1000)]:
        for epoch in range(4):
            per_rank = []
            for r in range(n):
                s = cls(list(range(dlen)), b, k, n, r, seed=42)
                s.set_epoch(epoch)
                per_rank.append(next(iter(s)))
            out.append(dict(n=n, b=b, k=k, dataset_len=dlen, seed=42, epoch=epoch, per_rank=per_rank))
    json.dump(out, open(os.path.join(HERE, "sampler.json"), "w"))


def gold_stat_tracker():
    from adv_grpo.stat_tracking import PerPromptStatTracker
    ns = extract_defs(TP, {"calculate_zero_std_ratio"}, {"np": np})
    zsr = ns["calculate_zero_std_ratio"]
    out = {}
    rng = np.random.RandomState(5)
    # (a) the reference's own __main__ smoke case (stat_tracking.py:81-94)
    cases = {"toy": (["a", "b", "a", "c", "b", "a"], np.array([1, 2, 3, 4, 5, 6], dtype=np.float64))}
    # (b) epoch-shaped: 48 prompts x 16 images, rewards [768, 2] float32 (TP:926-930)
    ids = np.repeat(np.arange(48), 16); rng.shuffle(ids)
    r = rng.randn(768).astype(np.float32) * 0.05 + 0.8
    cases["epoch"] = ([f"prompt number {i}" for i in ids], np.stack([r, r], 1))
    # (c) a zero-std group and a singleton group
    ids3 = np.array([0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3])
    r3 = rng.randn(13).astype(np.float32); r3[4:8] = 0.25
    cases["zero_std"] = ([f"p{i}" for i in ids3], np.stack([r3, r3 * 2], 1))
    for name, (prompts, rewards) in cases.items():
        uniq = {p: i for i, p in enumerate(sorted(set(prompts)))}
        out[f"{name}/group_ids"] = np.array([uniq[p] for p in prompts], dtype=np.int32)
        out[f"{name}/rewards"] = rewards
        for gs in (True, False):
            tr = PerPromptStatTracker(gs)
            adv = tr.update(prompts, rewards)
            out[f"{name}/adv_global{int(gs)}"] = adv
            out[f"{name}/stats{int(gs)}"] = np.array(tr.get_stats(), dtype=np.float64)
        ori = rewards if rewards.ndim == 1 else rewards[:, 0]
        out[f"{name}/zero_std"] = np.array(zsr(prompts, {"ori_avg": ori}), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "stat_tracker.npz"), **out)


def gold_losses():
    out = {}
    # ---- GRPO loss block, exec'd from the trainer's source text (TP:1111-1162)
    block = extract_block(TP, "# grpo logic", 'info["loss"].append(loss)')
    from collections import defaultdict
    for name, clip, scale in [("a", 1e-5, 3e-5), ("b", 1e-4, 1e-3), ("c", 0.2, 0.5)]:
        g = torch.Generator().manual_seed(len(name) + int(clip * 1e6))
        old = -torch.rand(8, 2, generator=g)
        lp = (old[:, 0] + torch.randn(8, generator=g) * scale).requires_grad_(True)
        adv = torch.randn(8, 2, generator=g) * 3
        cfg = types.SimpleNamespace(train=types.SimpleNamespace(adv_clip_max=5, clip_range=clip, beta=0.0))
        ns = dict(torch=torch, config=cfg, sample={"advantages": adv, "log_probs": old}, j=0,
                  log_prob=lp, info=defaultdict(list))
        exec(block, ns)
        ns["loss"].backward()
        out[f"grpo_{name}/log_prob"] = lp.detach().numpy(); out[f"grpo_{name}/old"] = old[:, 0].numpy()
        out[f"grpo_{name}/adv"] = adv[:, 0].numpy(); out[f"grpo_{name}/clip"] = np.array(clip)
        out[f"grpo_{name}/grad"] = lp.grad.numpy()
        for k, v in ns["info"].items():
            out[f"grpo_{name}/{k}"] = v[0].detach().numpy()
    # ---- the same block with config.train.beta > 0: the KL term against prev_sample_mean_ref (TP:1126-1130,1158-1160)
    g = torch.Generator().manual_seed(77)
    old = -torch.rand(6, 2, generator=g)
    lp = (old[:, 0] + torch.randn(6, generator=g) * 2e-5).requires_grad_(True)
    adv = torch.randn(6, 2, generator=g) * 3
    mean = torch.randn(6, 4, 8, 8, generator=g).requires_grad_(True)
    mean_ref = (mean.detach() + 0.05 * torch.randn(6, 4, 8, 8, generator=g))
    cfg = types.SimpleNamespace(train=types.SimpleNamespace(adv_clip_max=5, clip_range=1e-5, beta=0.04))
    ns = dict(torch=torch, config=cfg, sample={"advantages": adv, "log_probs": old}, j=0, log_prob=lp, info=defaultdict(list),
              prev_sample_mean=mean, prev_sample_mean_ref=mean_ref)
    exec(block, ns)
    ns["loss"].backward()
    out["grpo_kl/log_prob"] = lp.detach().numpy(); out["grpo_kl/old"] = old[:, 0].numpy(); out["grpo_kl/adv"] = adv[:, 0].numpy()
    out["grpo_kl/mean"] = mean.detach().numpy(); out["grpo_kl/mean_ref"] = mean_ref.numpy()
    out["grpo_kl/beta"] = np.array(0.04); out["grpo_kl/clip"] = np.array(1e-5)
    out["grpo_kl/grad_log_prob"] = lp.grad.numpy(); out["grpo_kl/grad_mean"] = mean.grad.numpy()
    for k, v in ns["info"].items():
        out[f"grpo_kl/{k}"] = v[0].detach().numpy()
    # ---- CLIPCriterion.calc_loss (pick_score_training.py:118-199)
    from adv_grpo.pick_score_training import CLIPCriterion, CLIPCriterionConfig
    crit = CLIPCriterion(CLIPCriterionConfig())
    g = torch.Generator().manual_seed(0)
    f = [torch.nn.functional.normalize(torch.randn(6, 32, generator=g), dim=-1) for _ in range(3)]
    loss = crit.calc_loss(f[0], f[1], f[2], torch.tensor(100.0), torch.tensor(1.0), torch.tensor(0.0),
                          torch.tensor(1.0))
    out["clip/text"], out["clip/img0"], out["clip/img1"] = (a.numpy() for a in f)
    out["clip/loss"] = loss.numpy()
    # ---- the same criterion with the batch's labels, through CLIPCriterion.forward(model, batch) (pick_score_training.py:205-224) on a
    # stand-in model that returns fixed, bf16-exact, UN-normalised features (forward normalises them, :103-104): real-vs-fake (1, 0),
    # reversed (0, 1), a tie (0.5, 0.5: the log(0.5) offset of :181-183) and per-example labels
    g = torch.Generator().manual_seed(11)
    B, P = 5, 48
    raw = [(torch.randn(B, P, generator=g) * 3).to(torch.bfloat16).float() for _ in range(3)]

    class Feat:
        logit_scale = torch.tensor(2.0).log()
        def get_text_features(self, input_ids): return raw[0]
        def get_image_features(self, pixel_values): return torch.cat([raw[1], raw[2]])[:pixel_values.shape[0]]
    out["clipl/text"], out["clipl/img0"], out["clipl/img1"] = (a.numpy() for a in raw)
    out["clipl/logit_scale_exp"] = np.array(2.0, dtype=np.float32)
    cases = {"real_fake": (torch.tensor(1.0), torch.tensor(0.0)), "fake_real": (torch.tensor(0.0), torch.tensor(1.0)),
             "tie": (torch.tensor(0.5), torch.tensor(0.5)),
             "mixed": (torch.tensor([1.0, 0.0, 0.5, 1.0, 0.25]), torch.tensor([0.0, 1.0, 0.5, 0.0, 0.75]))}
    for name, (l0, l1) in cases.items():
        batch = {"input_ids": torch.zeros(B, 77, dtype=torch.long), "pixels_0": torch.zeros(B, 3, 2, 2), "pixels_1": torch.zeros(B, 3, 2, 2),
                 "label_0": l0, "label_1": l1, "num_examples_per_prompt": torch.tensor(1.0)}
        out[f"clipl/{name}/label_0"], out[f"clipl/{name}/label_1"] = l0.numpy(), l1.numpy()
        out[f"clipl/{name}/loss"] = crit(Feat(), batch).numpy()
    # ---- train_dino (TD:156-232) with a stand-in backbone returning fixed features
    from PIL import Image
    from oracle.losses import DinoHead
    ns = extract_defs(TD, {"train_dino"}, {"torch": torch})
    B, N, D = 4, 100, 24
    g = torch.Generator().manual_seed(3)
    feats = [torch.randn(B, N + 1, D, generator=g), torch.randn(B, N + 1, D, generator=g)]

    class Backbone:
        def __init__(self): self.calls = 0
        def eval(self): return self
        def forward_features(self, x):
            self.calls += 1
            return feats[self.calls - 1]
    torch.manual_seed(21)
    head = DinoHead(D, 16)
    out["dino/head"] = np.concatenate([p.detach().numpy().ravel() for p in head.parameters()])
    opt = torch.optim.Adam(head.parameters(), lr=1e-3, betas=(0.5, 0.999))
    pil = [Image.fromarray(np.full((8, 8, 3), 100 + i, dtype=np.uint8)) for i in range(B)]
    torch.manual_seed(77)
    d_loss, acc = ns["train_dino"](Backbone(), head, None, pil, pil, opt,
                                   types.SimpleNamespace(device="cpu"), n_patches=64, patch_loss_weight=0.3)
    torch.manual_seed(77)
    out["dino/idx_real"] = torch.randint(0, N, (B, 64)).numpy()
    out["dino/idx_fake"] = torch.randint(0, N, (B, 64)).numpy()
    out["dino/feats_real"], out["dino/feats_fake"] = feats[0].numpy(), feats[1].numpy()
    out["dino/d_loss"], out["dino/acc"] = np.array(d_loss), np.array(acc)
    out["dino/head_after"] = np.concatenate([p.detach().numpy().ravel() for p in head.parameters()])
    # ---- EMA (ema.py:33-52)
    from adv_grpo.ema import EMAModuleWrapper
    p = [torch.nn.Parameter(torch.arange(6.0).reshape(2, 3))]
    ema = EMAModuleWrapper(p, decay=0.9, update_step_interval=8, device="cpu")
    decays, snaps = [], []
    for step in range(40):
        with torch.no_grad():
            p[0].add_(0.5)
        ema.step(p, step)
        decays.append(ema.get_current_decay(step)); snaps.append(ema.ema_parameters[0].clone().numpy())
    out["ema/decay"] = np.array(decays); out["ema/params"] = np.stack(snaps)
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)


def gold_rewards():
    import adv_grpo.rewards as RW
    from oracle.losses import DinoHead
    out = {}
    # ---- multi_score aggregation with two toy scorers registered through the reference registry path:
    # multi_score builds factories by name, so drive _fn through dino_patch_cotrain + a plain scorer.
    B, N, D = 3, 100, 24
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(B, N + 1, D, generator=g)

    class Backbone:
        def forward_features(self, x): return feats.to(x.dtype)
    torch.manual_seed(4)
    head = DinoHead(D, 16)
    images = torch.rand(B, 3, 32, 32, generator=g)
    fn = RW.multi_score("cpu", {"dino_patch_cotrain": 1.0})
    torch.manual_seed(55)
    details, _ = fn(images.to(torch.bfloat16), ["p"] * B, [{}] * B, scorer=Backbone(), head=head.to(torch.bfloat16))
    torch.manual_seed(55)
    idx = torch.randint(0, N, (B, 64))
    out["dino_patch/feats"] = feats.numpy(); out["dino_patch/idx"] = idx.numpy()
    out["dino_patch/head"] = np.concatenate([p.detach().float().numpy().ravel() for p in head.parameters()])
    out["dino_patch/scores"] = details["dino_patch_cotrain"].float().numpy()
    out["dino_patch/avg"] = np.array([float(a) for a in details["avg"]])
    # preprocess alone (rewards.py:379-391), fp32 in -> bf16 out
    pre = RW.dino_patch_cotrain_score.__code__.co_consts  # noqa: F841 (nested fn not addressable); use _fn path:
    class Capture:
        def forward_features(self, x):
            out["dino_patch/preprocessed"] = x.float().numpy()[:, :, ::37, ::37]
            return feats.to(x.dtype)
    torch.manual_seed(55)
    fn(images.to(torch.bfloat16), ["p"] * B, [{}] * B, scorer=Capture(), head=head)
    out["dino_patch/images"] = images.numpy()
    # ---- weighted sum over two scorers: monkey-register toy factories under existing names
    def toy_a(device):
        return lambda images, prompts, metadata: (torch.tensor([0.25, 0.5, 1.0]), {})
    def toy_b():
        return lambda images, prompts, metadata: (np.array([3.0, 2.0, 1.0]), {})
    saved = RW.aesthetic_score, RW.jpeg_compressibility
    RW.aesthetic_score, RW.jpeg_compressibility = toy_a, toy_b
    try:
        fn2 = RW.multi_score("cpu", {"aesthetic": 0.3, "jpeg_compressibility": 0.7})
        det, _ = fn2(None, ["p"] * 3, [{}] * 3)
    finally:
        RW.aesthetic_score, RW.jpeg_compressibility = saved
    out["multi/avg"] = np.array([float(a) for a in det["avg"]])
    np.savez_compressed(os.path.join(HERE, "rewards.npz"), **out)


def gold_rollout():
    from adv_grpo.diffusers_patch.sd3_pipeline_with_logprob_fast import pipeline_with_logprob_random as ref_rollout
    from adv_grpo.diffusers_patch.sd3_sde_with_logprob import sde_step_with_logprob_new as ref_step
    ns = extract_defs(TP, {"compute_log_prob"}, {"torch": torch, "sde_step_with_logprob": ref_step})
    ref_clp = ns["compute_log_prob"]
    import contextlib
    out = {}
    for name, dtype, steps, T, G, hw in [("fp32", torch.float32, 4, 2, 2, 64), ("bf16", torch.bfloat16, 10, 2, 4, 64)]:
        net = StandinVelocity()
        sch = FlowMatchEulerScheduler()
        g = torch.Generator().manual_seed(17)
        pe = torch.randn(1, 5, 32, generator=g).to(dtype); ppe = torch.randn(1, 16, generator=g).to(dtype)
        npe = torch.randn(1, 5, 32, generator=g).to(dtype); nppe = torch.randn(1, 16, generator=g).to(dtype)
        lat0 = torch.randn(G, 16, hw // 8, hw // 8, generator=g)

        class Pipe:
            default_sample_size = 8; vae_scale_factor = 8
            transformer = net; scheduler = sch
            _execution_device = "cpu"
            def check_inputs(self, *a, **k): pass
            def encode_prompt(self, **k):
                return (k["prompt_embeds"], k["negative_prompt_embeds"], k["pooled_prompt_embeds"],
                        k["negative_pooled_prompt_embeds"])
            def prepare_latents(self, b, c, h, w, dt, device, generator, latents):
                return lat0.to(dt)
            @property
            def do_classifier_free_guidance(self): return self._guidance_scale > 1
            @property
            def guidance_scale(self): return self._guidance_scale
            @property
            def clip_skip(self): return self._clip_skip
            @property
            def joint_attention_kwargs(self): return self._joint_attention_kwargs
            def progress_bar(self, total=None):
                class PB:
                    def update(s): pass
                return contextlib.nullcontext(PB())
            def maybe_free_model_hooks(self): pass
        pipe = Pipe()
        pipe.transformer.config = types.SimpleNamespace(in_channels=16)
        pipe.vae = types.SimpleNamespace(config=types.SimpleNamespace(scaling_factor=1.5305, shift_factor=0.0609),
                                         dtype=torch.float32,
                                         decode=lambda z, return_dict=False: (standin_vae_decode(z),))
        pipe.image_processor = types.SimpleNamespace(
            postprocess=lambda im, output_type="pt": (im / 2 + 0.5).clamp(0, 1))
        torch.manual_seed(99)
        noises = [torch.randn(G, 16, hw // 8, hw // 8) for _ in range(steps)]
        torch.manual_seed(99)
        with torch.no_grad():
            image, lats, lps, tss = ref_rollout(
                pipe, prompt_embeds=pe, pooled_prompt_embeds=ppe, negative_prompt_embeds=npe,
                negative_pooled_prompt_embeds=nppe, num_inference_steps=steps, guidance_scale=4.5,
                output_type="pt", height=hw, width=hw, noise_level=0.8, mini_num_image_per_prompt=G,
                train_num_steps=T, process_index=0, sample_num_steps=steps, random_timestep=0)
        lat = torch.stack(lats, 1); lp = torch.stack(lps, 1); ts = torch.stack(tss, 1)
        sample = {"latents": lat[:, :-1], "next_latents": lat[:, 1:], "timesteps": ts, "log_probs": lp}
        cfg = types.SimpleNamespace(train=types.SimpleNamespace(cfg=True),
                                    sample=types.SimpleNamespace(guidance_scale=4.5, noise_level=0.8))
        embeds = torch.cat([npe.repeat(G, 1, 1), pe.repeat(G, 1, 1)])
        pooled = torch.cat([nppe.repeat(G, 1), ppe.repeat(G, 1)])
        replay_lp, replay_mean = [], []
        for j in range(T):
            with torch.no_grad():
                _, lpj, mj, _ = ref_clp(net, types.SimpleNamespace(scheduler=sch), sample, j, embeds, pooled, cfg)
            replay_lp.append(lpj); replay_mean.append(mj)
        f = lambda a: a.float().numpy()
        out.update({f"{name}/pe": f(pe), f"{name}/ppe": f(ppe), f"{name}/npe": f(npe), f"{name}/nppe": f(nppe),
                    f"{name}/lat0": f(lat0), f"{name}/noises": torch.stack(noises).numpy(),
                    f"{name}/image": f(image), f"{name}/latents": f(lat), f"{name}/log_probs": f(lp),
                    f"{name}/timesteps": f(ts), f"{name}/replay_log_probs": f(torch.stack(replay_lp, 1)),
                    f"{name}/replay_mean": f(torch.stack(replay_mean, 1)),
                    f"{name}/meta": np.array([steps, T, G, hw], dtype=np.int64)})
    np.savez_compressed(os.path.join(HERE, "rollout.npz"), **out)


if __name__ == "__main__":
    install_shims()
    gold_sde(); gold_sampler(); gold_stat_tracker(); gold_losses(); gold_rewards(); gold_rollout()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
