"""Which GEMM launches of the PickScore scorer (CLIP ViT-H/14, 8 images + 1 prompt) cost what: HIP events around every launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import ops, synthetic, vit  # noqa: E402
from adv_grpo_amd.model_configs import ClipConfig  # noqa: E402

dev = torch.device("cuda", 0)
cfg = ClipConfig()
with synthetic.on_device(dev):
    clip = vit.CLIPModel(synthetic.clip_weights(cfg, 777), cfg, dev)
img = torch.rand(8, 3, 512, 512, device=dev).to(torch.bfloat16)
ids = synthetic.clip_input_ids(1, 3).to(dev)


def score():
    return vit.pickscore_scores(clip.get_image_features(images=img), clip.get_text_features(ids).expand(8, -1).contiguous(), clip.logit_scale)


for _ in range(3):
    score()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    score()
e.record()
torch.cuda.synchronize()
print(f"scorer call: {s.elapsed_time(e) / 10:.2f} ms")
# the product scorer object: text tower on its own stream beside the image tower
from adv_grpo_amd.pickscore_scorer import PickScoreScorer  # noqa: E402
from adv_grpo_amd import rewards  # noqa: E402
sc = PickScoreScorer(dev, dtype=torch.bfloat16, model_sd=synthetic.clip_weights(cfg, 777), clip_cfg=cfg)
ids8 = ids.expand(8, -1).contiguous()
prompts = rewards.PromptBatch(["a prompt"] * 8, ids8) if hasattr(rewards, "PromptBatch") else ids8
for _ in range(3):
    sc(prompts, img)
torch.cuda.synchronize()
s.record()
for _ in range(10):
    sc(prompts, img)
e.record()
torch.cuda.synchronize()
print(f"PickScoreScorer call (text tower on a side stream, one unique prompt): {s.elapsed_time(e) / 10:.2f} ms")
ops.PROFILE, ops.PROFILE_STRIDE = [], 1
for _ in range(3):
    score()
torch.cuda.synchronize()
per = {}
for name, fl, a, b, shape in ops.PROFILE:
    k = (name, shape[:3])
    v = per.setdefault(k, [0, 0.0, fl])
    v[0] += 1
    v[1] += a.elapsed_time(b)
tot = sum(v[1] for v in per.values()) / 3
print(f"GEMM launches: {sum(v[0] for v in per.values()) // 3} per call, {tot:.2f} ms per call")
for (name, shape), (n, ms, fl) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:42s} M,N,K={shape}  n={n // 3:3d}  {ms / n * 1e3:7.1f} us each  {ms / 3:6.2f} ms per call  {fl / (ms / n * 1e-3) / 1e12:6.1f} TFLOP/s")
