"""The PickScore image tower alone (CLIP ViT-H/14, 8 images of 512 x 512 -> 224 x 224, M = 2056 rows): time per call and, under rocprofv3, per kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import synthetic, vit  # noqa: E402
from adv_grpo_amd.model_configs import ClipConfig  # noqa: E402

dev = torch.device("cuda", 0)
cfg = ClipConfig()
with synthetic.on_device(dev):
    clip = vit.CLIPModel(synthetic.clip_weights(cfg, 777), cfg, dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
img = torch.rand(B, 3, 512, 512, device=dev).to(torch.bfloat16)
for _ in range(3):
    clip.get_image_features(images=img)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(N):
    clip.get_image_features(images=img)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / N
v = cfg.vision if hasattr(cfg, "vision") else None
print(f"image tower, batch {B}: {ms:.2f} ms per call")
