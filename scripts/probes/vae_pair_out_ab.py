"""Same-box A/B of the pair-output epilogue (conv2 in front of an upsampler writes the upsampler operand rows): decode of 8 x 512^2 images."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import synthetic  # noqa: E402
from adv_grpo_amd.model_configs import VaeConfig  # noqa: E402
from adv_grpo_amd.vae import AutoencoderKLDecoder  # noqa: E402

dev = torch.device("cuda", 0)
cfg = VaeConfig()
vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(cfg, 99, fp16_checkpoint=True), cfg, dev, mode="bf16x3")
lat = torch.randn(8, 16, 64, 64, device=dev).to(torch.bfloat16)
imgs = {}
for rep in range(3):
    for flag in (True, False):
        vae.fused_pair_out = flag
        for _ in range(2):
            vae.decode_to_image(lat)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            img = vae.decode_to_image(lat)
        e.record()
        torch.cuda.synchronize()
        imgs[flag] = img
        print(f"fused_pair_out={flag}: {s.elapsed_time(e) / 5:.2f} ms per decode of 8 x 512^2")
print("max |image difference|:", (imgs[True] - imgs[False]).abs().max().item())
