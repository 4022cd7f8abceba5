"""Decode of 8 x 512^2 images with the batch split over 1 / 2 / 4 / 8 HIP streams (AutoencoderKLDecoder.n_streams)."""
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
ref = None
for rep in range(2):
    for n in (1, 2, 4, 8):
        vae.n_streams = n
        for _ in range(2):
            vae.decode_to_image(lat)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            img = vae.decode_to_image(lat)
        e.record()
        torch.cuda.synchronize()
        ref = img if ref is None else ref
        print(f"n_streams={n}: {s.elapsed_time(e) / 5:.2f} ms per decode of 8 x 512^2; bit-identical to 1 stream: {torch.equal(img, ref)}")
# side streams measured to overlap with the calling stream (ops.concurrent_stream) instead of torch's next pool stream
from adv_grpo_amd import ops  # noqa: E402
main = torch.cuda.current_stream(dev)
vae.n_streams = 2
vae._side[main.cuda_stream] = [ops.concurrent_stream(dev, [main])]
for rep in range(2):
    for _ in range(2):
        vae.decode_to_image(lat)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        img = vae.decode_to_image(lat)
    e.record()
    torch.cuda.synchronize()
    print(f"n_streams=2, measured-concurrent side stream: {s.elapsed_time(e) / 5:.2f} ms; bit-identical: {torch.equal(img, ref)}")
