"""fp32-equivalent decode of one group (8 latents 64 x 64 -> 512^2; argv: batch, latent side) under the convolution's wide-tile threshold
(ADVGRPO_X3_WIDE_MIN, experiments build): time (one stream and the product's two-stream split), sha of the image."""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import synthetic  # noqa: E402
from adv_grpo_amd.model_configs import VaeConfig  # noqa: E402
from adv_grpo_amd.vae import AutoencoderKLDecoder  # noqa: E402

cfg = VaeConfig()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lat = torch.randn(B, 16, hw, hw, generator=torch.Generator().manual_seed(11)).to(torch.bfloat16).cuda()
tag = os.environ.get("ADVGRPO_X3_WIDE_MIN", "default")
for single in (False, True):
    dec = AutoencoderKLDecoder(synthetic.vae_decoder_weights(cfg, 4321, fp16_checkpoint=True), cfg, "cuda", mode="bf16x3", f16_single=single)
    for two in (False, True):
        dec.two_streams = two
        for _ in range(2):
            img = dec.decode_to_image(lat)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            img = dec.decode_to_image(lat)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        sha = hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"WIDE_MIN={tag:>10s} {'f16x1' if single else 'f16x2'} {'two streams' if two else 'one stream '}: median {ts[3]:6.2f} ms  min {ts[0]:6.2f} ms  sha {sha}")
