#!/usr/bin/env python3
"""Decode time of one GRPO group (8 latents, 64x64x16 -> 512^2) in the two VAE modes, same box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adv_grpo_amd import synthetic
from adv_grpo_amd.model_configs import VaeConfig
from adv_grpo_amd.vae import AutoencoderKLDecoder

cfg = VaeConfig()
W = synthetic.vae_decoder_weights(cfg, 4321)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lat = torch.randn(B, 16, hw, hw, device="cuda").to(torch.bfloat16)
for mode in ("bf16", "bf16x3", "bf16x3 one stream", "bf16x3 3 streams", "bf16x3 4 streams"):
    dec = AutoencoderKLDecoder(W, cfg, "cuda", mode=mode.split()[0])
    dec.two_streams = "one" not in mode
    if "streams" in mode:
        dec.n_streams = int(mode.split()[1])
    for _ in range(2):
        dec.decode_to_image(lat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dec.decode_to_image(lat)
    torch.cuda.synchronize()
    print(f"{mode}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per {B} x {8*hw}^2 decode, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
