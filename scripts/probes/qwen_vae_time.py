"""Decode time of the Qwen-Image VAE decoder (8 x 1024^2, both modes) next to the SD3 decoder that stood in for it."""
import sys, time
import torch
sys.path.insert(0, ".")
from adv_grpo_amd import synthetic
from adv_grpo_amd.model_configs import QwenVaeConfig, VaeConfig
from adv_grpo_amd.qwen_vae import AutoencoderKLQwenImageDecoder, flops_decode
from adv_grpo_amd.vae import AutoencoderKLDecoder

dev = "cuda"
lat = torch.randn(8, 16, 128, 128, device=dev).to(torch.bfloat16)
def t(dec, n=3):
    dec.decode_to_image(lat); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        dec.decode_to_image(lat)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
qc = QwenVaeConfig()
for mode in ("bf16x3", "bf16"):
    with synthetic.on_device(dev):
        d = AutoencoderKLQwenImageDecoder(synthetic.qwen_vae_decoder_weights(qc, 2468, dtype=torch.bfloat16), qc, dev, mode=mode)
    ms = t(d)
    print(f"qwen vae {mode}: {ms:.1f} ms per 8 x 1024^2  ({8 * flops_decode(qc, 128, 128) / ms / 1e9:.0f} TFLOP/s algorithmic)")
    del d
vc = VaeConfig()
with synthetic.on_device(dev):
    d = AutoencoderKLDecoder(synthetic.vae_decoder_weights(vc, 4321, fp16_checkpoint=True), vc, dev, mode="bf16x3")
print(f"sd3 vae bf16x3 (f16x2 convs): {t(d):.1f} ms per 8 x 1024^2")
