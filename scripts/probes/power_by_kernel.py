"""Socket power and shader clock while ONE kind of kernel runs back to back (rocm-smi sampled on a host thread): which parts of
the rollout sit at the package power limit and which have headroom.  Usage: power_by_kernel.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import PowerSampler
from adv_grpo_amd import ops, synthetic
from adv_grpo_amd.model_configs import VaeConfig
from adv_grpo_amd.vae import AutoencoderKLDecoder
bf = torch.bfloat16
rnd = lambda *s, k=1.0: (torch.randn(*s, device="cuda") * k).to(bf)


def run(name, fn, seconds=6.0):
    fn(); torch.cuda.synchronize()
    with PowerSampler(0, period=0.4) as p:
        t0 = time.time(); n = 0
        while time.time() - t0 < seconds:
            for _ in range(10):
                fn()
            torch.cuda.synchronize(); n += 10
        dt = time.time() - t0
    s = p.summary()
    print(f"{name:28s} {dt / n * 1e6:9.1f} us/call  sclk {s['sclk_mhz_median']} MHz ({s['sclk_mhz_min']}-{s['sclk_mhz_max']})  power {s['socket_power_w_median']:.0f} W (max {s['socket_power_w_max']:.0f})")


x, w, b = rnd(19664, 1536), rnd(6144, 1536, k=0.05), rnd(6144)
o = torch.empty(19664, 6144, dtype=bf, device="cuda")
run("gemm8p FF1 bf16", lambda: ops.gemm(x, w, bias=b, act="gelu_tanh", out=o))
x2, w2, b2 = rnd(19664, 6144), rnd(1536, 6144, k=0.05), rnd(1536)
o2 = torch.empty(19664, 1536, dtype=bf, device="cuda")
run("gemm8p FF2 bf16 (K=6144)", lambda: ops.gemm(x2, w2, bias=b2, out=o2))
qx, qw = ops.quant_fp8_rows(x), ops.quant_fp8_rows(w)
run("gemm8p FF1 fp8", lambda: ops.gemm_grouped_fp8([ops.gemm_desc_fp8(qx, qw, bias=b, act="gelu_tanh", out=o)]))
qkv = rnd(16, 1229, 3 * 1536)
ao = torch.empty(16, 1229, 1536, dtype=bf, device="cuda")
run("attention fwd S=1229", lambda: ops.attention(qkv[..., :1536], qkv[..., 1536:3072], qkv[..., 3072:], 24, out=ao))
with synthetic.on_device("cuda"):
    vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 4321), VaeConfig(), "cuda", mode="bf16x3")
    vae16 = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 4321), VaeConfig(), "cuda", mode="bf16")
lat = rnd(8, 16, 64, 64)
run("VAE decode bf16x3 (8 img)", lambda: vae.decode_to_image(lat))
run("VAE decode bf16 (8 img)", lambda: vae16.decode_to_image(lat))
ln_x, sc = rnd(19664, 1536), rnd(16, 3072, k=0.3)
run("layernorm_mod", lambda: ops.layernorm_mod(ln_x, scale=sc[:, :1536], shift=sc[:, 1536:], rows_per_batch=1229))
