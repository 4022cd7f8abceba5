"""LoRA on the MMDiT attention projections, as peft applies it in the reference (test infrastructure).

Restates scripts/train_sd3_fast_pickscore.py:490-511: LoraConfig(r=32, lora_alpha=64, init_lora_weights="gaussian",
target_modules = attn.{add_k_proj, add_q_proj, add_v_proj, to_add_out, to_k, to_out.0, to_q, to_v}) on every
transformer block (suffix match, so attn2.* is not adapted).  peft computes y = W x + b + (alpha/r) B (A x); in exact
arithmetic that equals (W + (alpha/r) B A) x + b, which is what effective_weights() builds (differentiable w.r.t. A, B).
PARITY UNPINNED (peft is absent from the image); the arithmetic identity above is the whole content."""
import torch

TARGETS = ("to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj", "to_add_out")


def adapter_names(cfg):
    out = []
    for i in range(cfg.num_layers):
        for n in TARGETS:
            if n == "to_add_out" and i == cfg.num_layers - 1:
                continue
            out.append(f"transformer_blocks.{i}.attn.{n}")
    return out


def init_lora(cfg, r=32, seed=0, zero_b=True):
    g = torch.Generator().manual_seed(seed)
    D = cfg.dim
    sd = {}
    for n in adapter_names(cfg):
        sd[n + ".lora_A.weight"] = torch.randn(r, D, generator=g) / r
        sd[n + ".lora_B.weight"] = torch.zeros(D, r) if zero_b else torch.randn(D, r, generator=g) * 0.02
    return sd


def effective_weights(W, lora, alpha=64, r=32):
    """dict like W with the adapted projection weights replaced by W + (alpha/r) B A (autograd-tracked)."""
    out = dict(W)
    s = alpha / r
    for name in {k.rsplit(".lora_", 1)[0] for k in lora}:
        A, B = lora[name + ".lora_A.weight"], lora[name + ".lora_B.weight"]
        out[name + ".weight"] = W[name + ".weight"] + s * (B.to(W[name + ".weight"].dtype) @ A.to(W[name + ".weight"].dtype))
    return out
