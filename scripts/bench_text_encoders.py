"""Prompt-encoding time at the real SD3 sizes (CLIP-L 12x768, CLIP-G 32x1280, T5-XXL v1.1 24x4096) with random weights."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import text_encoders as te
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s, sc=0.02: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
def clip_sd(D, ff, L, P, vocab=49408):
    sd = {"text_model.embeddings.token_embedding.weight": R(vocab, D), "text_model.embeddings.position_embedding.weight": R(77, D),
          "text_model.final_layer_norm.weight": torch.ones(D, device=dev), "text_model.final_layer_norm.bias": torch.zeros(D, device=dev),
          "text_projection.weight": R(P, D)}
    for i in range(L):
        p = f"text_model.encoder.layers.{i}"
        for n in ("layer_norm1", "layer_norm2"):
            sd[f"{p}.{n}.weight"] = torch.ones(D, device=dev); sd[f"{p}.{n}.bias"] = torch.zeros(D, device=dev)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[f"{p}.self_attn.{n}.weight"] = R(D, D); sd[f"{p}.self_attn.{n}.bias"] = torch.zeros(D, device=dev)
        sd[f"{p}.mlp.fc1.weight"] = R(ff, D); sd[f"{p}.mlp.fc1.bias"] = torch.zeros(ff, device=dev)
        sd[f"{p}.mlp.fc2.weight"] = R(D, ff); sd[f"{p}.mlp.fc2.bias"] = torch.zeros(D, device=dev)
    return sd
def t5_sd(D=4096, H=64, ff=10240, L=24, vocab=32128):
    sd = {"shared.weight": R(vocab, D, sc=1.0), "encoder.final_layer_norm.weight": torch.ones(D, device=dev),
          "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": R(32, H, sc=1.0)}
    for i in range(L):
        p = f"encoder.block.{i}.layer"
        sd[f"{p}.0.layer_norm.weight"] = torch.ones(D, device=dev); sd[f"{p}.1.layer_norm.weight"] = torch.ones(D, device=dev)
        for n in ("q", "k", "v", "o"):
            sd[f"{p}.0.SelfAttention.{n}.weight"] = R(H * 64, D) if n != "o" else R(D, H * 64)
        sd[f"{p}.1.DenseReluDense.wi_0.weight"] = R(ff, D); sd[f"{p}.1.DenseReluDense.wi_1.weight"] = R(ff, D)
        sd[f"{p}.1.DenseReluDense.wo.weight"] = R(D, ff)
    return sd
cl = te.CLIPTextEncoder(clip_sd(768, 3072, 12, 768), 12, 12, "quick_gelu", 2, dev)
cg = te.CLIPTextEncoder(clip_sd(1280, 5120, 32, 1280), 32, 20, "gelu", 2, dev)
t5 = te.T5Encoder(t5_sd(), 24, 64, device=dev)
for B in (1, 16):
    ids = torch.randint(3, 30000, (B, 77), device=dev); ids_t5 = torch.randint(3, 30000, (B, 128), device=dev)
    for _ in range(2): pe, pooled = te.encode_prompt(cl, cg, t5, ids, ids, ids_t5)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): pe, pooled = te.encode_prompt(cl, cg, t5, ids, ids, ids_t5)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f"encode_prompt B={B}: {dt * 1e3:.2f} ms  -> {tuple(pe.shape)} {tuple(pooled.shape)} finite={bool(torch.isfinite(pe.float()).all())}")
