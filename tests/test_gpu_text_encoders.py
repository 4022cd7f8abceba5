"""GPU: prompt encoding (CLIP-L / CLIP-G text towers + T5 v1.1 encoder) on the HIP kernels vs the oracle
(oracle/text_encoders.py, itself pinned against transformers and the reference's encode_prompt)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clip_sd(D, ff, L, P, vocab, g):
    sd = {"text_model.embeddings.token_embedding.weight": torch.randn(vocab, D, generator=g) * 0.5,
          "text_model.embeddings.position_embedding.weight": torch.randn(77, D, generator=g) * 0.1,
          "text_model.final_layer_norm.weight": 1 + 0.1 * torch.randn(D, generator=g),
          "text_model.final_layer_norm.bias": 0.1 * torch.randn(D, generator=g),
          "text_projection.weight": torch.randn(P, D, generator=g) * D ** -0.5}
    for i in range(L):
        p = f"text_model.encoder.layers.{i}"
        for n in ("layer_norm1", "layer_norm2"):
            sd[f"{p}.{n}.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
            sd[f"{p}.{n}.bias"] = 0.1 * torch.randn(D, generator=g)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[f"{p}.self_attn.{n}.weight"] = torch.randn(D, D, generator=g) * D ** -0.5
            sd[f"{p}.self_attn.{n}.bias"] = 0.1 * torch.randn(D, generator=g)
        sd[f"{p}.mlp.fc1.weight"] = torch.randn(ff, D, generator=g) * D ** -0.5
        sd[f"{p}.mlp.fc1.bias"] = 0.1 * torch.randn(ff, generator=g)
        sd[f"{p}.mlp.fc2.weight"] = torch.randn(D, ff, generator=g) * ff ** -0.5
        sd[f"{p}.mlp.fc2.bias"] = 0.1 * torch.randn(D, generator=g)
    return {k: v.to(torch.bfloat16).float() for k, v in sd.items()}        # bf16-representable: both sides see the same weights


def _t5_sd(D, H, ff, L, vocab, g):
    inner = H * 64
    sd = {"shared.weight": torch.randn(vocab, D, generator=g), "encoder.final_layer_norm.weight": 1 + 0.1 * torch.randn(D, generator=g),
          "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": torch.randn(32, H, generator=g)}
    for i in range(L):
        p = f"encoder.block.{i}.layer"
        sd[f"{p}.0.layer_norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
        sd[f"{p}.1.layer_norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
        for n in ("q", "k", "v"):
            sd[f"{p}.0.SelfAttention.{n}.weight"] = torch.randn(inner, D, generator=g) * (D ** -0.5 if n != "q" else (D * 64) ** -0.5)
        sd[f"{p}.0.SelfAttention.o.weight"] = torch.randn(D, inner, generator=g) * inner ** -0.5
        sd[f"{p}.1.DenseReluDense.wi_0.weight"] = torch.randn(ff, D, generator=g) * D ** -0.5
        sd[f"{p}.1.DenseReluDense.wi_1.weight"] = torch.randn(ff, D, generator=g) * D ** -0.5
        sd[f"{p}.1.DenseReluDense.wo.weight"] = torch.randn(D, ff, generator=g) * ff ** -0.5
    return {k: v.to(torch.bfloat16).float() for k, v in sd.items()}


def _rel(a, b):
    return ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()


def test_encode_prompt_matches_oracle():
    from adv_grpo_amd import text_encoders as te
    from oracle import text_encoders as ote
    g = torch.Generator().manual_seed(5)
    sd_l, sd_g = _clip_sd(128, 256, 3, 64, 99, g), _clip_sd(192, 384, 4, 128, 99, g)
    sd_t = _t5_sd(384, 3, 320, 3, 120, g)
    ids = torch.randint(2, 97, (2, 77), generator=g); ids[:, 0] = 97; ids[0, 20:] = 98; ids[1, 60:] = 98
    ids_t5 = torch.randint(2, 120, (2, 128), generator=g)
    want_pe, want_pooled = ote.encode_prompt((sd_l, 3, 2, "quick_gelu", 98), (sd_g, 4, 3, "gelu", 98), (sd_t, 3, 3, 64),
                                             ids, ids, ids_t5)
    cl = te.CLIPTextEncoder(sd_l, 3, 2, "quick_gelu", 98)
    cg = te.CLIPTextEncoder(sd_g, 4, 3, "gelu", 98)
    t5 = te.T5Encoder(sd_t, 3, 3)
    pe, pooled = te.encode_prompt(cl, cg, t5, ids, ids, ids_t5)
    assert pe.shape == want_pe.shape == (2, 77 + 128, 384) and pooled.shape == want_pooled.shape
    # bf16 activations vs the fp32 oracle
    assert _rel(pe[:, :77, :128], want_pe[:, :77, :128]) < 2e-2          # CLIP-L penultimate
    assert _rel(pe[:, :77, 128:320], want_pe[:, :77, 128:320]) < 2e-2    # CLIP-G penultimate
    assert (pe[:, :77, 320:] == 0).all()                                 # zero padding up to the T5 width
    assert _rel(pe[:, 77:], want_pe[:, 77:]) < 2e-2                      # T5
    assert _rel(pooled, want_pooled) < 2e-2


def test_attention_bias_and_rmsnorm_rows_vs_torch():
    from adv_grpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    B, H, S = 2, 3, 100
    qkv = torch.randn(B, S, 3 * H * 64, device="cuda", generator=g).to(torch.bfloat16)
    bias = torch.randn(H, S, S, device="cuda", generator=g)
    q, k, v = (qkv[..., i * H * 64:(i + 1) * H * 64] for i in range(3))
    out = ops.attention_bias(q, k, v, H, bias, scale=0.2)
    qh, kh, vh = (t.float().view(B, S, H, 64).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.2 + bias[None], -1) @ vh).transpose(1, 2).reshape(B, S, H * 64)
    assert ((out.float() - ref).abs().max() / ref.abs().max()).item() < 2e-2
    x = torch.randn(77, 4096, device="cuda", generator=g).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(4096, device="cuda", generator=g)).to(torch.bfloat16)
    y = ops.rmsnorm_rows(x, w)
    xr = x.float()
    want = w * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16)
    assert torch.equal(y, want)


def test_encode_prompt_matches_reference_golden():
    """HIP path vs the committed outputs of the reference's encode_prompt (tests/golden/text_encoders.npz)."""
    import os
    import numpy as np
    from adv_grpo_amd import text_encoders as te
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "text_encoders.npz"))
    sd = {"l": {}, "g": {}, "t": {}}
    for k in g.files:
        if "/" in k:
            tag, name = k.split("/", 1)
            sd[tag][name] = torch.from_numpy(g[k]).view(torch.bfloat16).float()
    ids, ids_t5 = torch.from_numpy(g["ids"]), torch.from_numpy(g["ids_t5"])
    pe, pooled = te.encode_prompt(te.CLIPTextEncoder(sd["l"], 2, 1, "quick_gelu", 98), te.CLIPTextEncoder(sd["g"], 2, 2, "gelu", 98),
                                  te.T5Encoder(sd["t"], 2, 1), ids, ids, ids_t5)
    want_pe, want_pooled = torch.from_numpy(g["prompt_embeds"]), torch.from_numpy(g["pooled"])
    assert pe.shape == want_pe.shape and pooled.shape == want_pooled.shape
    assert _rel(pe[:, :77], want_pe[:, :77]) < 2e-2 and _rel(pe[:, 77:], want_pe[:, 77:]) < 2e-2 and _rel(pooled, want_pooled) < 2e-2
