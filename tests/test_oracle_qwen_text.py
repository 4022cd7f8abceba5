"""oracle/qwen_text.py (the Qwen2.5-VL text tower of config 5's prompt encoding) PINNED against the installed transformers'
Qwen2_5_VLTextModel on seeded random weights; the template-drop / padding step against its definition."""
import torch

from oracle import qwen_text as o


def _hf(cfg, seed):
    from transformers.models.qwen2_5_vl import configuration_qwen2_5_vl as C, modeling_qwen2_5_vl as M
    hd = cfg.head_dim
    tc = C.Qwen2_5_VLTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
                                max_position_embeddings=512, rms_norm_eps=cfg.rms_eps, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                                rope_scaling={"type": "mrope", "mrope_section": [hd // 8, 3 * hd // 16, 3 * hd // 16], "rope_theta": cfg.rope_theta})
    torch.manual_seed(seed)
    m = M.Qwen2_5_VLTextModel(tc).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.1)
            elif "norm" in n:
                p.copy_(1 + 0.1 * torch.randn_like(p))
    return m


def test_text_model_matches_transformers_qwen2_5_vl():
    cfg = o.QwenTextConfig(vocab_size=300, hidden_size=256, intermediate_size=512, num_layers=3, num_heads=4, num_kv_heads=2, rms_eps=1e-6,
                           rope_theta=1e6)
    m = _hf(cfg, 0)
    W = {k: v.detach() for k, v in m.state_dict().items()}
    ids = torch.randint(3, 300, (3, 23), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = m(input_ids=ids, output_hidden_states=True).hidden_states[-1]
        out = o.text_model_forward(W, cfg, ids)
    assert (out - ref).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-5
    # right-padded batch: the valid positions of a shorter sample are unchanged by the padding behind them
    mask = torch.ones(3, 23, dtype=torch.long)
    mask[1, 15:] = 0
    with torch.no_grad():
        ref_m = m(input_ids=ids, attention_mask=mask, output_hidden_states=True).hidden_states[-1]
        out_m = o.text_model_forward(W, cfg, ids, mask)
    assert (out_m[1, :15] - ref_m[1, :15]).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-5
    assert (out_m[1, :15] - out[1, :15]).abs().max().item() < 1e-5


def test_prompt_embeds_drop_the_template_and_right_pad():
    cfg = o.QwenTextConfig(vocab_size=100, hidden_size=64, intermediate_size=128, num_layers=1, num_heads=2, num_kv_heads=1)
    g = torch.Generator().manual_seed(2)
    W = {"embed_tokens.weight": torch.randn(100, 64, generator=g), "norm.weight": torch.ones(64)}
    p = "layers.0"
    for n, (a, b) in {"q_proj": (64, 64), "k_proj": (32, 64), "v_proj": (32, 64)}.items():
        W[f"{p}.self_attn.{n}.weight"], W[f"{p}.self_attn.{n}.bias"] = torch.randn(a, b, generator=g) / 8, torch.zeros(a)
    W[f"{p}.self_attn.o_proj.weight"] = torch.randn(64, 64, generator=g) / 8
    W[f"{p}.mlp.gate_proj.weight"], W[f"{p}.mlp.up_proj.weight"] = torch.randn(128, 64, generator=g) / 8, torch.randn(128, 64, generator=g) / 8
    W[f"{p}.mlp.down_proj.weight"] = torch.randn(64, 128, generator=g) / 11
    W[f"{p}.input_layernorm.weight"] = W[f"{p}.post_attention_layernorm.weight"] = torch.ones(64)
    ids = torch.randint(0, 100, (2, 50), generator=g)
    mask = torch.ones(2, 50, dtype=torch.long)
    mask[0, 44:] = 0
    emb, msk = o.qwen_prompt_embeds(W, cfg, ids, mask)
    assert emb.shape == (2, 16, 64) and msk.tolist() == [[1] * 10 + [0] * 6, [1] * 16]
    hs = o.text_model_forward(W, cfg, ids, mask)
    assert torch.equal(emb[0, :10], hs[0, 34:44]) and torch.equal(emb[1], hs[1, 34:]) and (emb[0, 10:] == 0).all()
