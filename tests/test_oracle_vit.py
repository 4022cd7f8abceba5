"""CPU: pin the ViT oracle (oracle/vit.py) against the installed transformers CLIPModel / Dinov2Model with
seeded random weights (the only third-party models on the path that are importable)."""
import pytest
import torch

transformers = pytest.importorskip("transformers")
from oracle import vit  # noqa: E402


def test_clip_oracle_matches_transformers():
    from transformers import CLIPConfig, CLIPModel
    cfg = vit.ClipConfig(v_hidden=160, v_layers=2, v_heads=2, v_mlp=320, image_size=56, patch=14, t_hidden=128,
                         t_layers=2, t_heads=2, t_mlp=256, vocab=1000, max_pos=77, proj=96, eos_token_id=999)
    hf = CLIPConfig(
        text_config=dict(hidden_size=cfg.t_hidden, intermediate_size=cfg.t_mlp, num_hidden_layers=cfg.t_layers,
                         num_attention_heads=cfg.t_heads, vocab_size=cfg.vocab, max_position_embeddings=77,
                         hidden_act="gelu", eos_token_id=cfg.eos_token_id, bos_token_id=998, pad_token_id=1),
        vision_config=dict(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_mlp, num_hidden_layers=cfg.v_layers,
                           num_attention_heads=cfg.v_heads, image_size=cfg.image_size, patch_size=14,
                           hidden_act="gelu"),
        projection_dim=cfg.proj)
    torch.manual_seed(0)
    m = CLIPModel(hf).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    W = {k: v.detach() for k, v in m.state_dict().items()}
    px = torch.randn(3, 3, 56, 56)
    ids = torch.randint(1, 990, (3, 77)); ids[0, 10] = 999; ids[1, 40] = 999; ids[2, 76] = 999; ids[1, 60] = 999
    with torch.no_grad():
        out_i = m.get_image_features(pixel_values=px)
        out_t = m.get_text_features(input_ids=ids)
        out_i = getattr(out_i, "pooler_output", out_i)   # transformers >= 5 returns an output object
        out_t = getattr(out_t, "pooler_output", out_t)
        assert torch.allclose(vit.clip_image_features(W, cfg, px), out_i, atol=2e-5, rtol=1e-4)
        assert torch.allclose(vit.clip_text_features(W, cfg, ids), out_t, atol=2e-5, rtol=1e-4)


def test_clip_oracle_legacy_eos_token_id_2_matches_transformers():
    """The released PickScore_v1 / laion CLIP-H config.json carries text_config.eos_token_id = 2: transformers then pools at argmax(input_ids)
    (the eos of the original CLIP vocabulary is its largest id), not at the first occurrence of token 2.  Pinned against the installed model."""
    from transformers import CLIPConfig, CLIPModel
    cfg = vit.ClipConfig(v_hidden=160, v_layers=1, v_heads=2, v_mlp=320, image_size=56, patch=14, t_hidden=128,
                         t_layers=2, t_heads=2, t_mlp=256, vocab=1000, max_pos=77, proj=96, eos_token_id=2)
    hf = CLIPConfig(
        text_config=dict(hidden_size=cfg.t_hidden, intermediate_size=cfg.t_mlp, num_hidden_layers=cfg.t_layers,
                         num_attention_heads=cfg.t_heads, vocab_size=cfg.vocab, max_position_embeddings=77,
                         hidden_act="gelu", eos_token_id=2, bos_token_id=0, pad_token_id=1),
        vision_config=dict(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_mlp, num_hidden_layers=cfg.v_layers,
                           num_attention_heads=cfg.v_heads, image_size=cfg.image_size, patch_size=14, hidden_act="gelu"),
        projection_dim=cfg.proj)
    torch.manual_seed(3)
    m = CLIPModel(hf).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    W = {k: v.detach() for k, v in m.state_dict().items()}
    ids = torch.randint(3, 990, (3, 77)); ids[0, 10] = 999; ids[1, 40] = 999; ids[2, 76] = 999
    ids[:, 5] = 2                                      # a token with id 2 in front of the real end: must NOT be the pooled position
    with torch.no_grad():
        out_t = m.get_text_features(input_ids=ids)
        out_t = getattr(out_t, "pooler_output", out_t)
        got = vit.clip_text_features(W, cfg, ids)
    assert torch.allclose(got, out_t, atol=2e-5, rtol=1e-4)
    import dataclasses
    wrong = vit.clip_text_features(W, dataclasses.replace(cfg, eos_token_id=999), ids[:, :])   # sanity: position 10/40/76 == argmax here
    assert torch.allclose(wrong, out_t, atol=2e-5, rtol=1e-4)


def test_dino_oracle_matches_transformers_dinov2():
    from transformers import Dinov2Config, Dinov2Model
    cfg = vit.DinoConfig(hidden=128, layers=2, heads=2, mlp=512, image_size=70, patch=14)
    hf = Dinov2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, mlp_ratio=4, image_size=70,
                      patch_size=14, layerscale_value=0.5)
    torch.manual_seed(1)
    m = Dinov2Model(hf).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    W = vit.dino_from_hf({k: v.detach() for k, v in m.state_dict().items()}, cfg)
    x = torch.randn(2, 3, 70, 70)
    with torch.no_grad():
        ref = m(pixel_values=x).last_hidden_state
        got = vit.dino_forward_features(W, cfg, x)
    assert got.shape == (2, 26, 128)
    assert torch.allclose(got, ref, atol=3e-5, rtol=1e-4)
