"""Pins oracle/text_encoders.py: against transformers' CLIPTextModelWithProjection / T5EncoderModel built
from small random configs, and against the REFERENCE's own encode_prompt run on those modules (imported from
/root/reference when it is present -- it never is on the GPU box, where only the transformers pin runs)."""
import os
import sys

import pytest
import torch

transformers = pytest.importorskip("transformers")


def _models(seed=0):
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection, T5Config, T5EncoderModel
    torch.manual_seed(seed)
    cl = CLIPTextModelWithProjection(CLIPTextConfig(vocab_size=99, hidden_size=64, intermediate_size=128, num_hidden_layers=3,
                                                    num_attention_heads=2, max_position_embeddings=77, projection_dim=48,
                                                    hidden_act="quick_gelu", eos_token_id=98, bos_token_id=97, pad_token_id=1)).eval()
    cg = CLIPTextModelWithProjection(CLIPTextConfig(vocab_size=99, hidden_size=128, intermediate_size=256, num_hidden_layers=4,
                                                    num_attention_heads=2, max_position_embeddings=77, projection_dim=80,
                                                    hidden_act="gelu", eos_token_id=98, bos_token_id=97, pad_token_id=1)).eval()
    t5 = T5EncoderModel(T5Config(vocab_size=120, d_model=256, d_kv=64, d_ff=320, num_layers=3, num_heads=3,
                                 feed_forward_proj="gated-gelu", relative_attention_num_buckets=32,
                                 relative_attention_max_distance=128)).eval()
    g = torch.Generator().manual_seed(seed + 1)
    ids = torch.randint(2, 97, (2, 77), generator=g)
    ids[:, 0] = 97
    ids[0, 20:] = 98
    ids[1, 55:] = 98                                   # EOS padding, as the CLIP tokenizers of SD3 pad
    ids_t5 = torch.randint(2, 120, (2, 40), generator=g)
    return cl, cg, t5, ids, ids_t5


def _specs(cl, cg, t5):
    from oracle import text_encoders as te  # noqa: F401
    sd = lambda m: {k: v.float() for k, v in m.state_dict().items()}
    return ((sd(cl), 3, 2, "quick_gelu", 98), (sd(cg), 4, 2, "gelu", 98), (sd(t5), 3, 3, 64))


def test_oracle_matches_transformers_modules():
    from oracle import text_encoders as te
    cl, cg, t5, ids, ids_t5 = _models()
    L, G, T = _specs(cl, cg, t5)
    with torch.no_grad():
        for m, spec in ((cl, L), (cg, G)):
            out = m(ids, output_hidden_states=True)
            pen, pooled = te.clip_text_hidden_and_pooled(*spec, ids)
            assert torch.allclose(pen, out.hidden_states[-2], atol=2e-5, rtol=1e-4)
            assert torch.allclose(pooled, out[0], atol=2e-5, rtol=1e-4)
        ref = t5(ids_t5)[0]
        got = te.t5_encoder(*T, ids_t5)
        assert torch.allclose(got, ref, atol=2e-5, rtol=1e-4)


@pytest.mark.skipif(not os.path.isdir("/root/reference/adv_grpo"), reason="reference tree not present (GPU box)")
def test_oracle_matches_reference_encode_prompt():
    sys.path.insert(0, "/root/reference")
    try:
        from adv_grpo.diffusers_patch.train_dreambooth_lora_sd3 import encode_prompt as ref_encode
    finally:
        sys.path.pop(0)
    from oracle import text_encoders as te
    cl, cg, t5, ids, ids_t5 = _models(3)
    L, G, T = _specs(cl, cg, t5)
    with torch.no_grad():
        pe_ref, pooled_ref = ref_encode([cl, cg, t5], [None, None, None], ["a", "b"], 40, device="cpu",
                                        text_input_ids_list=[ids, ids, ids_t5])
        pe, pooled = te.encode_prompt(L, G, T, ids, ids, ids_t5)
    assert pe.shape == pe_ref.shape == (2, 77 + 40, 256) and pooled.shape == pooled_ref.shape == (2, 48 + 80)
    assert torch.allclose(pe, pe_ref, atol=2e-5, rtol=1e-4) and torch.allclose(pooled, pooled_ref, atol=2e-5, rtol=1e-4)


def test_clip_legacy_eos_rule_matches_transformers():
    """The released CLIP-L / CLIP-G configs of SD3 carry eos_token_id = 2, for which transformers pools at argmax(ids)
    (the highest id = the real EOS token 49407) instead of the first occurrence of eos_token_id."""
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    from oracle import text_encoders as te
    torch.manual_seed(7)
    m = CLIPTextModelWithProjection(CLIPTextConfig(vocab_size=99, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                                   num_attention_heads=2, max_position_embeddings=77, projection_dim=32,
                                                   hidden_act="quick_gelu", eos_token_id=2, bos_token_id=0, pad_token_id=1)).eval()
    ids = torch.randint(3, 97, (2, 77)); ids[0, 9] = 98; ids[0, 10:] = 98; ids[1, 40:] = 98     # 98 plays the role of 49407
    with torch.no_grad():
        out = m(ids, output_hidden_states=True)
        pen, pooled = te.clip_text_hidden_and_pooled({k: v.float() for k, v in m.state_dict().items()}, 2, 2, "quick_gelu", 2, ids)
    assert torch.allclose(pen, out.hidden_states[-2], atol=2e-5, rtol=1e-4) and torch.allclose(pooled, out[0], atol=2e-5, rtol=1e-4)


def _load_text_golden():
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "text_encoders.npz"))
    sd = {"l": {}, "g": {}, "t": {}}
    for k in g.files:
        if "/" in k:
            tag, name = k.split("/", 1)
            sd[tag][name] = torch.from_numpy(g[k]).view(torch.bfloat16).float()
    return g, sd


def test_oracle_matches_committed_reference_golden():
    """tests/golden/text_encoders.npz = outputs of the reference's encode_prompt (made by make_golden_text.py)."""
    from oracle import text_encoders as te
    g, sd = _load_text_golden()
    ids, ids_t5 = torch.from_numpy(g["ids"]), torch.from_numpy(g["ids_t5"])
    pe, pooled = te.encode_prompt((sd["l"], 2, 1, "quick_gelu", 98), (sd["g"], 2, 2, "gelu", 98), (sd["t"], 2, 1, 64), ids, ids, ids_t5)
    assert torch.allclose(pe, torch.from_numpy(g["prompt_embeds"]), atol=2e-5, rtol=1e-4)
    assert torch.allclose(pooled, torch.from_numpy(g["pooled"]), atol=2e-5, rtol=1e-4)
