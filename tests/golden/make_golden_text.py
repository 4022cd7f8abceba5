#!/usr/bin/env python3
"""Golden fixture for the prompt-encoding path, made by RUNNING THE REFERENCE'S encode_prompt
(adv_grpo/diffusers_patch/train_dreambooth_lora_sd3.py:98-144, imported from /root/reference) on small, seeded
transformers modules (CLIPTextModelWithProjection x2, T5EncoderModel v1.1).  Stored: the modules' weights (bf16-representable
fp32), the token ids and the reference's outputs.  Runs only in the build container.  Usage: python tests/golden/make_golden_text.py"""
import os
import sys

import numpy as np
import torch
from transformers import CLIPTextConfig, CLIPTextModelWithProjection, T5Config, T5EncoderModel

sys.path.insert(0, "/root/reference")
from adv_grpo.diffusers_patch.train_dreambooth_lora_sd3 import encode_prompt  # noqa: E402

torch.manual_seed(20240917)
cl = CLIPTextModelWithProjection(CLIPTextConfig(vocab_size=99, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                                num_attention_heads=1, max_position_embeddings=77, projection_dim=64,
                                                hidden_act="quick_gelu", eos_token_id=98, bos_token_id=97, pad_token_id=1)).eval()
cg = CLIPTextModelWithProjection(CLIPTextConfig(vocab_size=99, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                                num_attention_heads=2, max_position_embeddings=77, projection_dim=64,
                                                hidden_act="gelu", eos_token_id=98, bos_token_id=97, pad_token_id=1)).eval()
t5 = T5EncoderModel(T5Config(vocab_size=120, d_model=192, d_kv=64, d_ff=128, num_layers=2, num_heads=1,
                             feed_forward_proj="gated-gelu", relative_attention_num_buckets=32,
                             relative_attention_max_distance=128)).eval()
for m in (cl, cg, t5):                              # weights every implementation can hold exactly in bf16
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.to(torch.bfloat16).float())
g = torch.Generator().manual_seed(5)
ids = torch.randint(2, 97, (2, 77), generator=g); ids[:, 0] = 97; ids[0, 20:] = 98; ids[1, 60:] = 98
ids_t5 = torch.randint(2, 120, (2, 128), generator=g)
with torch.no_grad():
    pe, pooled = encode_prompt([cl, cg, t5], [None, None, None], ["a", "b"], 128, device="cpu",
                               text_input_ids_list=[ids, ids, ids_t5])
out = {"ids": ids.numpy(), "ids_t5": ids_t5.numpy(), "prompt_embeds": pe.numpy(), "pooled": pooled.numpy()}
for tag, m in (("l", cl), ("g", cg), ("t", t5)):
    for k, v in m.state_dict().items():
        out[f"{tag}/{k}"] = v.to(torch.bfloat16).view(torch.int16).numpy()        # bf16 bit patterns: half the bytes
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "text_encoders.npz"), **out)
print("wrote text_encoders.npz", pe.shape, pooled.shape)
