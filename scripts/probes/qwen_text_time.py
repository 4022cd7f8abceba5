"""Prompt encoding time of config 5's text tower at full size (Qwen2.5-VL-7B language model: 28 layers, 3584 wide) for one prompt group."""
import sys, time
import torch
sys.path.insert(0, ".")
from adv_grpo_amd import synthetic
from adv_grpo_amd.model_configs import QwenTextConfig
from adv_grpo_amd.qwen_text_encoder import Qwen25VLTextEncoder
cfg = QwenTextConfig()
with synthetic.on_device("cuda"):
    enc = Qwen25VLTextEncoder(synthetic.qwen_text_weights(cfg, 1357, dtype=torch.bfloat16), cfg, "cuda")
print(f"weights: {torch.cuda.memory_allocated() / 2**30:.1f} GiB")
for B, T in ((2, 34 + 40), (2, 34 + 128)):
    ids = torch.randint(0, cfg.vocab_size, (B, T), device="cuda")
    mask = torch.ones(B, T, dtype=torch.long)
    enc.encode_prompt(ids, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        emb, _m = enc.encode_prompt(ids, mask)
    torch.cuda.synchronize()
    print(f"prompt + negative prompt, {T} tokens each: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms -> embeds {tuple(emb.shape)}, finite {bool(torch.isfinite(emb.float()).all())}")
