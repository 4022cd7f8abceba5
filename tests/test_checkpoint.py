"""LoRA checkpoint layout: PEFT file names, key names, shapes; save -> load round trip.  Host code only."""
import json
import os

import torch

from adv_grpo_amd import checkpoint


def _state(layers=3, D=64, r=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    st = {}
    for i in range(layers):
        names = ["to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj"]
        if i != layers - 1:
            names.append("to_add_out")                       # the last block is context_pre_only: no to_add_out
        for n in names:
            st[f"transformer_blocks.{i}.attn.{n}.lora_A.weight"] = torch.randn(r, D, generator=g)
            st[f"transformer_blocks.{i}.attn.{n}.lora_B.weight"] = torch.randn(D, r, generator=g)
    return st


def test_peft_layout_and_round_trip(tmp_path):
    st = _state()
    path = checkpoint.checkpoint_dir(str(tmp_path), 120)
    assert path.endswith(os.path.join("checkpoints", "checkpoint-120", "lora"))          # TP:390-391
    checkpoint.save_lora(path, st)
    assert sorted(os.listdir(path)) == ["adapter_config.json", "adapter_model.safetensors"]
    cfg = json.load(open(os.path.join(path, "adapter_config.json")))
    assert cfg["peft_type"] == "LORA" and cfg["r"] == 32 and cfg["lora_alpha"] == 64      # TP:500-505
    assert cfg["init_lora_weights"] == "gaussian"
    assert set(cfg["target_modules"]) == {"attn.add_k_proj", "attn.add_q_proj", "attn.add_v_proj", "attn.to_add_out",
                                          "attn.to_k", "attn.to_out.0", "attn.to_q", "attn.to_v"}   # TP:490-499
    from safetensors.torch import load_file
    raw = load_file(os.path.join(path, "adapter_model.safetensors"))
    assert "base_model.model.transformer_blocks.0.attn.to_q.lora_A.weight" in raw
    assert "base_model.model.transformer_blocks.1.attn.to_out.0.lora_B.weight" in raw
    assert all(k.startswith("base_model.model.") for k in raw)
    back, cfg2 = checkpoint.load_lora(path)
    assert back.keys() == st.keys() and cfg2 == cfg
    for k in st:
        assert torch.equal(back[k], st[k])


def test_load_accepts_named_adapter_keys(tmp_path):
    from safetensors.torch import save_file
    st = _state(layers=1)
    named = {"base_model.model." + k.replace(".lora_A.", ".lora_A.default.").replace(".lora_B.", ".lora_B.default."): v
             for k, v in st.items()}
    os.makedirs(tmp_path / "lora")
    save_file(named, str(tmp_path / "lora" / "adapter_model.safetensors"))
    json.dump(checkpoint.adapter_config(), open(tmp_path / "lora" / "adapter_config.json", "w"))
    back, _ = checkpoint.load_lora(str(tmp_path / "lora"))
    assert back.keys() == st.keys()
