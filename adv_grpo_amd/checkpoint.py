"""LoRA checkpoint I/O in the on-disk layout of the reference.

save_ckpt (scripts/train_sd3_fast_pickscore.py:389-398) calls PEFT's ``save_pretrained`` on the (EMA-swapped) LoRA
transformer into ``<save_dir>/checkpoints/checkpoint-<step>/lora``; ``config.train.lora_path`` feeds
``PeftModel.from_pretrained`` (TP:506-509).  PEFT (not installed here, pinned by setup.py) writes
  adapter_model.safetensors : "base_model.model.<module path>.lora_A.weight" [r, in], "...lora_B.weight" [out, r]
                              (the adapter name "default" is stripped on save)
  adapter_config.json       : LoraConfig fields (peft_type, r, lora_alpha, target_modules, init_lora_weights, ...).
PEFT cannot be imported here, so the layout is restated from its documented on-disk format and is NOT pinned against a
file written by PEFT itself ("format unpinned"; the round trip and the key / config contents are tested).
These functions are host-side format code: tensors in, files out (and back); they never touch the GPU."""
import json
import os

PREFIX = "base_model.model."
TARGET_MODULES = ["attn.add_k_proj", "attn.add_q_proj", "attn.add_v_proj", "attn.to_add_out", "attn.to_k",
                  "attn.to_out.0", "attn.to_q", "attn.to_v"]                       # TP:490-499


def adapter_config(r=32, lora_alpha=64, base_model="stabilityai/stable-diffusion-3.5-medium"):
    """LoraConfig(r=32, lora_alpha=64, init_lora_weights="gaussian", target_modules=...) (TP:500-505) as PEFT serialises it."""
    return {"peft_type": "LORA", "task_type": None, "base_model_name_or_path": base_model, "inference_mode": True,
            "r": r, "lora_alpha": lora_alpha, "lora_dropout": 0.0, "bias": "none", "fan_in_fan_out": False,
            "init_lora_weights": "gaussian", "target_modules": sorted(TARGET_MODULES), "modules_to_save": None,
            "use_rslora": False, "use_dora": False, "rank_pattern": {}, "alpha_pattern": {}}


def checkpoint_dir(save_dir, global_step):
    return os.path.join(save_dir, "checkpoints", f"checkpoint-{global_step}", "lora")   # TP:390-391


def save_lora(path, lora_state, r=32, lora_alpha=64, base_model="stabilityai/stable-diffusion-3.5-medium"):
    """lora_state: {"transformer_blocks.<i>.attn.<proj>.lora_A.weight": [r, in], "...lora_B.weight": [out, r]} (any
    float dtype, any device).  Writes the two PEFT files into `path`."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    tensors = {PREFIX + k: v.detach().to("cpu").contiguous() for k, v in lora_state.items()}
    save_file(tensors, os.path.join(path, "adapter_model.safetensors"), metadata={"format": "pt"})
    with open(os.path.join(path, "adapter_config.json"), "w") as f:
        json.dump(adapter_config(r, lora_alpha, base_model), f, indent=2, sort_keys=True)


def load_lora(path):
    """-> (lora_state with the PEFT prefix stripped, adapter_config dict).  Accepts the adapter name PEFT leaves in
    keys of some versions ("...lora_A.default.weight")."""
    from safetensors.torch import load_file
    raw = load_file(os.path.join(path, "adapter_model.safetensors"))
    with open(os.path.join(path, "adapter_config.json")) as f:
        cfg = json.load(f)
    out = {}
    for k, v in raw.items():
        k = k[len(PREFIX):] if k.startswith(PREFIX) else k
        k = k.replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B.")
        out[k] = v
    if cfg.get("peft_type", "LORA") != "LORA":
        raise ValueError(f"{path}: not a LoRA adapter ({cfg.get('peft_type')})")
    return out, cfg


# ---------------------------------------------------------------- resume state (what the reference does not save)
# save_ckpt upstream writes the LoRA only (TP:389-398): the DINO head (TD:592-603) is never written and optimizer / EMA /
# step state cannot be restored.  SURVEY 8(f2) asks for it: one extra safetensors file next to the PEFT ones,
# `trainer_state.safetensors` (+ `trainer_state.json` for the scalars), which PEFT ignores when it loads the adapter.
RESUME_FILE, RESUME_META = "trainer_state.safetensors", "trainer_state.json"


def save_resume_state(path, tensors, scalars):
    """tensors: {name: tensor} (flat f32 vectors: LoRA master weights, Adam moments, EMA, discriminator parameters and
    moments ...); scalars: JSON-serialisable dict (global_step, epoch, opt_step ...)."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    save_file({k: v.detach().to("cpu").contiguous() for k, v in tensors.items()}, os.path.join(path, RESUME_FILE),
              metadata={"format": "pt"})
    with open(os.path.join(path, RESUME_META), "w") as f:
        json.dump(scalars, f, indent=2, sort_keys=True)


def load_resume_state(path):
    """-> (tensors, scalars) or (None, None) when the checkpoint holds the adapter only (e.g. one written upstream)."""
    from safetensors.torch import load_file
    if not os.path.exists(os.path.join(path, RESUME_FILE)):
        return None, None
    with open(os.path.join(path, RESUME_META)) as f:
        scalars = json.load(f)
    return load_file(os.path.join(path, RESUME_FILE)), scalars
