"""Real checkpoints: read a Hugging Face snapshot directory into the state dicts + configs the model classes take.

What the reference does first is ``StableDiffusion3Pipeline.from_pretrained(config.pretrained.model)`` (scripts/train_sd3_fast_pickscore.py:
447-449), ``PickScoreScorer`` loading ``yuvalkirstain/PickScore_v1`` (adv_grpo/pickscore_scorer.py:8-14) and
``timm.create_model("vit_base_patch14_dinov2.lvd142m", pretrained=True)`` (scripts/train_sd3_fast_dino_patch.py:589).  There is no network here,
so nothing is downloaded: these functions take the LOCAL directory such a download leaves behind (the ``snapshots/<rev>/`` directory of the HF
cache, or a ``save_pretrained`` directory) and

  * read ``config.json`` of every component and turn it into the dataclasses of model_configs.py -- refusing a config whose class name or any
    architecture field this build does not implement (a wrong ``num_layers`` / ``dual_attention_layers`` / ``qk_norm`` is an error, not a
    silently different model);
  * check EVERY tensor name and shape of the safetensors files (single file or ``*.index.json`` shards) against the architecture's layout
    (``synthetic.shapes(<builder>, cfg)``: the same code that lays out the synthetic weights) from the file headers, before a byte of tensor
    data is read: missing keys, unexpected keys and shape mismatches are listed in the error;
  * return ``(state_dict, config)`` with diffusers / transformers / timm key names, which is what ``SD3Transformer2DModel`` (mmdit.py),
    ``AutoencoderKLDecoder`` (vae.py), ``CLIPTextEncoder`` / ``T5Encoder`` (text_encoders.py), ``PickScoreScorer`` (pickscore_scorer.py),
    ``DinoV2`` (vit.py) and the Qwen-Image classes are built from.

Layout read (diffusers ``model_index.json`` pipelines):  ``transformer/`` ``vae/`` ``text_encoder/`` ``text_encoder_2/`` ``text_encoder_3/``
each with ``config.json`` + ``diffusion_pytorch_model[.fp16].safetensors`` | ``model[.fp16].safetensors`` | ``<stem>.safetensors.index.json`` +
shards.  Scorers: a ``CLIPModel`` directory (``config.json`` + ``model.safetensors``) and a timm directory (``config.json`` with
``architecture`` + ``model.safetensors``).  ``pytorch_model.bin`` pickles are read with ``torch.load(weights_only=True)`` when no safetensors
file exists.  Tested on synthetic snapshots written in exactly this layout (tests/test_hub.py); no released checkpoint exists in this image.
"""
import glob
import json
import os

import torch

from . import synthetic
from .model_configs import (ClipConfig, ClipTextConfig, DinoConfig, MMDiTConfig, QwenMMDiTConfig, QwenTextConfig, QwenVaeConfig, T5Config,
                            VaeConfig)


class HubError(RuntimeError):
    pass


# ---------------------------------------------------------------------------------------------------------------- files
_STEMS = ("diffusion_pytorch_model", "model")


def read_config(component_dir):
    path = os.path.join(component_dir, "config.json")
    if not os.path.isfile(path):
        raise HubError(f"{component_dir}: no config.json (expected a Hugging Face snapshot component directory)")
    with open(path) as f:
        return json.load(f)


def _weight_files(component_dir):
    """-> list of weight files of the component (safetensors shards by their index, a single safetensors file, or .bin pickles)."""
    for stem in _STEMS:
        for variant in ("", ".fp16", ".bf16"):
            idx = os.path.join(component_dir, f"{stem}.safetensors.index{variant}.json")
            idx2 = os.path.join(component_dir, f"{stem}{variant}.safetensors.index.json")
            for i in (idx, idx2):
                if os.path.isfile(i):
                    with open(i) as f:
                        wm = json.load(f)["weight_map"]
                    files = sorted(set(wm.values()))
                    missing = [x for x in files if not os.path.isfile(os.path.join(component_dir, x))]
                    if missing:
                        raise HubError(f"{i} lists shards that are not there: {missing[:4]}")
                    return [os.path.join(component_dir, x) for x in files], wm
            one = os.path.join(component_dir, f"{stem}{variant}.safetensors")
            if os.path.isfile(one):
                return [one], None
    bins = sorted(glob.glob(os.path.join(component_dir, "pytorch_model*.bin")))
    if bins:
        return bins, None
    raise HubError(f"{component_dir}: no weights found (looked for {{{', '.join(_STEMS)}}}[.fp16].safetensors, their .index.json shards and "
                   "pytorch_model*.bin)")


def header_shapes(component_dir):
    """{tensor name: shape} from the files' headers (safetensors: no tensor data is read)."""
    files, wm = _weight_files(component_dir)
    out = {}
    for f in files:
        if f.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(f, framework="pt") as sf:
                for k in sf.keys():
                    if k in out:
                        raise HubError(f"{component_dir}: tensor {k!r} appears in more than one shard")
                    out[k] = tuple(sf.get_slice(k).get_shape())
        else:
            for k, v in torch.load(f, map_location="cpu", weights_only=True).items():
                out[k] = tuple(v.shape)
    if wm is not None:
        absent = sorted(set(wm) - set(out))
        if absent:
            raise HubError(f"{component_dir}: the shard index names tensors no shard holds: {absent[:4]}")
    return out


def read_tensors(component_dir, keys=None, rename=None):
    """Load the component's tensors (CPU, checkpoint dtype).  keys: only these (after renaming); rename: callable old name -> new name / None."""
    files, _ = _weight_files(component_dir)
    out = {}
    for f in files:
        if f.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(f, framework="pt") as sf:
                for k in sf.keys():
                    n = rename(k) if rename else k
                    if n is not None and (keys is None or n in keys):
                        out[n] = sf.get_tensor(k)
        else:
            for k, v in torch.load(f, map_location="cpu", weights_only=True).items():
                n = rename(k) if rename else k
                if n is not None and (keys is None or n in keys):
                    out[n] = v
    return out


def validate(have, want, what, optional=()):
    """have / want: {name: shape}.  Raises HubError naming what is missing, what is not expected and what has the wrong shape."""
    missing = sorted(k for k in want if k not in have)
    extra = sorted(k for k in have if k not in want and k not in optional)
    wrong = sorted((k, have[k], want[k]) for k in want if k in have and tuple(have[k]) != tuple(want[k]))
    if missing or extra or wrong:
        lines = [f"{what}: the checkpoint does not have this architecture's state-dict layout"]
        if missing:
            lines.append(f"  {len(missing)} missing, e.g. {missing[:5]}")
        if extra:
            lines.append(f"  {len(extra)} unexpected, e.g. {extra[:5]}")
        if wrong:
            lines.append(f"  {len(wrong)} with another shape, e.g. " + "; ".join(f"{k}: {h} != {w}" for k, h, w in wrong[:5]))
        raise HubError("\n".join(lines))


def _expect(cfg, what, _defaults=None, **fields):
    """Every named field of a config.json must have exactly this value (None in the file counts as absent for False/None).  A field that is
    ABSENT from the file reads as the owning library's constructor default (`_defaults`: configs saved before a field existed omit it, and
    diffusers / transformers then build the model with that default); without a listed default an absent field is refused."""
    for k, v in fields.items():
        got = cfg.get(k, None)
        if k not in cfg and _defaults and k in _defaults:
            got = _defaults[k]
        if isinstance(v, (list, tuple)):
            ok = got is not None and list(got) == list(v)
        elif isinstance(v, float):
            ok = got is not None and abs(float(got) - v) <= 1e-9 * max(1.0, abs(v))
        else:
            ok = got == v or (v in (False, None) and got in (False, None))
        if not ok:
            raise HubError(f"{what}: config.json has {k} = {got!r}; this build implements {k} = {v!r}")


def _class(cfg, what, *names):
    got = cfg.get("_class_name") or (cfg.get("architectures") or [None])[0]
    if got not in names:
        raise HubError(f"{what}: config.json describes a {got!r}, expected one of {names}")


# ---------------------------------------------------------------------------------------------------------------- SD3 / SD3.5
def sd3_transformer_config(cfg, what="transformer"):
    """diffusers SD3Transformer2DModel config.json -> MMDiTConfig (SD3-medium, SD3.5-medium "MMDiT-X", SD3.5-large)."""
    _class(cfg, what, "SD3Transformer2DModel")
    heads, hd = int(cfg["num_attention_heads"]), int(cfg["attention_head_dim"])
    qk = cfg.get("qk_norm")
    if qk not in (None, "rms_norm"):
        raise HubError(f"{what}: qk_norm = {qk!r} is not implemented (None or 'rms_norm')")
    if hd != 64:
        raise HubError(f"{what}: attention_head_dim = {hd}; the MMDiT attention kernels of this path serve head dim 64")
    _expect(cfg, what, caption_projection_dim=heads * hd)
    out = MMDiTConfig(num_layers=int(cfg["num_layers"]), num_heads=heads, head_dim=hd, in_channels=int(cfg["in_channels"]),
                      out_channels=int(cfg.get("out_channels") or cfg["in_channels"]), patch_size=int(cfg["patch_size"]),
                      joint_attention_dim=int(cfg["joint_attention_dim"]), pooled_projection_dim=int(cfg["pooled_projection_dim"]),
                      pos_embed_max_size=int(cfg["pos_embed_max_size"]),
                      dual_attention_layers=tuple(int(i) for i in (cfg.get("dual_attention_layers") or ())), qk_norm=qk == "rms_norm")
    if any(i < 0 or i >= out.num_layers for i in out.dual_attention_layers):
        raise HubError(f"{what}: dual_attention_layers {out.dual_attention_layers} outside 0..{out.num_layers - 1}")
    return out


def _mmdit_shapes(cfg):
    want = synthetic.shapes(synthetic.mmdit_weights, cfg)
    if not cfg.qk_norm:
        want = {k: v for k, v in want.items() if ".norm_q." not in k and ".norm_k." not in k and ".norm_added_" not in k}
    return want


def load_sd3_transformer(component_dir):
    """-> (state dict with diffusers names, MMDiTConfig)."""
    cfg = sd3_transformer_config(read_config(component_dir), component_dir)
    want = _mmdit_shapes(cfg)
    validate(header_shapes(component_dir), want, component_dir)
    return read_tensors(component_dir, keys=set(want)), cfg


_VAE_LEGACY = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def _vae_rename(k):
    """decoder.* (+ the pre-0.18 attention names diffusers still converts on load); encoder / quant convs are not on the path."""
    if not k.startswith("decoder."):
        return None
    parts = k.split(".")
    if "attentions" in parts and parts[-2] in _VAE_LEGACY:
        parts[-2:-1] = _VAE_LEGACY[parts[-2]].split(".")
        return ".".join(parts)
    return k


def _vae_shape(name, shape):
    """the legacy checkpoints hold the mid-block attention projections as 1 x 1 convolutions [C, C, 1, 1]"""
    if ".attentions." in name and name.endswith(".weight") and len(shape) == 4 and tuple(shape[2:]) == (1, 1):
        return tuple(shape[:2])
    return tuple(shape)


def load_vae_decoder(component_dir):
    """AutoencoderKL of SD3 / SD3.5 -> (decoder.* state dict upcast to float32 as ``vae.to(torch.float32)`` does (TP:481), VaeConfig)."""
    cfg = read_config(component_dir)
    _class(cfg, component_dir, "AutoencoderKL")
    # (absent = diffusers' AutoencoderKL.__init__ defaults: act_fn "silu", use_post_quant_conv True, mid_block_add_attention True)
    _expect(cfg, component_dir, _defaults=dict(act_fn="silu", use_post_quant_conv=True, mid_block_add_attention=True),
            act_fn="silu", use_post_quant_conv=False, mid_block_add_attention=True)
    out = VaeConfig(latent_channels=int(cfg["latent_channels"]), block_out_channels=tuple(cfg["block_out_channels"]),
                    layers_per_block=int(cfg["layers_per_block"]), norm_num_groups=int(cfg["norm_num_groups"]),
                    scaling_factor=float(cfg["scaling_factor"]), shift_factor=float(cfg.get("shift_factor") or 0.0))
    want = synthetic.shapes(synthetic.vae_decoder_weights, out)
    have = {n: _vae_shape(n, s) for n, s in ((_vae_rename(k), s) for k, s in header_shapes(component_dir).items()) if n is not None}
    validate(have, want, component_dir)
    sd = read_tensors(component_dir, keys=set(want), rename=_vae_rename)
    return {k: v.reshape(want[k]).float() for k, v in sd.items()}, out


def load_clip_text(component_dir):
    """CLIPTextModelWithProjection (text_encoder / text_encoder_2) -> (state dict, ClipTextConfig)."""
    cfg = read_config(component_dir)
    _class(cfg, component_dir, "CLIPTextModelWithProjection")
    act = cfg.get("hidden_act", "quick_gelu")
    if act not in ("quick_gelu", "gelu"):
        raise HubError(f"{component_dir}: hidden_act = {act!r} is not implemented")
    out = ClipTextConfig(hidden=int(cfg["hidden_size"]), layers=int(cfg["num_hidden_layers"]), heads=int(cfg["num_attention_heads"]),
                         mlp=int(cfg["intermediate_size"]), proj=int(cfg["projection_dim"]), vocab=int(cfg["vocab_size"]),
                         max_pos=int(cfg["max_position_embeddings"]), act=act, eos_token_id=int(cfg.get("eos_token_id", 2)))
    if out.hidden // out.heads != 64:
        raise HubError(f"{component_dir}: head dim {out.hidden // out.heads}; the text towers run on the head-dim-64 attention kernels")
    want = synthetic.shapes(synthetic.clip_text_weights, out)
    optional = {"text_model.embeddings.position_ids"}
    validate(header_shapes(component_dir), want, component_dir, optional=optional)
    return read_tensors(component_dir, keys=set(want)), out


def load_t5_encoder(component_dir):
    """T5EncoderModel (text_encoder_3) -> (state dict, T5Config)."""
    cfg = read_config(component_dir)
    _class(cfg, component_dir, "T5EncoderModel")
    _expect(cfg, component_dir, _defaults=dict(d_kv=64), d_kv=64)                 # (transformers' T5Config default)
    ff = cfg.get("feed_forward_proj", "gated-gelu")
    if ff != "gated-gelu":
        raise HubError(f"{component_dir}: feed_forward_proj = {ff!r}; only T5 v1.1's gated-gelu is implemented")
    out = T5Config(d_model=int(cfg["d_model"]), layers=int(cfg["num_layers"]), heads=int(cfg["num_heads"]), d_kv=int(cfg.get("d_kv", 64)),
                   d_ff=int(cfg["d_ff"]), vocab=int(cfg["vocab_size"]), num_buckets=int(cfg.get("relative_attention_num_buckets", 32)),
                   max_distance=int(cfg.get("relative_attention_max_distance", 128)))
    want = synthetic.shapes(synthetic.t5_encoder_weights, out)
    validate(header_shapes(component_dir), want, component_dir, optional={"encoder.embed_tokens.weight"})
    return read_tensors(component_dir, keys=set(want)), out


# ---------------------------------------------------------------------------------------------------------------- scorers
def load_pickscore(model_dir):
    """A transformers CLIPModel directory (yuvalkirstain/PickScore_v1; adv_grpo/pickscore_scorer.py:8-14) -> (state dict, ClipConfig)."""
    cfg = read_config(model_dir)
    _class(cfg, model_dir, "CLIPModel")
    v, t = cfg.get("vision_config") or {}, cfg.get("text_config") or {}
    need = lambda d, k, what: d[k] if k in d else (_ for _ in ()).throw(HubError(f"{model_dir}: {what}.{k} missing from config.json"))
    act = v.get("hidden_act", "gelu")
    if act != t.get("hidden_act", act) or act not in ("gelu", "quick_gelu"):
        raise HubError(f"{model_dir}: hidden_act {act!r} / {t.get('hidden_act')!r} is not implemented")
    out = ClipConfig(v_hidden=int(need(v, "hidden_size", "vision_config")), v_layers=int(need(v, "num_hidden_layers", "vision_config")),
                     v_heads=int(need(v, "num_attention_heads", "vision_config")), v_mlp=int(need(v, "intermediate_size", "vision_config")),
                     image_size=int(need(v, "image_size", "vision_config")), patch=int(need(v, "patch_size", "vision_config")),
                     t_hidden=int(need(t, "hidden_size", "text_config")), t_layers=int(need(t, "num_hidden_layers", "text_config")),
                     t_heads=int(need(t, "num_attention_heads", "text_config")), t_mlp=int(need(t, "intermediate_size", "text_config")),
                     vocab=int(t.get("vocab_size", 49408)), max_pos=int(t.get("max_position_embeddings", 77)),
                     proj=int(cfg.get("projection_dim", 1024)), eos_token_id=int(t.get("eos_token_id", 49407)), act=act)
    if out.patch != 14:
        raise HubError(f"{model_dir}: patch_size {out.patch}; the preprocessing kernels write 14 x 14 patch rows")
    want = synthetic.shapes(synthetic.clip_weights, out)
    validate(header_shapes(model_dir), want, model_dir,
             optional={"vision_model.embeddings.position_ids", "text_model.embeddings.position_ids"})
    return read_tensors(model_dir, keys=set(want)), out


def load_timm_dinov2(model_dir):
    """A timm hub directory of vit_base_patch14_dinov2.lvd142m (TD:589; config.json ``architecture`` + model.safetensors) -> (state dict,
    DinoConfig)."""
    cfg = read_config(model_dir)
    arch = cfg.get("architecture")
    if arch != "vit_base_patch14_dinov2":
        raise HubError(f"{model_dir}: architecture = {arch!r}; the reward path is built for 'vit_base_patch14_dinov2'")
    out = DinoConfig()
    size = (cfg.get("pretrained_cfg") or {}).get("input_size")
    if size is not None and list(size) != [3, out.image_size, out.image_size]:
        raise HubError(f"{model_dir}: pretrained_cfg.input_size = {size}; expected [3, {out.image_size}, {out.image_size}]")
    want = synthetic.shapes(synthetic.dino_weights, out)
    validate(header_shapes(model_dir), want, model_dir)
    return read_tensors(model_dir, keys=set(want)), out


# ---------------------------------------------------------------------------------------------------------------- Qwen-Image
def load_qwen_transformer(component_dir):
    """diffusers QwenImageTransformer2DModel -> (state dict, QwenMMDiTConfig)."""
    cfg = read_config(component_dir)
    _class(cfg, component_dir, "QwenImageTransformer2DModel")
    out = QwenMMDiTConfig(num_layers=int(cfg["num_layers"]), num_heads=int(cfg["num_attention_heads"]), head_dim=int(cfg["attention_head_dim"]),
                          in_channels=int(cfg["in_channels"]), out_channels=int(cfg.get("out_channels") or 16), patch_size=int(cfg["patch_size"]),
                          joint_attention_dim=int(cfg["joint_attention_dim"]), axes_dims_rope=tuple(cfg["axes_dims_rope"]))
    if out.head_dim != 128 or sum(out.axes_dims_rope) != out.head_dim:
        raise HubError(f"{component_dir}: head dim {out.head_dim} / rotary axes {out.axes_dims_rope}: built for 128 = 16 + 56 + 56-style splits")
    if cfg.get("guidance_embeds"):
        raise HubError(f"{component_dir}: guidance_embeds = true is not implemented")
    want = synthetic.shapes(synthetic.qwen_mmdit_weights, out)
    validate(header_shapes(component_dir), want, component_dir)
    return read_tensors(component_dir, keys=set(want)), out


def load_qwen_vae_decoder(component_dir):
    """diffusers AutoencoderKLQwenImage -> (post_quant_conv + decoder.* state dict, QwenVaeConfig)."""
    cfg = read_config(component_dir)
    _class(cfg, component_dir, "AutoencoderKLQwenImage")
    out = QwenVaeConfig(base_dim=int(cfg["base_dim"]), z_dim=int(cfg["z_dim"]), dim_mult=tuple(cfg["dim_mult"]),
                        num_res_blocks=int(cfg["num_res_blocks"]), latents_mean=tuple(cfg["latents_mean"]), latents_std=tuple(cfg["latents_std"]))
    want = synthetic.shapes(synthetic.qwen_vae_decoder_weights, out)
    keep = lambda k: k if (k.startswith("decoder.") or k.startswith("post_quant_conv.")) and ".time_conv." not in k else None
    have = {n: s for n, s in ((keep(k), s) for k, s in header_shapes(component_dir).items()) if n is not None}
    validate(have, want, component_dir)
    return {k: v.float() for k, v in read_tensors(component_dir, keys=set(want), rename=keep).items()}, out


_QWEN_TEXT_PREFIXES = ("model.language_model.", "language_model.model.", "model.")


def _qwen_text_rename(k):
    if k.startswith(("visual.", "model.visual.", "lm_head.")):
        return None
    for p in _QWEN_TEXT_PREFIXES:
        if k.startswith(p):
            return k[len(p):]
    return k


def load_qwen_text_encoder(component_dir):
    """Qwen2_5_VLForConditionalGeneration (Qwen-Image's text_encoder) -> (language-model state dict, QwenTextConfig); the vision tower and
    the LM head are not on the path."""
    cfg = read_config(component_dir)
    _class(cfg, component_dir, "Qwen2_5_VLForConditionalGeneration", "Qwen2_5_VLModel", "Qwen2_5_VLTextModel")
    t = cfg.get("text_config") or cfg
    out = QwenTextConfig(vocab_size=int(t["vocab_size"]), hidden_size=int(t["hidden_size"]), intermediate_size=int(t["intermediate_size"]),
                         num_layers=int(t["num_hidden_layers"]), num_heads=int(t["num_attention_heads"]),
                         num_kv_heads=int(t["num_key_value_heads"]), rms_eps=float(t.get("rms_norm_eps", 1e-6)),
                         rope_theta=float(t.get("rope_theta", 1e6)))
    want = synthetic.shapes(synthetic.qwen_text_weights, out)
    have = {n: s for n, s in ((_qwen_text_rename(k), s) for k, s in header_shapes(component_dir).items()) if n is not None}
    validate(have, want, component_dir)
    return read_tensors(component_dir, keys=set(want), rename=_qwen_text_rename), out


# ---------------------------------------------------------------------------------------------------------------- pipelines
def pipeline_kind(snapshot_dir):
    """'sd3' | 'qwen' from model_index.json (or from the transformer's class when the index is absent)."""
    idx = os.path.join(snapshot_dir, "model_index.json")
    name = None
    if os.path.isfile(idx):
        with open(idx) as f:
            name = json.load(f).get("_class_name")
    if name is None:
        name = read_config(os.path.join(snapshot_dir, "transformer")).get("_class_name")
    if name in ("StableDiffusion3Pipeline", "SD3Transformer2DModel"):
        return "sd3"
    if name in ("QwenImagePipeline", "QwenImageTransformer2DModel"):
        return "qwen"
    raise HubError(f"{snapshot_dir}: pipeline class {name!r} is not one this build serves (StableDiffusion3Pipeline, QwenImagePipeline)")


def load_pipeline(snapshot_dir, text_encoders=False):
    """-> dict: kind, transformer = (sd, cfg), vae = (sd, cfg) and, with text_encoders=True, the prompt encoders of the pipeline
    (sd3: text_encoder, text_encoder_2, text_encoder_3; qwen: text_encoder).  Everything is validated before anything is returned."""
    if not os.path.isdir(snapshot_dir):
        raise HubError(f"{snapshot_dir}: not a directory (pass the local snapshot of config.pretrained.model; nothing is downloaded here)")
    kind = pipeline_kind(snapshot_dir)
    sub = lambda n: os.path.join(snapshot_dir, n)
    out = {"kind": kind}
    if kind == "sd3":
        out["transformer"] = load_sd3_transformer(sub("transformer"))
        out["vae"] = load_vae_decoder(sub("vae"))
        if text_encoders:
            out["text_encoder"] = load_clip_text(sub("text_encoder"))
            out["text_encoder_2"] = load_clip_text(sub("text_encoder_2"))
            out["text_encoder_3"] = load_t5_encoder(sub("text_encoder_3"))
    else:
        out["transformer"] = load_qwen_transformer(sub("transformer"))
        out["vae"] = load_qwen_vae_decoder(sub("vae"))
        if text_encoders:
            out["text_encoder"] = load_qwen_text_encoder(sub("text_encoder"))
    return out
