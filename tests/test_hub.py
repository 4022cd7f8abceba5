"""adv_grpo_amd/hub.py: real-checkpoint loading, tested without a checkpoint -- synthetic snapshots are written to tmp_path in the Hugging
Face layout (model_index.json, per-component config.json, single-file and sharded safetensors, a legacy-named VAE, a Qwen2.5-VL
text encoder with its prefixes) and read back: tensors bit-identical, configs equal to the dataclasses that generated them, and every
kind of mismatch (wrong num_layers / dual_attention_layers / class, missing / unexpected / mis-shaped tensors, missing shards) refused
with the reason.  GPU: the loaded dicts build models whose outputs are bit-identical to the dict-built ones, through the launcher too."""
import dataclasses
import json
import os
import subprocess
import sys

import pytest
import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from adv_grpo_amd import hub, synthetic  # noqa: E402
from adv_grpo_amd.model_configs import (ClipConfig, ClipTextConfig, DinoConfig, MMDiTConfig, QwenMMDiTConfig, QwenTextConfig,  # noqa: E402
                                        QwenVaeConfig, T5Config, VaeConfig)

MM = MMDiTConfig(num_layers=3, num_heads=2, joint_attention_dim=64, pooled_projection_dim=64, pos_embed_max_size=16, dual_attention_layers=(0, 1))
VAE = VaeConfig(block_out_channels=(32, 32, 64, 64), norm_num_groups=8)
CL = ClipTextConfig(hidden=128, layers=2, heads=2, mlp=256, proj=64, vocab=300)
CG = ClipTextConfig(hidden=192, layers=2, heads=3, mlp=384, proj=96, vocab=300, act="gelu")
T5 = T5Config(d_model=128, layers=2, heads=2, d_ff=256, vocab=300)
PICK = ClipConfig(v_hidden=128, v_layers=2, v_heads=2, v_mlp=256, image_size=28, t_hidden=64, t_layers=2, t_heads=1, t_mlp=128, vocab=300, proj=32,
                  eos_token_id=299)
QM = QwenMMDiTConfig(num_layers=2, num_heads=2, joint_attention_dim=64)
QV = QwenVaeConfig(base_dim=16)
QT = QwenTextConfig(vocab_size=300, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1)


def _save(d, sd, stem="diffusion_pytorch_model", shards=1, dtype=None):
    os.makedirs(d, exist_ok=True)
    sd = {k: (v.to(dtype) if dtype else v).contiguous() for k, v in sd.items()}
    if shards == 1:
        save_file(sd, os.path.join(d, f"{stem}.safetensors"))
        return
    keys, wm = sorted(sd), {}
    for i in range(shards):
        name = f"{stem}-{i + 1:05d}-of-{shards:05d}.safetensors"
        part = {k: sd[k] for k in keys[i::shards]}
        save_file(part, os.path.join(d, name))
        wm.update({k: name for k in part})
    json.dump({"metadata": {}, "weight_map": wm}, open(os.path.join(d, f"{stem}.safetensors.index.json"), "w"))


def _cfg(d, **kw):
    os.makedirs(d, exist_ok=True)
    json.dump(kw, open(os.path.join(d, "config.json"), "w"))


def mmdit_json(c, **over):
    j = dict(_class_name="SD3Transformer2DModel", num_layers=c.num_layers, num_attention_heads=c.num_heads, attention_head_dim=c.head_dim,
             in_channels=c.in_channels, out_channels=c.out_channels, patch_size=c.patch_size, joint_attention_dim=c.joint_attention_dim,
             pooled_projection_dim=c.pooled_projection_dim, pos_embed_max_size=c.pos_embed_max_size, caption_projection_dim=c.dim,
             dual_attention_layers=list(c.dual_attention_layers), qk_norm="rms_norm", sample_size=128)
    j.update(over)
    return j


def write_sd3_snapshot(root, legacy_vae=False, MM=MM, VAE=VAE):
    """-> the dicts that were written (checkpoint dtypes: fp16 transformer / VAE, as released)."""
    os.makedirs(root, exist_ok=True)
    json.dump({"_class_name": "StableDiffusion3Pipeline"}, open(os.path.join(root, "model_index.json"), "w"))
    W = {"transformer": synthetic.mmdit_weights(MM, 1), "vae": synthetic.vae_decoder_weights(VAE, 2), "text_encoder": synthetic.clip_text_weights(CL, 3),
         "text_encoder_2": synthetic.clip_text_weights(CG, 4), "text_encoder_3": synthetic.t5_encoder_weights(T5, 5)}
    _cfg(os.path.join(root, "transformer"), **mmdit_json(MM))
    _save(os.path.join(root, "transformer"), W["transformer"], shards=3, dtype=torch.float16)
    _cfg(os.path.join(root, "vae"), _class_name="AutoencoderKL", act_fn="silu", latent_channels=16, block_out_channels=list(VAE.block_out_channels),
         layers_per_block=2, norm_num_groups=VAE.norm_num_groups, scaling_factor=1.5305, shift_factor=0.0609, use_post_quant_conv=False, use_quant_conv=False,
         mid_block_add_attention=True)
    vae = dict(W["vae"])
    vae["encoder.conv_in.weight"] = torch.zeros(8, 3, 3, 3)                      # a full AutoencoderKL carries its encoder too
    if legacy_vae:
        a = "decoder.mid_block.attentions.0"
        for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            vae[f"{a}.{old}.weight"] = vae.pop(f"{a}.{new}.weight")[:, :, None, None]
            vae[f"{a}.{old}.bias"] = vae.pop(f"{a}.{new}.bias")
    _save(os.path.join(root, "vae"), vae, dtype=torch.float16)
    for name, c in (("text_encoder", CL), ("text_encoder_2", CG)):
        _cfg(os.path.join(root, name), architectures=["CLIPTextModelWithProjection"], hidden_size=c.hidden, num_hidden_layers=c.layers,
             num_attention_heads=c.heads, intermediate_size=c.mlp, projection_dim=c.proj, vocab_size=c.vocab, max_position_embeddings=77,
             hidden_act=c.act, eos_token_id=2)
        sd = dict(W[name]); sd["text_model.embeddings.position_ids"] = torch.arange(77)[None]
        _save(os.path.join(root, name), sd, stem="model")
    _cfg(os.path.join(root, "text_encoder_3"), architectures=["T5EncoderModel"], d_model=T5.d_model, num_layers=T5.layers, num_heads=T5.heads,
         d_kv=64, d_ff=T5.d_ff, vocab_size=T5.vocab, feed_forward_proj="gated-gelu", relative_attention_num_buckets=32,
         relative_attention_max_distance=128)
    sd = dict(W["text_encoder_3"]); sd["encoder.embed_tokens.weight"] = sd["shared.weight"].clone()
    _save(os.path.join(root, "text_encoder_3"), sd, stem="model", shards=2)
    return W


def test_sd3_snapshot_round_trip(tmp_path):
    W = write_sd3_snapshot(str(tmp_path))
    out = hub.load_pipeline(str(tmp_path), text_encoders=True)
    assert out["kind"] == "sd3"
    assert out["transformer"][1] == MM and out["vae"][1] == VAE
    assert out["text_encoder"][1] == CL and out["text_encoder_2"][1] == CG and out["text_encoder_3"][1] == T5
    for name, cast in (("transformer", lambda t: t.half()), ("vae", lambda t: t.half().float()), ("text_encoder", lambda t: t),
                       ("text_encoder_2", lambda t: t), ("text_encoder_3", lambda t: t)):
        sd = out[name][0]
        assert set(sd) == set(W[name])
        for k, v in W[name].items():
            assert torch.equal(sd[k], cast(v)), (name, k)
    assert all(v.dtype == torch.float32 for v in out["vae"][0].values())          # vae.to(torch.float32), TP:481


def test_legacy_vae_attention_names_are_converted(tmp_path):
    W = write_sd3_snapshot(str(tmp_path), legacy_vae=True)
    sd, cfg = hub.load_vae_decoder(str(tmp_path / "vae"))
    assert cfg == VAE and set(sd) == set(W["vae"])
    for k, v in W["vae"].items():
        assert torch.equal(sd[k], v.half().float()), k


def test_absent_config_fields_read_as_the_library_defaults(tmp_path):
    """a vae/config.json saved before `mid_block_add_attention` / `act_fn` existed omits them and diffusers builds the model with
    its constructor defaults (True / "silu") -- accepted; an absent `use_post_quant_conv` means diffusers' default True -- still refused.
    Likewise T5's d_kv (transformers' default 64)."""
    write_sd3_snapshot(str(tmp_path))
    vdir, tdir = str(tmp_path / "vae"), str(tmp_path / "text_encoder_3")
    j = json.load(open(os.path.join(vdir, "config.json")))
    for k in ("mid_block_add_attention", "act_fn"):
        j.pop(k, None)
    _cfg(vdir, **j)
    sd, cfg = hub.load_vae_decoder(vdir)
    assert cfg == VAE
    j.pop("use_post_quant_conv")
    _cfg(vdir, **j)
    with pytest.raises(hub.HubError, match="use_post_quant_conv"):
        hub.load_vae_decoder(vdir)
    t = json.load(open(os.path.join(tdir, "config.json")))
    t.pop("d_kv", None)
    _cfg(tdir, **t)
    assert hub.load_t5_encoder(tdir)[1] == T5


def test_pickscore_config_with_the_legacy_eos_token_id(tmp_path):
    """the released PickScore_v1 config.json carries text_config.eos_token_id = 2; the loader keeps it and the towers
    pool at argmax(input_ids) for it (vit.py / vit_x3.py; pinned vs transformers in tests/test_oracle_vit.py, on the GPU in test_gpu_vit.py)."""
    import dataclasses
    Wp = synthetic.clip_weights(PICK, 7)
    p = str(tmp_path / "pick")
    _cfg(p, architectures=["CLIPModel"], projection_dim=PICK.proj,
         vision_config=dict(hidden_size=PICK.v_hidden, num_hidden_layers=PICK.v_layers, num_attention_heads=PICK.v_heads,
                            intermediate_size=PICK.v_mlp, image_size=PICK.image_size, patch_size=14, hidden_act="gelu"),
         text_config=dict(hidden_size=PICK.t_hidden, num_hidden_layers=PICK.t_layers, num_attention_heads=PICK.t_heads,
                          intermediate_size=PICK.t_mlp, vocab_size=PICK.vocab, max_position_embeddings=77, eos_token_id=2, hidden_act="gelu"))
    torch.save(Wp, os.path.join(p, "pytorch_model.bin"))
    sd, cfg = hub.load_pickscore(p)
    assert cfg == dataclasses.replace(PICK, eos_token_id=2)
    from oracle import vit as o
    ids = torch.randint(3, PICK.vocab - 1, (3, 77)); ids[0, 9] = PICK.vocab - 1; ids[1, 30] = PICK.vocab - 1; ids[2, 76] = PICK.vocab - 1
    ids[:, 4] = 2                                                                 # an id-2 token before the end is not the pooled position
    W32 = {k: v.float() for k, v in sd.items()}
    a = o.clip_text_features(W32, cfg, ids)
    b = o.clip_text_features(W32, dataclasses.replace(cfg, eos_token_id=PICK.vocab - 1), ids)
    assert torch.allclose(a, b) and (a[0] - a[1]).abs().max() > 1e-4


@pytest.mark.parametrize("over, match", [
    (dict(num_layers=4), "missing"),                                   # config says 4 blocks, the weights hold 3
    (dict(num_layers=2), "unexpected"),
    (dict(dual_attention_layers=[0]), "unexpected"),                   # attn2 weights of block 1 have no place
    (dict(dual_attention_layers=[0, 1, 2]), "missing"),
    (dict(dual_attention_layers=[0, 7]), "outside"),
    (dict(pos_embed_max_size=24), "another shape"),
    (dict(_class_name="FluxTransformer2DModel"), "describes"),
    (dict(qk_norm="layer_norm"), "qk_norm"),
    (dict(attention_head_dim=128), "head dim"),
    (dict(caption_projection_dim=999), "caption_projection_dim"),
])
def test_transformer_config_mismatches_are_refused(tmp_path, over, match):
    write_sd3_snapshot(str(tmp_path))
    _cfg(str(tmp_path / "transformer"), **mmdit_json(MM, **over))
    with pytest.raises(hub.HubError, match=match):
        hub.load_pipeline(str(tmp_path))


def test_damaged_snapshots_are_refused(tmp_path):
    write_sd3_snapshot(str(tmp_path))
    t = tmp_path / "transformer"
    with pytest.raises(hub.HubError, match="not a directory"):
        hub.load_pipeline(str(tmp_path / "nope"))
    os.rename(t / "diffusion_pytorch_model-00002-of-00003.safetensors", t / "gone")
    with pytest.raises(hub.HubError, match="shards that are not there"):
        hub.load_sd3_transformer(str(t))
    os.rename(t / "gone", t / "diffusion_pytorch_model-00002-of-00003.safetensors")
    hub.load_sd3_transformer(str(t))
    # a tensor of another shape, a tensor that should not be there
    sd = synthetic.mmdit_weights(MM, 1)
    sd["proj_out.weight"] = torch.zeros(7, MM.dim)
    for f in os.listdir(t):
        if f.endswith(".safetensors") or f.endswith(".index.json"):
            os.remove(t / f)
    _save(str(t), sd)
    with pytest.raises(hub.HubError, match="proj_out.weight"):
        hub.load_sd3_transformer(str(t))
    os.remove(tmp_path / "vae" / "config.json")
    with pytest.raises(hub.HubError, match="config.json"):
        hub.load_vae_decoder(str(tmp_path / "vae"))
    os.remove(tmp_path / "text_encoder" / "model.safetensors")
    with pytest.raises(hub.HubError, match="no weights found"):
        hub.load_clip_text(str(tmp_path / "text_encoder"))


def test_scorer_directories(tmp_path):
    Wp, Wd = synthetic.clip_weights(PICK, 7), synthetic.dino_weights(DinoConfig(layers=1), 8)
    p, d = str(tmp_path / "pick"), str(tmp_path / "dino")
    _cfg(p, architectures=["CLIPModel"], projection_dim=PICK.proj,
         vision_config=dict(hidden_size=PICK.v_hidden, num_hidden_layers=PICK.v_layers, num_attention_heads=PICK.v_heads,
                            intermediate_size=PICK.v_mlp, image_size=PICK.image_size, patch_size=14, hidden_act="gelu"),
         text_config=dict(hidden_size=PICK.t_hidden, num_hidden_layers=PICK.t_layers, num_attention_heads=PICK.t_heads,
                          intermediate_size=PICK.t_mlp, vocab_size=PICK.vocab, max_position_embeddings=77, eos_token_id=299, hidden_act="gelu"))
    os.makedirs(p, exist_ok=True)
    torch.save(Wp, os.path.join(p, "pytorch_model.bin"))                         # PickScore_v1 also ships the pickle: read when no safetensors
    sd, cfg = hub.load_pickscore(p)
    assert cfg == PICK and all(torch.equal(sd[k], v) for k, v in Wp.items())
    _cfg(d, architecture="vit_base_patch14_dinov2", pretrained_cfg=dict(input_size=[3, 518, 518]))
    _save(d, Wd, stem="model")
    with pytest.raises(hub.HubError, match="missing"):                           # one block instead of twelve
        hub.load_timm_dinov2(d)
    _cfg(d, architecture="vit_large_patch14_dinov2")
    with pytest.raises(hub.HubError, match="architecture"):
        hub.load_timm_dinov2(d)


def test_dino_directory_full_layout(tmp_path):
    W = {k: torch.zeros(v, dtype=torch.float16) for k, v in synthetic.shapes(synthetic.dino_weights, DinoConfig()).items()}
    d = str(tmp_path / "dino")
    _cfg(d, architecture="vit_base_patch14_dinov2", pretrained_cfg=dict(input_size=[3, 518, 518]))
    _save(d, W, stem="model")
    sd, cfg = hub.load_timm_dinov2(d)
    assert cfg == DinoConfig() and set(sd) == set(W)


def test_qwen_snapshot_round_trip(tmp_path):
    root = str(tmp_path)
    json.dump({"_class_name": "QwenImagePipeline"}, open(os.path.join(root, "model_index.json"), "w"))
    Wt, Wv, Wx = synthetic.qwen_mmdit_weights(QM, 1), synthetic.qwen_vae_decoder_weights(QV, 2), synthetic.qwen_text_weights(QT, 3)
    _cfg(os.path.join(root, "transformer"), _class_name="QwenImageTransformer2DModel", num_layers=2, num_attention_heads=2, attention_head_dim=128,
         in_channels=64, out_channels=16, patch_size=2, joint_attention_dim=64, axes_dims_rope=[16, 56, 56], guidance_embeds=False)
    _save(os.path.join(root, "transformer"), Wt, shards=2, dtype=torch.bfloat16)
    _cfg(os.path.join(root, "vae"), _class_name="AutoencoderKLQwenImage", base_dim=16, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2,
         latents_mean=list(QV.latents_mean), latents_std=list(QV.latents_std))
    vae = dict(Wv); vae["encoder.conv_in.weight"] = torch.zeros(4, 3, 3, 3, 3); vae["decoder.up_blocks.0.upsamplers.0.time_conv.weight"] = torch.zeros(4, 4, 3, 1, 1)
    _save(os.path.join(root, "vae"), vae, dtype=torch.bfloat16)
    _cfg(os.path.join(root, "text_encoder"), architectures=["Qwen2_5_VLForConditionalGeneration"],
         text_config=dict(vocab_size=300, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                          rms_norm_eps=1e-6, rope_theta=1e6))
    txt = {f"model.language_model.{k}": v for k, v in Wx.items()}
    txt["model.visual.patch_embed.proj.weight"] = torch.zeros(4, 4); txt["lm_head.weight"] = torch.zeros(300, 256)
    _save(os.path.join(root, "text_encoder"), txt, stem="model", shards=2)
    out = hub.load_pipeline(root, text_encoders=True)
    assert out["kind"] == "qwen" and out["transformer"][1] == QM and out["vae"][1] == QV and out["text_encoder"][1] == QT
    assert all(torch.equal(out["transformer"][0][k], v.to(torch.bfloat16)) for k, v in Wt.items())
    assert all(torch.equal(out["vae"][0][k], v.to(torch.bfloat16).float()) for k, v in Wv.items())
    assert set(out["text_encoder"][0]) == set(Wx) and all(torch.equal(out["text_encoder"][0][k], v) for k, v in Wx.items())


def test_shapes_checklists_match_the_builders():
    """synthetic.shapes() (the validator's checklist) is exactly what the builders create."""
    for b, c in ((synthetic.mmdit_weights, MM), (synthetic.vae_decoder_weights, VAE), (synthetic.clip_text_weights, CL), (synthetic.t5_encoder_weights, T5),
                 (synthetic.clip_weights, PICK), (synthetic.qwen_mmdit_weights, QM), (synthetic.qwen_vae_decoder_weights, QV), (synthetic.qwen_text_weights, QT)):
        assert synthetic.shapes(b, c) == {k: tuple(v.shape) for k, v in b(c).items()}, b.__name__


@pytest.mark.gpu
def test_loaded_models_match_dict_built_models(tmp_path):
    """SD3 transformer, VAE decoder and the three prompt encoders built from the snapshot give the bits of the models built from the dicts
    that were written (same checkpoint dtypes)."""
    from adv_grpo_amd.mmdit import SD3Transformer2DModel
    from adv_grpo_amd.text_encoders import CLIPTextEncoder, T5Encoder
    from adv_grpo_amd.vae import AutoencoderKLDecoder
    FULL_VAE = VaeConfig()                                   # (the decoder kernels are built for the released widths)
    W = write_sd3_snapshot(str(tmp_path), VAE=FULL_VAE)
    out = hub.load_pipeline(str(tmp_path), text_encoders=True)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(2, 16, 16, 16, generator=g).cuda().to(torch.bfloat16)
    ctx = torch.randn(2, 20, 64, generator=g).cuda().to(torch.bfloat16)
    pooled = torch.randn(2, 64, generator=g).cuda().to(torch.bfloat16)
    t = torch.tensor([500.0, 500.0]).cuda()
    a = SD3Transformer2DModel({k: v.half() for k, v in W["transformer"].items()}, MM, "cuda")
    b = SD3Transformer2DModel(*out["transformer"], "cuda")
    assert torch.equal(a(lat, t, ctx, pooled)[0], b(lat, t, ctx, pooled)[0])
    va = AutoencoderKLDecoder({k: v.half().float() for k, v in W["vae"].items()}, FULL_VAE, "cuda", mode="bf16x3")
    vb = AutoencoderKLDecoder(*out["vae"], "cuda", mode="bf16x3")
    z = torch.randn(1, 16, 8, 8, generator=g).cuda()
    assert torch.equal(va.decode_to_image(z), vb.decode_to_image(z))
    ids = torch.randint(3, 290, (2, 77), generator=g); ids[:, 30:] = 299
    for name, c in (("text_encoder", CL), ("text_encoder_2", CG)):
        ea = CLIPTextEncoder(W[name], c.layers, c.heads, c.act, c.eos_token_id)
        eb = CLIPTextEncoder(out[name][0], out[name][1].layers, out[name][1].heads, out[name][1].act, out[name][1].eos_token_id)
        assert all(torch.equal(x, y) for x, y in zip(ea(ids), eb(ids)))
    ta, tb = T5Encoder(W["text_encoder_3"], T5.layers, T5.heads), T5Encoder(out["text_encoder_3"][0], T5.layers, T5.heads)
    assert torch.equal(ta(ids[:, :40]), tb(ids[:, :40]))


@pytest.mark.gpu
def test_launcher_runs_an_epoch_from_a_snapshot(tmp_path):
    """scripts/train_sd3_fast.py --weights <snapshot> --pickscore-weights <dir>: one G epoch of the kept entry point on models loaded
    from disk (reduced-depth SD3 transformer with the released prompt widths, the full VAE, a small CLIP scorer); a snapshot whose
    config.json disagrees with its weights stops the launcher before any model is built."""
    mm = MMDiTConfig(num_layers=2, num_heads=2, pos_embed_max_size=64, dual_attention_layers=(0,))
    write_sd3_snapshot(str(tmp_path / "sd3"), MM=mm, VAE=VaeConfig())
    pick = ClipConfig(v_hidden=128, v_layers=2, v_heads=2, v_mlp=256, t_hidden=64, t_layers=2, t_heads=1, t_mlp=128, proj=32)
    p = str(tmp_path / "pick")
    _cfg(p, architectures=["CLIPModel"], projection_dim=pick.proj,
         vision_config=dict(hidden_size=pick.v_hidden, num_hidden_layers=pick.v_layers, num_attention_heads=pick.v_heads,
                            intermediate_size=pick.v_mlp, image_size=224, patch_size=14, hidden_act="gelu"),
         text_config=dict(hidden_size=pick.t_hidden, num_hidden_layers=pick.t_layers, num_attention_heads=pick.t_heads,
                          intermediate_size=pick.t_mlp, vocab_size=pick.vocab, max_position_embeddings=77, eos_token_id=49407, hidden_act="gelu"))
    _save(p, synthetic.clip_weights(pick, 7), stem="model")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_sd3_fast.py"), "--config",
           os.path.join(ROOT, "config", "grpo.py") + ":pickscore_cotrain_sd3_fast", "--resolution", "256", "--epochs", "1", "--batches", "1",
           "--images-per-prompt", "8", "--no-train-d", "--weights", str(tmp_path / "sd3"), "--pickscore-weights", p,
           "--log", str(tmp_path / "log.jsonl")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["phase"] == "G" and line["global_step"] == 1
    _cfg(str(tmp_path / "sd3" / "transformer"), **mmdit_json(mm, num_layers=3))
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "HubError" in r.stderr and "missing" in r.stderr
