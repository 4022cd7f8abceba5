"""Seeded synthetic weights / inputs with the exact shapes of the models on the hot path.

No checkpoints or tokenizers exist on the build or GPU boxes (no network), so measurements and
parity tests use seeded random weights of the real architecture: SD3.5-medium
MMDiT-X, SD3 VAE decoder, CLIP ViT-H/14 (PickScore), DINOv2 ViT-B/14, DINO head.  Keys follow the
upstream state_dict names (diffusers / transformers), so real checkpoints can replace them.
Weights ~ N(0, 1/fan_in) (keeps activations O(1) through 24 blocks, which makes parity tests
sensitive to every block), biases ~ N(0, 0.1^2), norm weights ~ 1 + N(0, 0.1^2).
"""
import math

import numpy as np
import torch


_DEVICE = "cpu"


class on_device:
    """Context manager: draw the synthetic weights directly on a device (bench start-up time).  The
    values then come from that device's generator stream, not the CPU one the parity tests use."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        global _DEVICE
        self.prev, _DEVICE = _DEVICE, self.device

    def __exit__(self, *a):
        global _DEVICE
        _DEVICE = self.prev


class _ShapeOnly:
    """Stand-in generator of the "meta" device: the builders then produce tensors with shapes and no storage."""
    device = torch.device("meta")


def _gen(seed):
    if str(_DEVICE) == "meta":
        return _ShapeOnly()
    return torch.Generator(device=_DEVICE).manual_seed(seed)


def _randn(*shape, generator):
    if generator.device.type == "meta":
        return torch.empty(*shape, device="meta")
    return torch.randn(*shape, generator=generator, device=generator.device)


def shapes(builder, *args, **kwargs):
    """{key: shape} of what ``builder(*args, **kwargs)`` (one of the *_weights functions below) would create, without creating it:
    the architecture's state-dict layout as a checklist (adv_grpo_amd/hub.py validates checkpoints against it)."""
    with on_device("meta"):
        return {k: tuple(v.shape) for k, v in builder(*args, **kwargs).items()}


def linear_(W, name, out_f, in_f, g, bias=True, std=None):
    std = (1.0 / math.sqrt(in_f)) if std is None else std
    W[name + ".weight"] = _randn(out_f, in_f, generator=g) * std
    if bias:
        W[name + ".bias"] = _randn(out_f, generator=g) * 0.1


def sincos_pos_embed_2d(dim, grid_size, base_size):
    """diffusers get_2d_sincos_pos_embed(dim, grid_size, base_size=base_size, interpolation_scale=1)."""
    def emb1d(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh = np.arange(grid_size, dtype=np.float32) / (grid_size / base_size)
    gw = np.arange(grid_size, dtype=np.float32) / (grid_size / base_size)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)
    emb = np.concatenate([emb1d(dim // 2, grid[0]), emb1d(dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).float().unsqueeze(0).to(_DEVICE)


def mmdit_weights(cfg, seed=1234):
    """fp32 CPU weights dict keyed like diffusers SD3Transformer2DModel.state_dict()."""
    g = _gen(seed)
    D, W = cfg.dim, {}
    ps, C = cfg.patch_size, cfg.in_channels
    W["pos_embed.proj.weight"] = _randn(D, C, ps, ps, generator=g) / math.sqrt(C * ps * ps)
    W["pos_embed.proj.bias"] = _randn(D, generator=g) * 0.1
    W["pos_embed.pos_embed"] = sincos_pos_embed_2d(D, cfg.pos_embed_max_size, 64)
    linear_(W, "time_text_embed.timestep_embedder.linear_1", D, 256, g)
    linear_(W, "time_text_embed.timestep_embedder.linear_2", D, D, g)
    linear_(W, "time_text_embed.text_embedder.linear_1", D, cfg.pooled_projection_dim, g)
    linear_(W, "time_text_embed.text_embedder.linear_2", D, D, g)
    linear_(W, "context_embedder", D, cfg.joint_attention_dim, g)
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}"
        dual = i in cfg.dual_attention_layers
        last = i == cfg.num_layers - 1
        linear_(W, f"{p}.norm1.linear", (9 if dual else 6) * D, D, g, std=0.5 / math.sqrt(D))
        linear_(W, f"{p}.norm1_context.linear", (2 if last else 6) * D, D, g, std=0.5 / math.sqrt(D))
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0"):
            linear_(W, f"{p}.attn.{n}", D, D, g)
        if not last:
            linear_(W, f"{p}.attn.to_add_out", D, D, g)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            W[f"{p}.attn.{n}.weight"] = 1 + 0.1 * _randn(cfg.head_dim, generator=g)
        if dual:
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                linear_(W, f"{p}.attn2.{n}", D, D, g)
            for n in ("norm_q", "norm_k"):
                W[f"{p}.attn2.{n}.weight"] = 1 + 0.1 * _randn(cfg.head_dim, generator=g)
        linear_(W, f"{p}.ff.net.0.proj", 4 * D, D, g)
        linear_(W, f"{p}.ff.net.2", D, 4 * D, g)
        if not last:
            linear_(W, f"{p}.ff_context.net.0.proj", 4 * D, D, g)
            linear_(W, f"{p}.ff_context.net.2", D, 4 * D, g)
    linear_(W, "norm_out.linear", 2 * D, D, g, std=0.5 / math.sqrt(D))
    linear_(W, "proj_out", ps * ps * cfg.out_channels, D, g)
    return W


def qwen_mmdit_weights(cfg, seed=4242, dtype=None):
    """Weights dict keyed like diffusers QwenImageTransformer2DModel.state_dict() (Qwen/Qwen-Image: 60 blocks of two-stream
    attention + MLP, 20 B parameters at full size).  dtype: cast every tensor as it is drawn (bf16 at full size: an fp32 copy
    of the whole model is 82 GB); values are the bf16 rounding of the same fp32 stream either way."""
    g = _gen(seed)
    D, W = cfg.dim, {}
    cast = (lambda t: t) if dtype is None else (lambda t: t.to(dtype))

    def lin(name, out_f, in_f, std=None):
        tmp = {}
        linear_(tmp, name, out_f, in_f, g, std=std)
        for k, v in tmp.items():
            W[k] = cast(v)
    lin("img_in", D, cfg.in_channels)
    lin("txt_in", D, cfg.joint_attention_dim)
    W["txt_norm.weight"] = cast(1 + 0.1 * _randn(cfg.joint_attention_dim, generator=g))
    lin("time_text_embed.timestep_embedder.linear_1", D, 256)
    lin("time_text_embed.timestep_embedder.linear_2", D, D)
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}"
        lin(f"{p}.img_mod.1", 6 * D, D, std=0.5 / math.sqrt(D))
        lin(f"{p}.txt_mod.1", 6 * D, D, std=0.5 / math.sqrt(D))
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(f"{p}.attn.{n}", D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            W[f"{p}.attn.{n}.weight"] = cast(1 + 0.1 * _randn(cfg.head_dim, generator=g))
        for s in ("img_mlp", "txt_mlp"):
            lin(f"{p}.{s}.net.0.proj", 4 * D, D)
            lin(f"{p}.{s}.net.2", D, 4 * D)
    lin("norm_out.linear", 2 * D, D, std=0.5 / math.sqrt(D))
    lin("proj_out", cfg.patch_size * cfg.patch_size * cfg.out_channels, D)
    return W


def prompt_embeddings(seed=7, n_tokens=205, ctx_dim=4096, pooled_dim=2048):
    """Synthetic prompt: (prompt_embeds [1,205,4096], pooled [1,2048], negative ..., negative pooled ...)."""
    g = torch.Generator().manual_seed(seed)   # inputs always come from the CPU stream
    return (torch.randn(1, n_tokens, ctx_dim, generator=g), torch.randn(1, pooled_dim, generator=g),
            torch.randn(1, n_tokens, ctx_dim, generator=g), torch.randn(1, pooled_dim, generator=g))


def conv_(W, name, cout, cin, k, g):
    W[name + ".weight"] = _randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    W[name + ".bias"] = _randn(cout, generator=g) * 0.1


def norm_(W, name, c, g):
    W[name + ".weight"] = 1 + 0.1 * _randn(c, generator=g)
    W[name + ".bias"] = 0.1 * _randn(c, generator=g)


def vae_decoder_weights(cfg, seed=4321, fp16_checkpoint=False):
    """fp32 CPU weights keyed like diffusers AutoencoderKL.state_dict() (decoder.* only).  fp16_checkpoint: every tensor rounded
    to fp16 and upcast again -- what the reference holds after loading the released fp16 VAE checkpoint and vae.to(float32)
    (TP:447,481); such weights are exact in one 16-bit piece (the decoder's f16x2 path, adv_grpo_amd/vae.py)."""
    g = _gen(seed)
    W = {}
    ch = list(reversed(cfg.block_out_channels))          # 512, 512, 256, 128
    conv_(W, "decoder.conv_in", ch[0], cfg.latent_channels, 3, g)

    def res(p, ci, co):
        norm_(W, f"{p}.norm1", ci, g); conv_(W, f"{p}.conv1", co, ci, 3, g)
        norm_(W, f"{p}.norm2", co, g); conv_(W, f"{p}.conv2", co, co, 3, g)
        if ci != co:
            conv_(W, f"{p}.conv_shortcut", co, ci, 1, g)
    res("decoder.mid_block.resnets.0", ch[0], ch[0])
    a = "decoder.mid_block.attentions.0"
    norm_(W, f"{a}.group_norm", ch[0], g)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        linear_(W, f"{a}.{n}", ch[0], ch[0], g)
    res("decoder.mid_block.resnets.1", ch[0], ch[0])
    prev = ch[0]
    for i, co in enumerate(ch):
        for j in range(cfg.layers_per_block + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        prev = co
        if i < len(ch) - 1:
            conv_(W, f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3, g)
    norm_(W, "decoder.conv_norm_out", ch[-1], g)
    conv_(W, "decoder.conv_out", 3, ch[-1], 3, g)
    if fp16_checkpoint:
        W = {k: v.half().float() for k, v in W.items()}
    return W


def qwen_text_weights(cfg, seed=1357, dtype=None):
    """Weights keyed like transformers' Qwen2_5_VLTextModel.state_dict() (created on torch's default device: use on_device)."""
    g = _gen(seed)
    W = {"embed_tokens.weight": _randn(cfg.vocab_size, cfg.hidden_size, generator=g), "norm.weight": 1 + 0.1 * _randn(cfg.hidden_size, generator=g)}
    D, hd = cfg.hidden_size, cfg.head_dim
    for i in range(cfg.num_layers):
        p, a = f"layers.{i}", f"layers.{i}.self_attn"
        for n, out_f in (("q_proj", cfg.num_heads * hd), ("k_proj", cfg.num_kv_heads * hd), ("v_proj", cfg.num_kv_heads * hd)):
            W[f"{a}.{n}.weight"] = _randn(out_f, D, generator=g) / math.sqrt(D)
            W[f"{a}.{n}.bias"] = 0.1 * _randn(out_f, generator=g)
        W[f"{a}.o_proj.weight"] = _randn(D, cfg.num_heads * hd, generator=g) / math.sqrt(D)
        W[f"{p}.mlp.gate_proj.weight"] = _randn(cfg.intermediate_size, D, generator=g) / math.sqrt(D)
        W[f"{p}.mlp.up_proj.weight"] = _randn(cfg.intermediate_size, D, generator=g) / math.sqrt(D)
        W[f"{p}.mlp.down_proj.weight"] = _randn(D, cfg.intermediate_size, generator=g) / math.sqrt(cfg.intermediate_size)
        W[f"{p}.input_layernorm.weight"] = 1 + 0.1 * _randn(D, generator=g)
        W[f"{p}.post_attention_layernorm.weight"] = 1 + 0.1 * _randn(D, generator=g)
    if dtype is not None:
        W = {k: v.to(dtype) for k, v in W.items()}
    return W


def qwen_vae_decoder_weights(cfg, seed=2468, dtype=None):
    """fp32 CPU weights keyed like diffusers AutoencoderKLQwenImage.state_dict() (post_quant_conv + decoder.*; the upsamplers'
    `time_conv`, which a still image never runs, is left out).  3-D kernels [Co, Ci, kt, kh, kw]; dtype: round every tensor
    through it (the released checkpoint is bf16)."""
    g = _gen(seed)
    W = {}

    def conv3d(name, co, ci, k):
        W[name + ".weight"] = _randn(co, ci, k, k, k, generator=g) / math.sqrt(ci * k * k)      # (only one temporal tap meets data)
        W[name + ".bias"] = _randn(co, generator=g) * 0.1

    def gamma(name, c, nd):
        W[name + ".gamma"] = (1 + 0.1 * _randn(c, generator=g)).view(c, *([1] * nd))

    def res(p, ci, co):
        gamma(f"{p}.norm1", ci, 3); conv3d(f"{p}.conv1", co, ci, 3)
        gamma(f"{p}.norm2", co, 3); conv3d(f"{p}.conv2", co, co, 3)
        if ci != co:
            conv3d(f"{p}.conv_shortcut", co, ci, 1)
    d = cfg.dims
    conv3d("post_quant_conv", cfg.z_dim, cfg.z_dim, 1)
    conv3d("decoder.conv_in", d[0], cfg.z_dim, 3)
    res("decoder.mid_block.resnets.0", d[0], d[0])
    a = "decoder.mid_block.attentions.0"
    gamma(f"{a}.norm", d[0], 2)
    conv_(W, f"{a}.to_qkv", 3 * d[0], d[0], 1, g)
    conv_(W, f"{a}.proj", d[0], d[0], 1, g)
    res("decoder.mid_block.resnets.1", d[0], d[0])
    for i in range(len(cfg.dim_mult)):
        ci, co, up = cfg.up_block_io(i)
        for j in range(cfg.num_res_blocks + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", ci if j == 0 else co, co)
        if up:
            conv_(W, f"decoder.up_blocks.{i}.upsamplers.0.resample.1", co // 2, co, 3, g)
    gamma("decoder.norm_out", d[-1], 3)
    conv3d("decoder.conv_out", 3, d[-1], 3)
    if dtype is not None:
        W = {k: v.to(dtype).float() for k, v in W.items()}
    return W


def clip_text_weights(cfg, seed=555):
    """fp32 weights keyed like transformers CLIPTextModelWithProjection.state_dict() (SD3's text_encoder / text_encoder_2)."""
    g = _gen(seed)
    D = cfg.hidden
    W = {"text_model.embeddings.token_embedding.weight": _randn(cfg.vocab, D, generator=g) * 0.5,
         "text_model.embeddings.position_embedding.weight": _randn(cfg.max_pos, D, generator=g) * 0.1}
    for i in range(cfg.layers):
        p = f"text_model.encoder.layers.{i}"
        _ln(W, f"{p}.layer_norm1", D, g); _ln(W, f"{p}.layer_norm2", D, g)
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            linear_(W, f"{p}.self_attn.{nm}", D, D, g)
        linear_(W, f"{p}.mlp.fc1", cfg.mlp, D, g)
        linear_(W, f"{p}.mlp.fc2", D, cfg.mlp, g)
    _ln(W, "text_model.final_layer_norm", D, g)
    linear_(W, "text_projection", cfg.proj, D, g, bias=False)
    return W


def t5_encoder_weights(cfg, seed=666):
    """fp32 weights keyed like transformers T5EncoderModel.state_dict() (T5 v1.1: gated GELU, no biases; SD3's text_encoder_3).
    ``encoder.embed_tokens.weight`` is the tied copy of ``shared.weight`` a checkpoint may or may not carry: not listed."""
    g = _gen(seed)
    D, inner = cfg.d_model, cfg.heads * cfg.d_kv
    W = {"shared.weight": _randn(cfg.vocab, D, generator=g),
         "encoder.final_layer_norm.weight": 1 + 0.1 * _randn(D, generator=g),
         "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": _randn(cfg.num_buckets, cfg.heads, generator=g)}
    for i in range(cfg.layers):
        p = f"encoder.block.{i}.layer"
        W[f"{p}.0.layer_norm.weight"] = 1 + 0.1 * _randn(D, generator=g)
        W[f"{p}.1.layer_norm.weight"] = 1 + 0.1 * _randn(D, generator=g)
        for n in ("q", "k", "v"):
            W[f"{p}.0.SelfAttention.{n}.weight"] = _randn(inner, D, generator=g) * ((D * cfg.d_kv) ** -0.5 if n == "q" else D ** -0.5)
        W[f"{p}.0.SelfAttention.o.weight"] = _randn(D, inner, generator=g) * inner ** -0.5
        W[f"{p}.1.DenseReluDense.wi_0.weight"] = _randn(cfg.d_ff, D, generator=g) * D ** -0.5
        W[f"{p}.1.DenseReluDense.wi_1.weight"] = _randn(cfg.d_ff, D, generator=g) * D ** -0.5
        W[f"{p}.1.DenseReluDense.wo.weight"] = _randn(D, cfg.d_ff, generator=g) * cfg.d_ff ** -0.5
    return W


def _ln(W, name, d, g):
    W[name + ".weight"] = 1 + 0.1 * _randn(d, generator=g)
    W[name + ".bias"] = 0.1 * _randn(d, generator=g)


def clip_weights(cfg, seed=777):
    """fp32 CPU weights keyed like transformers CLIPModel.state_dict() (PickScore_v1 architecture)."""
    g = _gen(seed)
    W = {"logit_scale": torch.tensor(math.log(100.0), device=_DEVICE)}

    def layers(pfx, n, d, mlp):
        for i in range(n):
            p = f"{pfx}.encoder.layers.{i}"
            _ln(W, f"{p}.layer_norm1", d, g); _ln(W, f"{p}.layer_norm2", d, g)
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                linear_(W, f"{p}.self_attn.{nm}", d, d, g)
            linear_(W, f"{p}.mlp.fc1", mlp, d, g)
            linear_(W, f"{p}.mlp.fc2", d, mlp, g)
    v = "vision_model"
    W[f"{v}.embeddings.patch_embedding.weight"] = _randn(cfg.v_hidden, 3, cfg.patch, cfg.patch, generator=g) / math.sqrt(3 * cfg.patch ** 2)
    W[f"{v}.embeddings.class_embedding"] = _randn(cfg.v_hidden, generator=g) * 0.5
    npos = (cfg.image_size // cfg.patch) ** 2 + 1
    W[f"{v}.embeddings.position_embedding.weight"] = _randn(npos, cfg.v_hidden, generator=g) * 0.5
    _ln(W, f"{v}.pre_layrnorm", cfg.v_hidden, g)
    layers(v, cfg.v_layers, cfg.v_hidden, cfg.v_mlp)
    _ln(W, f"{v}.post_layernorm", cfg.v_hidden, g)
    linear_(W, "visual_projection", cfg.proj, cfg.v_hidden, g, bias=False)
    t = "text_model"
    W[f"{t}.embeddings.token_embedding.weight"] = _randn(cfg.vocab, cfg.t_hidden, generator=g) * 0.5
    W[f"{t}.embeddings.position_embedding.weight"] = _randn(cfg.max_pos, cfg.t_hidden, generator=g) * 0.5
    layers(t, cfg.t_layers, cfg.t_hidden, cfg.t_mlp)
    _ln(W, f"{t}.final_layer_norm", cfg.t_hidden, g)
    linear_(W, "text_projection", cfg.proj, cfg.t_hidden, g, bias=False)
    return W


def dino_weights(cfg, seed=888):
    """fp32 CPU weights keyed like timm vit_base_patch14_dinov2 state_dict()."""
    g = _gen(seed)
    D = cfg.hidden
    W = {"patch_embed.proj.weight": _randn(D, 3, cfg.patch, cfg.patch, generator=g) / math.sqrt(3 * cfg.patch ** 2),
         "patch_embed.proj.bias": _randn(D, generator=g) * 0.1,
         "cls_token": _randn(1, 1, D, generator=g) * 0.5,
         "pos_embed": _randn(1, (cfg.image_size // cfg.patch) ** 2 + 1, D, generator=g) * 0.5}
    for i in range(cfg.layers):
        p = f"blocks.{i}"
        _ln(W, f"{p}.norm1", D, g); _ln(W, f"{p}.norm2", D, g)
        linear_(W, f"{p}.attn.qkv", 3 * D, D, g); linear_(W, f"{p}.attn.proj", D, D, g)
        linear_(W, f"{p}.mlp.fc1", cfg.mlp, D, g); linear_(W, f"{p}.mlp.fc2", D, cfg.mlp, g)
        W[f"{p}.ls1.gamma"] = 0.3 + 0.1 * _randn(D, generator=g)
        W[f"{p}.ls2.gamma"] = 0.3 + 0.1 * _randn(D, generator=g)
    _ln(W, "norm", D, g)
    return W


def dino_head_weights(in_dim=768, hidden=512, seed=999):
    """DINOHead (train_sd3_fast_dino_patch.py:592-603): layers.0 Linear(in,512), layers.2 Linear(512,1)."""
    g = _gen(seed)
    W = {}
    linear_(W, "layers.0", hidden, in_dim, g, std=1.0)      # inputs are unit vectors: keep logits O(1)
    linear_(W, "layers.2", 1, hidden, g)
    return W


def clip_input_ids(n, seed=3, vocab=49408, eos=49407, length=77):
    """Synthetic CLIP token ids: random tokens, EOS at a random position >= 8, pad (= EOS) after."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, eos - 1, (n, length), generator=g)
    pos = torch.randint(8, length, (n,), generator=g)
    for i in range(n):
        ids[i, pos[i]:] = eos
    return ids
