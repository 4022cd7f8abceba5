"""Prompt encoding on the gfx950 kernels: the three text encoders behind
``compute_text_embeddings`` (scripts/train_sd3_fast_pickscore.py:186-193) -> ``encode_prompt``
(adv_grpo/diffusers_patch/train_dreambooth_lora_sd3.py:98-144):

  CLIP-L and CLIP-G  (CLIPTextModelWithProjection: penultimate hidden state + projected pooled state)
  T5-XXL v1.1        (T5EncoderModel: RMSNorm, un-scaled attention + relative position bias, gated GELU)

Weights are given as transformers state dicts (same key names); token ids come from the caller (tokenizers are host
code outside the accelerated path).  Same GEMM / attention / norm kernels as the MMDiT; what is new for T5 is
``advgrpo_rmsnorm_rows``, the additive score bias of ``advgrpo_attention_fwd_bias`` and the ``mul_aux`` GEMM epilogue
(gate of the gated-GELU feed-forward)."""
import torch

from . import ops
from .vit import _bf, _Encoder, pack_clip_layers


class CLIPTextEncoder:
    def __init__(self, sd, n_layers, heads, act, eos_token_id, device="cuda"):
        dev = torch.device(device)
        self.device, self.eos = dev, eos_token_id
        t = "text_model"
        layers = pack_clip_layers(sd, t, n_layers, dev)
        self.body = _Encoder(layers[:-1], heads, 1e-5, act, causal=True)
        self.last = _Encoder(layers[-1:], heads, 1e-5, act, causal=True)
        self.tok_emb = _bf(sd[f"{t}.embeddings.token_embedding.weight"], dev)
        self.pos = _bf(sd[f"{t}.embeddings.position_embedding.weight"], dev)
        self.final_ln = (_bf(sd[f"{t}.final_layer_norm.weight"], dev), _bf(sd[f"{t}.final_layer_norm.bias"], dev))
        self.proj = _bf(sd["text_projection.weight"], dev)

    @torch.no_grad()
    def __call__(self, input_ids):
        """-> (hidden_states[-2] [B,S,D], text_embeds [B,P]) as _encode_prompt_with_clip takes them (TD3:84-88)."""
        ids = input_ids.to(self.device)
        B, S = ids.shape
        D = self.tok_emb.shape[1]
        x = (self.tok_emb[ids] + self.pos[:S][None]).reshape(B * S, D).contiguous()   # embedding gather: index plumbing
        x = self.body(x, B, S)
        pen = x.view(B, S, D).clone()
        x = self.last(x, B, S)
        eos = ids.int().argmax(dim=-1) if self.eos == 2 else (ids == self.eos).int().argmax(dim=-1)
        pooled = x.view(B, S, D)[torch.arange(B, device=self.device), eos].contiguous()
        pooled = ops.layernorm_mod(pooled, w=self.final_ln[0], b=self.final_ln[1], eps=1e-5)
        return pen, ops.gemm(pooled, self.proj)


class T5Encoder:
    def __init__(self, sd, n_layers, heads, d_kv=64, num_buckets=32, max_distance=128, device="cuda"):
        assert d_kv == 64, "the attention kernels serve head dim 64"
        dev = torch.device(device)
        self.device, self.H, self.nb, self.md = dev, heads, num_buckets, max_distance
        self.emb = _bf(sd["shared.weight"], dev)
        self.rel = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].to(dev, torch.bfloat16).float()
        self.final_w = _bf(sd["encoder.final_layer_norm.weight"], dev)
        self.blocks = []
        for i in range(n_layers):
            p = f"encoder.block.{i}.layer"
            a, f = f"{p}.0.SelfAttention", f"{p}.1.DenseReluDense"
            self.blocks.append({
                "ln1": _bf(sd[f"{p}.0.layer_norm.weight"], dev), "ln2": _bf(sd[f"{p}.1.layer_norm.weight"], dev),
                "qkv": _bf(torch.cat([sd[f"{a}.q.weight"], sd[f"{a}.k.weight"], sd[f"{a}.v.weight"]]), dev),
                "o": _bf(sd[f"{a}.o.weight"], dev), "wi0": _bf(sd[f"{f}.wi_0.weight"], dev),
                "wi1": _bf(sd[f"{f}.wi_1.weight"], dev), "wo": _bf(sd[f"{f}.wo.weight"], dev)})
        self._bias = {}

    def position_bias(self, S):
        """[H,S,S] f32: T5Attention.compute_bias for the bidirectional encoder (bucket arithmetic = index plumbing)."""
        if S not in self._bias:
            import math
            rel = torch.arange(S, device=self.device)[None, :] - torch.arange(S, device=self.device)[:, None]
            nb = self.nb // 2
            ret = (rel > 0).long() * nb
            n = rel.abs()
            max_exact = nb // 2
            large = max_exact + (torch.log(n.float() / max_exact) / math.log(self.md / max_exact) * (nb - max_exact)).long()
            large = torch.min(large, torch.full_like(large, nb - 1))
            bucket = ret + torch.where(n < max_exact, n, large)
            self._bias[S] = self.rel[bucket].permute(2, 0, 1).contiguous()
        return self._bias[S]

    @torch.no_grad()
    def __call__(self, input_ids):
        ids = input_ids.to(self.device)
        B, S = ids.shape
        D = self.emb.shape[1]
        inner = self.H * 64
        x = self.emb[ids].reshape(B * S, D).contiguous()
        bias = self.position_bias(S)
        for L in self.blocks:
            h = ops.rmsnorm_rows(x, L["ln1"])
            qkv = ops.gemm(h, L["qkv"]).view(B, S, 3 * inner)
            o = ops.attention_bias(qkv[:, :, :inner], qkv[:, :, inner:2 * inner], qkv[:, :, 2 * inner:], self.H, bias, scale=1.0)
            ops.gemm(o.view(B * S, inner), L["o"], residual=x, out=x)
            h = ops.rmsnorm_rows(x, L["ln2"])
            g = ops.gemm(h, L["wi0"], act="gelu_tanh")
            u = ops.gemm_train(h, L["wi1"], act="mul_aux", aux_in=g)                  # gelu(h W0^T) * (h W1^T)
            ops.gemm(u, L["wo"], residual=x, out=x)
        return ops.rmsnorm_rows(x, self.final_w).view(B, S, D)


@torch.no_grad()
def encode_prompt(clip_l, clip_g, t5, ids_l, ids_g, ids_t5):
    """encode_prompt (TD3:98-144) from token ids: CLIP-L (+) CLIP-G penultimate states side by side, zero-padded to the T5
    width, followed by the T5 states along the sequence; pooled = the two projected pooled states side by side."""
    pl, pooled_l = clip_l(ids_l)
    pg, pooled_g = clip_g(ids_g)
    t5e = t5(ids_t5)
    B, S, _ = pl.shape
    out = torch.zeros(B, S + t5e.shape[1], t5e.shape[2], dtype=torch.bfloat16, device=t5e.device)
    out[:, :S, :pl.shape[2]] = pl
    out[:, :S, pl.shape[2]:pl.shape[2] + pg.shape[2]] = pg
    out[:, S:] = t5e
    return out, torch.cat([pooled_l, pooled_g], dim=-1)
