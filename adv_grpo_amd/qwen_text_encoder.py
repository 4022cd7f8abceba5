"""Qwen2.5-VL language model on text-only input: the prompt encoder of Qwen-Image (BASELINE config 5), on the gfx950 kernels.

The Qwen-Image twin of ``encode_prompt`` (scripts/train_dreambooth_lora_sd3.py:98-144 as used at TP:628-651; the reference names the
config at config/grpo.py:324,330 and ships no Qwen-Image code, README.md:75): what QwenImagePipeline._get_qwen_prompt_embeds computes
after tokenisation (tokenizers stay host code outside the path, as for SD3) -- ``hidden_states[-1]`` of the 28-layer decoder, the first 34
(template) tokens dropped, the batch right-padded.  Weights: transformers' text-model state dict (``layers.N.self_attn.q_proj.weight``
...).  Oracle: oracle/qwen_text.py, PINNED against the installed transformers (tests/test_oracle_qwen_text.py).

It runs once per prompt (as the SD3 towers do: 3.6 % of a rollout step there), so it is assembled from the existing kernels with two small
new ones: RMSNorm rows, fused q|k|v Linear, `advgrpo_rope_half` (rotate_half rotary on the 28 query + 4 key heads in place), causal
grouped-query attention over MATERIALISED scores (prompts are <= 512 tokens: batched GEMMs on head-major copies +
`advgrpo_softmax_rows_causal`; right padding needs no extra mask -- a valid query only sees keys before it), SiLU-gated feed-forward
through the `silu` and `mul_aux` GEMM epilogues."""
import torch

from . import _lib, ops
from .vit import _bf

DROP_IDX = 34      # QwenImagePipeline.prompt_template_encode_start_idx


class Qwen25VLTextEncoder:
    def __init__(self, sd, cfg, device="cuda"):
        self.cfg, self.device = cfg, torch.device(device)
        dev = self.device
        self.embed = _bf(sd["embed_tokens.weight"], dev)
        self.final_w = _bf(sd["norm.weight"], dev)
        self.layers = []
        for i in range(cfg.num_layers):
            p, a = f"layers.{i}", f"layers.{i}.self_attn"
            self.layers.append({
                "ln1": _bf(sd[f"{p}.input_layernorm.weight"], dev), "ln2": _bf(sd[f"{p}.post_attention_layernorm.weight"], dev),
                "qkv.w": _bf(torch.cat([sd[f"{a}.q_proj.weight"], sd[f"{a}.k_proj.weight"], sd[f"{a}.v_proj.weight"]]), dev),
                "qkv.b": _bf(torch.cat([sd[f"{a}.q_proj.bias"], sd[f"{a}.k_proj.bias"], sd[f"{a}.v_proj.bias"]]), dev),
                "o": _bf(sd[f"{a}.o_proj.weight"], dev), "gate": _bf(sd[f"{p}.mlp.gate_proj.weight"], dev),
                "up": _bf(sd[f"{p}.mlp.up_proj.weight"], dev), "down": _bf(sd[f"{p}.mlp.down_proj.weight"], dev)})
        self._cs = {}

    def _cos_sin(self, T):
        def make():
            cfg = self.cfg
            inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float32) / cfg.head_dim))
            ang = torch.outer(torch.arange(T, dtype=torch.float32), inv)
            return torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().to(self.device)       # [T, hd / 2, 2]
        return ops.cached(self._cs, T, make)

    def _attention(self, qkv, B, T):
        """Causal grouped-query attention over materialised scores.  qkv [B*T, (H + 2 KV) hd] (q | k | v, rotary applied) -> [B*T, H hd]."""
        cfg, lib = self.cfg, _lib.load()
        H, KV, hd = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
        Tp = (T + 63) // 64 * 64
        bf16 = torch.bfloat16
        x = qkv.view(B, T, H + 2 * KV, hd)
        q = torch.zeros(B * H, Tp, hd, dtype=bf16, device=qkv.device)
        q[:, :T] = x[:, :, :H].permute(0, 2, 1, 3).reshape(B * H, T, hd)
        k = torch.zeros(B * H, Tp, hd, dtype=bf16, device=qkv.device)
        k[:, :T] = x[:, :, H:H + KV].permute(0, 2, 1, 3).repeat_interleave(H // KV, dim=1).reshape(B * H, T, hd)
        vt = torch.zeros(B * H, hd, Tp, dtype=bf16, device=qkv.device)
        vt[:, :, :T] = x[:, :, H + KV:].permute(0, 2, 3, 1).repeat_interleave(H // KV, dim=1).reshape(B * H, hd, T)
        sc = ops.bmm_nt(q, k, out_dtype=torch.float32, alpha=hd ** -0.5)                     # [BH, Tp, Tp]
        p16 = torch.empty(B * H, Tp, Tp, dtype=bf16, device=qkv.device)
        _lib.check(lib.advgrpo_softmax_rows_causal(sc.data_ptr(), p16.data_ptr(), B * H * Tp, Tp, _lib.stream_ptr()))
        o = ops.bmm_nt(p16, vt)                                                              # [BH, Tp, hd]
        return o[:, :T].reshape(B, H, T, hd).permute(0, 2, 1, 3).reshape(B * T, H * hd).contiguous()

    @torch.no_grad()
    def __call__(self, input_ids, attention_mask=None):
        """-> hidden_states[-1] [B, T, hidden] bf16.  (Right padding needs no mask: see the module docstring; the rows of padding
        tokens are computed and meaningless, as in transformers.)"""
        cfg, lib = self.cfg, _lib.load()
        B, T = input_ids.shape
        if T > 512:
            raise ValueError(f"Qwen25VLTextEncoder: prompts of up to 512 tokens (got {T})")
        H, KV, hd, D = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim, cfg.hidden_size
        x = self.embed[input_ids.to(self.device)].reshape(B * T, D).contiguous()
        cs = self._cos_sin(T)
        for L in self.layers:
            h = ops.rmsnorm_rows(x, L["ln1"], eps=cfg.rms_eps)
            qkv = ops.gemm(h, L["qkv.w"], bias=L["qkv.b"])
            _lib.check(lib.advgrpo_rope_half(qkv.data_ptr(), qkv.stride(0), B * T, T, 0, H + KV, hd, cs.data_ptr(), _lib.stream_ptr()))
            o = self._attention(qkv, B, T)
            ops.gemm(o, L["o"], residual=x, out=x)
            h = ops.rmsnorm_rows(x, L["ln2"], eps=cfg.rms_eps)
            g = ops.gemm(h, L["gate"], act="silu")
            u = ops.gemm_train(h, L["up"], act="mul_aux", aux_in=g)                          # silu(h Wg^T) * (h Wu^T)
            ops.gemm(u, L["down"], residual=x, out=x)
        return ops.rmsnorm_rows(x, self.final_w, eps=cfg.rms_eps).view(B, T, D)

    @torch.no_grad()
    def encode_prompt(self, input_ids, attention_mask, drop_idx=DROP_IDX):
        """_get_qwen_prompt_embeds after tokenisation: (prompt_embeds [B, Lmax, hidden] bf16, mask [B, Lmax]) -- each sample's valid tokens
        minus the first drop_idx template tokens, right-padded with zeros.  The lengths come from the HOST copy of the mask (the
        tokenizer's output): no device synchronisation."""
        hs = self(input_ids, attention_mask)
        lens = [int(n) for n in attention_mask.cpu().sum(dim=1)]
        L = max(n - drop_idx for n in lens)
        emb = torch.zeros(hs.shape[0], L, hs.shape[2], dtype=hs.dtype, device=hs.device)
        msk = torch.zeros(hs.shape[0], L, dtype=torch.long, device=hs.device)
        for b, n in enumerate(lens):
            emb[b, :n - drop_idx] = hs[b, drop_idx:n]
            msk[b, :n - drop_idx] = 1
        return emb, msk
