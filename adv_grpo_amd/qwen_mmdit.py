"""Qwen-Image MMDiT forward on the gfx950 kernels (host orchestration only) -- BASELINE config 5's transformer.

The reference has no Qwen-Image trainer: config 5 exists there as a to-do (README.md:75 "Try more base models like
QWen-Image") and as the `pretrained.model` / `resolution` switches of config/grpo.py:324,330.  This class stands in for
diffusers' ``QwenImageTransformer2DModel`` at the same call sites the SD3 model serves
(adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:630-637, scripts/train_sd3_fast_pickscore.py:235-255) and keeps
the call signature the rollout uses there -- ``transformer(hidden_states, timestep, encoder_hidden_states,
pooled_projections, ..., return_dict=False)[0]`` -- so that ``pipeline_with_logprob_random`` drives it unchanged:
  * ``hidden_states`` are the UNPACKED 16-channel latents [B, 16, h, w]; the 2x2 packing of QwenImagePipeline._pack_latents and
    its inverse happen here (patchify kernel; proj_out rows re-ordered once at load time);
  * ``timestep`` is on the scheduler's 0..1000 scale (the SD3 convention of the rollout); QwenImagePipeline hands the model
    timestep / 1000 and the model's Timesteps(scale=1000) multiplies it back: the same sinusoid argument;
  * ``pooled_projections`` is accepted and ignored (Qwen-Image conditions on the timestep only).
Weights are loaded from a diffusers-named state dict (oracle/qwen_mmdit.py has the restated architecture, PARITY UNPINNED).
Assumption: every prompt of a batch -- the negative and the positive half of the CFG batch included -- is handed over at ONE text length
N_txt and no ``encoder_hidden_states_mask`` is applied: QwenImagePipeline pads to the longest prompt and masks the padding keys; here the caller
(qwen_text_encoder.py, the trainer's data path) pads with the tokenizer's pad token and the padding positions attend and are attended like any other token.
scripts/verify_against_diffusers.py pins the restatement against diffusers' masked forward on ragged lengths by running the shorter prompt unpadded here
(what the mask makes of it there).

Data layout in HBM (bf16), B = batch incl. the CFG halves, S = N_img + N_txt, D = 3072, 24 heads x 128:
  x [B*N_img, D], c [B*N_txt, D] residual streams, updated in place by the out-projection / FF2 epilogues
  qkv [B*S, 3D]  joint packed q|k|v, IMAGE rows first, then text rows, per sample (diffusers concatenates [text ; image];
                 attention without a mask does not see the order of its keys) -- the two QKV GEMMs scatter into it, one
                 launch applies the per-head RMSNorm(128) and the rotary embedding in place, attention reads the slices
  mods [Bm, 722*D]  all 60 x 12 + 2 modulation vectors of a forward from chunked skinny GEMMs (13.6 GB of weights; a rollout
                 computes them once for all timesteps, precompute_mods); Bm = 1 when every sample shares the timestep
fp8 (enable_fp8, the arithmetic BASELINE config 5 names): the block Linears on e4m3 operands exactly like the SD3 model's
fp8 mode (mmdit.py): LayerNorms write the e4m3 rows, attention / GELU outputs pass through the row quantiser.
"""
import math

import torch

from . import ops


class QwenImageTransformer2DModel:
    FP8_LINEARS = ("qkv", "cqkv", "out", "cout", "ff1", "cff1", "ff2", "cff2")
    MOD_CHUNK_ELEMS = 1 << 30           # weights of one modulation GEMM launch (elements): stays inside 32-bit offsets

    def __init__(self, state_dict, cfg, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.config = type("Cfg", (), {"in_channels": cfg.out_channels})()     # what prepare_latents draws: unpacked latents
        self._rope_cache = {}
        self.fp8 = None
        self._prepare(state_dict)

    # ------------------------------------------------------------------ weight preparation
    def _prepare(self, sd):
        cfg, dev = self.cfg, self.device
        D = cfg.dim
        bf = lambda t: t.to(device=dev, dtype=torch.bfloat16).contiguous()
        take = lambda k: sd.pop(k) if isinstance(sd, dict) else sd[k]       # (a dict is consumed: 41 GB of weights at full size)
        w = {}
        for name in ("img_in", "txt_in", "time_text_embed.timestep_embedder.linear_1", "time_text_embed.timestep_embedder.linear_2"):
            w[name + ".w"], w[name + ".b"] = bf(take(name + ".weight")), bf(take(name + ".bias"))
        w["txt_norm.w"] = bf(take("txt_norm.weight"))
        # proj_out rows are (c, py, px) (QwenImagePipeline._unpack_latents); the unpatchify kernel reads (py, px, c)
        C, ps = cfg.out_channels, cfg.patch_size
        pw, pb = take("proj_out.weight"), take("proj_out.bias")
        perm = torch.arange(C * ps * ps).view(C, ps * ps).t().reshape(-1).to(pw.device)      # new row pq * C + c <- old row c * 4 + pq
        w["proj_out.w"], w["proj_out.b"] = bf(pw[perm]), bf(pb[perm])
        self.w = w
        # modulation Linears (img_mod.1 / txt_mod.1 of every block, norm_out.linear) concatenated into a few matrices of at most
        # MOD_CHUNK_ELEMS elements: (weight [rows, D], bias [rows], first output column)
        self.mod_off, self.mod_chunks, self.blocks = {}, [], []
        cur = {"w": [], "b": [], "first": 0}
        off = 0

        def flush():
            if cur["w"]:
                self.mod_chunks.append((bf(torch.cat(cur["w"])), bf(torch.cat(cur["b"])), cur["first"]))
            cur["w"], cur["b"], cur["first"] = [], [], off

        def add_mod(key, name):
            nonlocal off
            wt, bt = take(name + ".weight"), take(name + ".bias")
            if sum(t.numel() for t in cur["w"]) + wt.numel() > self.MOD_CHUNK_ELEMS:
                flush()
            cur["w"].append(wt); cur["b"].append(bt)
            self.mod_off[key] = off
            off += wt.shape[0]
        for i in range(cfg.num_layers):
            p = f"transformer_blocks.{i}"
            add_mod(("x", i), f"{p}.img_mod.1")
            add_mod(("c", i), f"{p}.txt_mod.1")
            b = {}
            cat = lambda names: (bf(torch.cat([take(f"{p}.{n}.weight") for n in names])),
                                 bf(torch.cat([take(f"{p}.{n}.bias") for n in names])))
            b["qkv.w"], b["qkv.b"] = cat(["attn.to_q", "attn.to_k", "attn.to_v"])
            b["cqkv.w"], b["cqkv.b"] = cat(["attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj"])
            b["rms_x"] = bf(torch.stack([take(f"{p}.attn.norm_q.weight"), take(f"{p}.attn.norm_k.weight")]))
            b["rms_c"] = bf(torch.stack([take(f"{p}.attn.norm_added_q.weight"), take(f"{p}.attn.norm_added_k.weight")]))
            for n, k in (("attn.to_out.0", "out"), ("attn.to_add_out", "cout"), ("img_mlp.net.0.proj", "ff1"), ("img_mlp.net.2", "ff2"),
                         ("txt_mlp.net.0.proj", "cff1"), ("txt_mlp.net.2", "cff2")):
                b[k + ".w"], b[k + ".b"] = bf(take(f"{p}.{n}.weight")), bf(take(f"{p}.{n}.bias"))
            self.blocks.append(b)
        add_mod(("out",), "norm_out.linear")
        flush()
        self.n_mod = off

    # ------------------------------------------------------------------ fp8 Linears (the arithmetic BASELINE config 5 names)
    def enable_fp8(self):
        """The block Linears (QKV / out-projection / feed-forward of both streams) on fp8 e4m3 operands: scheme of
        include/advgrpo.h ("fp8 Linears"), identical to SD3Transformer2DModel.enable_fp8 (mmdit.py)."""
        self.fp8 = {}
        self.requantize()

    @torch.no_grad()
    def requantize(self):
        for i, b in enumerate(self.blocks):
            for key in self.FP8_LINEARS:
                self.fp8[(i, key)] = ops.quant_fp8_rows(b[key + ".w"], out=self.fp8.get((i, key)))

    # ------------------------------------------------------------------ rotary table
    def _rope(self, hh, ww, Nt):
        """[S, head_dim] f32, (cos, sin) of pair i at [s, 2i], [s, 2i+1]; image positions (hh x ww packed-latent grid, one
        frame) first, then the Nt text positions -- diffusers' QwenEmbedRope (scale_rope: centred spatial axes, text positions
        after the largest half-extent), restated in oracle/qwen_mmdit.py:rope_freqs."""
        def make():
            cfg = self.cfg
            ax, theta = cfg.axes_dims_rope, cfg.rope_theta
            def ang(index, dim):
                inv = 1.0 / torch.pow(torch.tensor(theta, dtype=torch.float32), torch.arange(0, dim, 2).float().div(dim))
                return torch.outer(index.float(), inv)
            if cfg.scale_rope:
                ih = torch.cat([torch.arange(-(hh - hh // 2), 0), torch.arange(hh // 2)])
                iw = torch.cat([torch.arange(-(ww - ww // 2), 0), torch.arange(ww // 2)])
                t0 = max(hh // 2, ww // 2)
            else:
                ih, iw, t0 = torch.arange(hh), torch.arange(ww), max(hh, ww)
            a_f = ang(torch.zeros(1), ax[0]).view(1, 1, -1).expand(hh, ww, -1)
            a_h = ang(ih, ax[1]).view(hh, 1, -1).expand(hh, ww, -1)
            a_w = ang(iw, ax[2]).view(1, ww, -1).expand(hh, ww, -1)
            img = torch.cat([a_f, a_h, a_w], dim=-1).reshape(hh * ww, -1)
            it = torch.arange(t0, t0 + Nt)
            txt = torch.cat([ang(it, d) for d in ax], dim=1)
            a = torch.cat([img, txt], dim=0)                                   # [S, head_dim / 2] angles
            return torch.stack([torch.cos(a), torch.sin(a)], dim=-1).reshape(a.shape[0], -1).contiguous().to(self.device)
        return ops.cached(self._rope_cache, (hh, ww, Nt), make)

    # ------------------------------------------------------------------ forward pieces
    def _temb(self, timestep):
        """timestep_embedder(Timesteps(sigma * 1000)) -> [Bt, D] bf16; `timestep` on the 0..1000 scale."""
        w = self.w
        t1, t2 = "time_text_embed.timestep_embedder.linear_1", "time_text_embed.timestep_embedder.linear_2"
        return ops.gemm(ops.gemm(ops.timestep_embedding(timestep), w[t1 + ".w"], bias=w[t1 + ".b"], act="silu"),
                        w[t2 + ".w"], bias=w[t2 + ".b"])

    def _mods(self, temb):
        s = ops.unary(temb, "silu")
        mods = torch.empty(temb.shape[0], self.n_mod, dtype=torch.bfloat16, device=temb.device)
        for wt, bt, first in self.mod_chunks:
            ops.gemm(s, wt, bias=bt, out=mods[:, first:first + wt.shape[0]])
        return mods

    @torch.no_grad()
    def embed_context(self, encoder_hidden_states):
        """txt_in(txt_norm(prompt embeddings)) [B, Nt, 3584] -> [B * Nt, D]: timestep-free, computed once per rollout."""
        B, Nt = encoder_hidden_states.shape[:2]
        e = encoder_hidden_states.to(torch.bfloat16).reshape(B * Nt, -1).contiguous()
        return ops.gemm(ops.rmsnorm_rows(e, self.w["txt_norm.w"], eps=1e-6), self.w["txt_in.w"], bias=self.w["txt_in.b"])

    @torch.no_grad()
    def precompute_mods(self, timesteps, pooled_projections=None):
        """The modulation rows of a whole rollout (one row per timestep: Qwen-Image's conditioning vector is the timestep
        embedding alone, the same for every sample) -> [T, 1, n_mod]; pass mods=result[i] to __call__."""
        return self._mods(self._temb(timesteps)).unsqueeze(1)

    @torch.no_grad()
    def __call__(self, hidden_states, timestep, encoder_hidden_states, pooled_projections=None, joint_attention_kwargs=None,
                 return_dict=False, out_dtype=None, return_intermediates=False, mods=None, context=None):
        cfg, w = self.cfg, self.w
        D, H, hd = cfg.dim, cfg.num_heads, cfg.head_dim
        B, C, h, wd = hidden_states.shape
        hh, ww = h // cfg.patch_size, wd // cfg.patch_size
        Ni, Nt = hh * ww, encoder_hidden_states.shape[1]
        S = Ni + Nt
        dev, bf16 = hidden_states.device, torch.bfloat16
        inter = {}

        x = ops.gemm(ops.patchify(hidden_states.contiguous()), w["img_in.w"], bias=w["img_in.b"])
        temb = None
        if mods is None or return_intermediates:
            # every sample of a rollout step shares the timestep (t.expand(B): stride 0): one conditioning row serves the batch
            t_rows = timestep[:1] if (timestep.dim() == 0 or timestep.numel() == 1 or timestep.stride(0) == 0) else timestep
            temb = self._temb(t_rows.reshape(-1))
        if mods is None:
            mods = self._mods(temb)
        mods = mods.expand(B, -1) if mods.shape[0] == 1 else mods
        c = context.clone() if context is not None else self.embed_context(encoder_hidden_states)
        if return_intermediates:
            inter.update(x0=x.view(B, Ni, D).clone(), c0=c.view(B, Nt, D).clone(), temb=temb.expand(B, -1).clone())
        rope = self._rope(hh, ww, Nt)

        def mod(key, j):
            o = self.mod_off[key] + j * D
            return mods[:, o:o + D]

        qkv = torch.empty(B * S, 3 * D, dtype=bf16, device=dev)
        qkv3 = qkv.view(B, S, 3 * D)
        att = torch.empty(B, S, D, dtype=bf16, device=dev)
        att2d = att.view(B * S, D)
        Mi, Mt = B * Ni, B * Nt
        f8 = self.fp8
        if f8 is not None:
            q_n = ops.Fp8Rows(torch.empty(Mi + Mt, D, dtype=torch.uint8, device=dev), torch.empty(Mi + Mt, dtype=torch.float32, device=dev))
            q_h = ops.Fp8Rows(torch.empty(Mi + Mt, 4 * D, dtype=torch.uint8, device=dev), torch.empty(Mi + Mt, dtype=torch.float32, device=dev))
            h_all = torch.empty(Mi + Mt, 4 * D, dtype=bf16, device=dev)
        else:
            nx_buf = torch.empty(Mi, D, dtype=bf16, device=dev)
            nc_buf = torch.empty(Mt, D, dtype=bf16, device=dev)

        def linears(i, b, items):
            """One grouped launch (image-stream Linear + its text-stream twin): items = (input, Linear key, epilogue kwargs)."""
            if f8 is not None:
                return ops.gemm_grouped_fp8([ops.gemm_desc_fp8(a, f8[(i, key)], bias=b[key + ".b"], **kw) for a, key, kw in items])
            return ops.gemm_grouped([ops.gemm_desc(a, b[key + ".w"], bias=b[key + ".b"], **kw) for a, key, kw in items])

        for i, b in enumerate(self.blocks):
            kx, kc = ("x", i), ("c", i)
            # --- norm1 + modulation, both streams; chunks of img_mod / txt_mod: (shift, scale, gate) x (attention, MLP)
            if f8 is not None:
                nx, nc = q_n.rows(0, Mi), q_n.rows(Mi, Mi + Mt)
                ops.layernorm_mod_fp8(x, nx, scale=mod(kx, 1), shift=mod(kx, 0), rows_per_batch=Ni)
                ops.layernorm_mod_fp8(c, nc, scale=mod(kc, 1), shift=mod(kc, 0), rows_per_batch=Nt)
            else:
                nx = ops.layernorm_mod(x, out=nx_buf, scale=mod(kx, 1), shift=mod(kx, 0), rows_per_batch=Ni)
                nc = ops.layernorm_mod(c, out=nc_buf, scale=mod(kc, 1), shift=mod(kc, 0), rows_per_batch=Nt)
            # --- joint attention: fused QKV projections scattered into the joint buffer, QK-norm + rotary in place, attention
            linears(i, b, [(nx, "qkv", dict(out=qkv, seg=(Ni, S, 0))), (nc, "cqkv", dict(out=qkv, seg=(Nt, S, Ni)))])
            ops.qk_norm_rope(qkv, S, Ni, 2 * H, hd, b["rms_x"], b["rms_c"], H, rope=rope, eps=1e-6)
            ops.attention(qkv3[:, :, :D], qkv3[:, :, D:2 * D], qkv3[:, :, 2 * D:], H, out=att)
            if f8 is not None:
                ops.quant_fp8_rows(att2d, out=q_n, split=(Ni, S))
                outs = [(q_n.rows(0, Mi), "out", dict(gate=mod(kx, 2), gate_rows=Ni, residual=x, out=x)),
                        (q_n.rows(Mi, Mi + Mt), "cout", dict(gate=mod(kc, 2), gate_rows=Nt, residual=c, out=c))]
            else:
                outs = [(att2d, "out", dict(gate=mod(kx, 2), gate_rows=Ni, residual=x, out=x, a_seg=(Ni, S, 0), M=Mi)),
                        (att2d, "cout", dict(gate=mod(kc, 2), gate_rows=Nt, residual=c, out=c, a_seg=(Nt, S, Ni), M=Mt))]
            linears(i, b, outs)
            # --- MLPs
            if f8 is not None:
                ops.layernorm_mod_fp8(x, q_n.rows(0, Mi), scale=mod(kx, 4), shift=mod(kx, 3), rows_per_batch=Ni)
                ops.layernorm_mod_fp8(c, q_n.rows(Mi, Mi + Mt), scale=mod(kc, 4), shift=mod(kc, 3), rows_per_batch=Nt)
                linears(i, b, [(q_n.rows(0, Mi), "ff1", dict(act="gelu_tanh", out=h_all[:Mi])),
                               (q_n.rows(Mi, Mi + Mt), "cff1", dict(act="gelu_tanh", out=h_all[Mi:]))])
                ops.quant_fp8_rows(h_all, out=q_h)
                hm = [q_h.rows(0, Mi), q_h.rows(Mi, Mi + Mt)]
            else:
                nx = ops.layernorm_mod(x, out=nx_buf, scale=mod(kx, 4), shift=mod(kx, 3), rows_per_batch=Ni)
                nc = ops.layernorm_mod(c, out=nc_buf, scale=mod(kc, 4), shift=mod(kc, 3), rows_per_batch=Nt)
                hm = linears(i, b, [(nx, "ff1", dict(act="gelu_tanh")), (nc, "cff1", dict(act="gelu_tanh"))])
            linears(i, b, [(hm[0], "ff2", dict(gate=mod(kx, 5), gate_rows=Ni, residual=x, out=x)),
                           (hm[1], "cff2", dict(gate=mod(kc, 5), gate_rows=Nt, residual=c, out=c))])
            if return_intermediates:
                inter[f"x{i + 1}"] = x.view(B, Ni, D).clone()
                inter[f"c{i + 1}"] = c.view(B, Nt, D).clone()
        nx = ops.layernorm_mod(x, scale=mod(("out",), 0), shift=mod(("out",), 1), rows_per_batch=Ni)
        tok = ops.gemm(nx, w["proj_out.w"], bias=w["proj_out.b"])
        out = ops.unpatchify(tok, B, cfg.out_channels, h, wd, out_dtype or bf16)
        if return_intermediates:
            return (out,), inter
        return (out,)


def flops_per_sample_forward(cfg, n_img, n_txt):
    """Algorithmic FLOPs of one sample's forward (the convention of SURVEY.md 8d: 2 x MACs of the Linears + 4 S^2 D per
    attention layer; embedders / modulation / norms not counted)."""
    D, L, S = cfg.dim, cfg.num_layers, n_img + n_txt
    return 2.0 * 12 * D * D * S * L + 4.0 * D * S * S * L
