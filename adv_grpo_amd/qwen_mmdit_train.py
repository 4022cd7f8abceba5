"""LoRA training of the Qwen-Image MMDiT on the gfx950 kernels: the update half (G-step) of BASELINE config 5.

The reference has no Qwen-Image code (README.md:75, config/grpo.py:324,330); this is the Qwen-Image twin of what
mmdit_train.py replaces for SD3 (scripts/train_sd3_fast_pickscore.py:1077-1187: peft LoRA r=32 / alpha=64 on the attention
projections attn.{to_q,to_k,to_v,to_out.0,add_q_proj,add_k_proj,add_v_proj,to_add_out} (TP:490-511), autograd through the
transformer call of compute_log_prob (TP:233-267), clip_grad_norm_ + AdamW (TP:1165-1171), EMAModuleWrapper), with the same
flat parameter / gradient / moment vectors, merged-LoRA forward, token-contracted adapter-gradient GEMMs (on the main stream here,
see __init__) and fused AdamW -- those methods are SD3TransformerLoRA's own, bound here unchanged.

What differs from the SD3 model:
  * ACTIVATION RECOMPUTATION PER BLOCK.  Sixty blocks at CFG batch 16 and 4224 joint tokens would keep ~7 GB of activations
    each (440 GB); `forward_train` keeps only the two residual streams entering every block and the block's attention output +
    log-sum-exp (0.8 GB per block) and `backward` re-runs the block's Linears and row kernels -- the same launches, so the same
    bits; not the attention, not the feed-forward's second Linear -- before differentiating it: ~1.25 x the matrix work for 1/9
    of the memory, the whole model + optimiser + checkpoints inside one GPU's 288 GB.
  * head dim 128: `attention_bwd_d128.hip`; QK-norm + rotary backward: `advgrpo_qk_norm_rope_bwd` (the forward's in-place kernel
    saves 1/rms per head).
  * every block is a full two-stream block (no dual attention, no context_pre_only last block): the text stream's gradient
    enters the last block as zero.
  * the modulation rows come from the timestep embedding alone and are shared by the batch (row stride 0).
fp8 Linears (enable_fp8, the arithmetic BASELINE config 5 names): the replay runs the SAME e4m3 launches as the rollout, so the
importance ratio starts at 1; the backward differentiates the bf16 Linear from the bf16 activations kept beside the e4m3 rows
(straight-through), exactly as SD3TransformerLoRA does.
"""
import torch

from . import ops
from .ema import EMAModuleWrapper
from .mmdit_train import RANK, RPAD, SD3TransformerLoRA, _Adapter
from .qwen_mmdit import QwenImageTransformer2DModel

TARGETS = ("to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj", "to_add_out")
GROUPS = {"qkv": ("to_q", "to_k", "to_v"), "cqkv": ("add_q_proj", "add_k_proj", "add_v_proj"), "out": ("to_out.0",), "cout": ("to_add_out",)}


class QwenImageTransformerLoRA(QwenImageTransformer2DModel):
    lora_mode = "merged"
    keep_attention = True      # the checkpoint of a block also holds its attention output + log-sum-exp (0.42 GB per block at config 5)

    def __init__(self, state_dict, cfg, device="cuda", lora_alpha=64, seed=0, lora_state=None):
        super().__init__(state_dict, cfg, device)
        self.scale = lora_alpha / RANK
        D = cfg.dim
        self.adapters, off = {}, 0
        for i in range(cfg.num_layers):
            for n in TARGETS:
                key = f"transformer_blocks.{i}.attn.{n}"
                self.adapters[key] = _Adapter(key, D, D, off, off + RPAD * D)
                off += RPAD * D + D * RPAD
        self.n_params = off
        dev = self.device
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        g = torch.Generator().manual_seed(seed)
        for key, ad in self.adapters.items():
            if lora_state is not None:
                A, Bm = lora_state[key + ".lora_A.weight"].float(), lora_state[key + ".lora_B.weight"].float()
            else:                                                  # init_lora_weights="gaussian": A ~ N(0, 1/r), B = 0
                A, Bm = torch.randn(RANK, ad.K, generator=g) / RANK, torch.zeros(ad.N, RANK)
            self.A_view(ad)[:RANK] = A.to(dev)
            self.B_view(ad)[:, :RANK] = Bm.to(dev)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.params_bf16 = self.params.to(torch.bfloat16)
        self.opt_step = 0
        self.ema_wrapper = EMAModuleWrapper([self.params], decay=0.9, update_step_interval=8, device=dev)
        self.ema = self.ema_wrapper.ema_parameters[0]
        # adapter gradients on the main stream: at this size the side stream of the SD3 model buys 1.5 % (3.33 vs 3.38 s) and in some
        # process / allocation histories its kernels starve beside the attention backward -- micro-steps of 5.6 - 6.3 s instead of 2.8 - 3.3
        # were measured (second micro-step of an epoch, later processes of a box session); off is the robust choice
        self.overlap_wgrad = False
        self._wgrad_stream = None
        self._base_T = {}
        self._prepare_transposes()
        self.refresh()

    # the flat-vector machinery is the SD3 model's (same adapter layout, same kernels)
    A_view = SD3TransformerLoRA.A_view
    B_view = SD3TransformerLoRA.B_view
    lora_state_dict = SD3TransformerLoRA.lora_state_dict
    load_lora_state = SD3TransformerLoRA.load_lora_state
    save_pretrained = SD3TransformerLoRA.save_pretrained
    lora_grads = SD3TransformerLoRA.lora_grads
    refresh = SD3TransformerLoRA.refresh
    merge_one_launch = SD3TransformerLoRA.merge_one_launch
    _groups = SD3TransformerLoRA._groups
    _group_buffers = SD3TransformerLoRA._group_buffers
    _refresh_one_launch = SD3TransformerLoRA._refresh_one_launch
    _refresh_per_adapter = SD3TransformerLoRA._refresh_per_adapter
    _lora_wgrad = SD3TransformerLoRA._lora_wgrad
    _lora_wgrad_now = SD3TransformerLoRA._lora_wgrad_now
    _lora_wgrad_group = SD3TransformerLoRA._lora_wgrad_group
    _lora_wgrad_group_now = SD3TransformerLoRA._lora_wgrad_group_now
    optimizer_step = SD3TransformerLoRA.optimizer_step
    ema_step = SD3TransformerLoRA.ema_step
    # the KL term's reference policy (train.beta > 0; g_step.micro_step): the same weight swap -- base bf16 weights and, in fp8 mode, base
    # e4m3 rows of the adapted projections -- around this class's own rollout forward (the method was missing, beta > 0 raised
    # AttributeError in the middle of an epoch)
    _fp8_base = None
    forward_reference = SD3TransformerLoRA.forward_reference

    def _base_forward(self, *a, **kw):
        return QwenImageTransformer2DModel.__call__(self, *a, **kw)

    def _prepare_transposes(self):
        """Transposed copies for the data-gradient GEMMs; base (un-merged) copies of the adapted weights and their transposes
        (the blocks hold the base weights until the first refresh() merges the adapters in)."""
        T = lambda w: w.t().contiguous()
        for i, b in enumerate(self.blocks):
            b["last"] = False                  # (refresh() asks: every Qwen-Image block has a text-stream output projection)
            for k in ("ff1", "ff2", "cff1", "cff2"):
                b[k + ".wT"] = T(b[k + ".w"])
            self._base_T[i] = {gk: (b[gk + ".w"].clone(), T(b[gk + ".w"])) for gk in GROUPS}
        self.w["proj_out.wT"] = T(self.w["proj_out.w"])

    # ------------------------------------------------------------------ one block: the launches of __call__ (bf16 or fp8 Linears)
    def _block(self, i, x, c, mods, rope, B, Ni, Nt, save=None, keep_att=None, att_kept=None):
        """In place on x [B * Ni, D], c [B * Nt, D].  `save` (a dict) receives what the block's backward needs -- the RE-RUN inside
        backward(): it takes the attention output and log-sum-exp the first run kept (`att_kept`) instead of launching the attention
        again, and stops after the feed-forward's first Linear (nothing differentiates the block's output).  `keep_att` (a list):
        the first run appends (att, lse)."""
        cfg, b = self.cfg, self.blocks[i]
        D, H, hd = cfg.dim, cfg.num_heads, cfg.head_dim
        S, Mi, Mt = Ni + Nt, B * Ni, B * Nt
        dev, bf16 = x.device, torch.bfloat16
        kx, kc = ("x", i), ("c", i)

        def mod(key, j):
            o = self.mod_off[key] + j * D
            return mods[:, o:o + D]

        f8 = self.fp8

        def linears(items):
            if f8 is not None:
                return ops.gemm_grouped_fp8([ops.gemm_desc_fp8(a, f8[(i, key)], bias=b[key + ".b"], **kw) for a, key, kw in items])
            return ops.gemm_grouped([ops.gemm_desc(a, b[key + ".w"], bias=b[key + ".b"], **kw) for a, key, kw in items])

        def rows8(M, K):
            return ops.Fp8Rows(torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, dtype=torch.float32, device=dev))

        def norm(t, key, j_scale, j_shift, rows, q=None):
            """LayerNorm + modulation: bf16 rows (what the backward keeps) and, in fp8 mode, the e4m3 rows the Linears read."""
            if f8 is None:
                y = ops.layernorm_mod(t, scale=mod(key, j_scale), shift=mod(key, j_shift), rows_per_batch=rows)
                return y, y
            y = torch.empty_like(t) if save is not None else None
            ops.layernorm_mod_fp8(t, q, out=y, scale=mod(key, j_scale), shift=mod(key, j_shift), rows_per_batch=rows)
            return y, q
        q_n = rows8(Mi + Mt, D) if f8 is not None else None
        nx, nx_in = norm(x, kx, 1, 0, Ni, q_n.rows(0, Mi) if f8 is not None else None)
        nc, nc_in = norm(c, kc, 1, 0, Nt, q_n.rows(Mi, Mi + Mt) if f8 is not None else None)
        qkv = torch.empty(B * S, 3 * D, dtype=bf16, device=dev)
        qkv3 = qkv.view(B, S, 3 * D)
        linears([(nx_in, "qkv", dict(out=qkv, seg=(Ni, S, 0))), (nc_in, "cqkv", dict(out=qkv, seg=(Nt, S, Ni)))])
        rs = torch.empty(B * S, 2 * H, dtype=torch.float32, device=dev) if save is not None else None
        ops.qk_norm_rope(qkv, S, Ni, 2 * H, hd, b["rms_x"], b["rms_c"], H, rope=rope, eps=1e-6, rs_out=rs)
        if att_kept is not None:
            att, lse = att_kept
        else:
            att = torch.empty(B, S, D, dtype=bf16, device=dev)
            lse = torch.empty(B, H, S, dtype=torch.float32, device=dev) if (save is not None or keep_att is not None) else None
            ops.attention(qkv3[:, :, :D], qkv3[:, :, D:2 * D], qkv3[:, :, 2 * D:], H, out=att, lse=lse)
            if keep_att is not None:
                keep_att.append((att, lse))
        att2d = att.view(B * S, D)
        if f8 is not None:
            ops.quant_fp8_rows(att2d, out=q_n, split=(Ni, S))
            linears([(q_n.rows(0, Mi), "out", dict(gate=mod(kx, 2), gate_rows=Ni, residual=x, out=x)),
                     (q_n.rows(Mi, Mi + Mt), "cout", dict(gate=mod(kc, 2), gate_rows=Nt, residual=c, out=c))])
        else:
            linears([(att2d, "out", dict(gate=mod(kx, 2), gate_rows=Ni, residual=x, out=x, a_seg=(Ni, S, 0), M=Mi)),
                     (att2d, "cout", dict(gate=mod(kc, 2), gate_rows=Nt, residual=c, out=c, a_seg=(Nt, S, Ni), M=Mt))])
        pre = cpre = None
        if save is not None:
            save.update(nx=nx, nc=nc, qkv=qkv, rs=rs, att=att, lse=lse, x_mid=x.clone(), c_mid=c.clone())
            pre = torch.empty(Mi, 4 * D, dtype=bf16, device=dev)
            cpre = torch.empty(Mt, 4 * D, dtype=bf16, device=dev)
            save.update(pre=pre, cpre=cpre)
        aux = (lambda t: dict(aux_out=t)) if save is not None else (lambda t: {})
        if f8 is not None:            # (the MLP's normalised inputs are not kept: its backward needs the pre-activations only)
            ops.layernorm_mod_fp8(x, q_n.rows(0, Mi), scale=mod(kx, 4), shift=mod(kx, 3), rows_per_batch=Ni)
            ops.layernorm_mod_fp8(c, q_n.rows(Mi, Mi + Mt), scale=mod(kc, 4), shift=mod(kc, 3), rows_per_batch=Nt)
            h_all = torch.empty(Mi + Mt, 4 * D, dtype=bf16, device=dev)
            linears([(q_n.rows(0, Mi), "ff1", dict(act="gelu_tanh", out=h_all[:Mi], **aux(pre))),
                     (q_n.rows(Mi, Mi + Mt), "cff1", dict(act="gelu_tanh", out=h_all[Mi:], **aux(cpre)))])
            if save is not None:
                return
            q_h = ops.quant_fp8_rows(h_all)
            hm = [q_h.rows(0, Mi), q_h.rows(Mi, Mi + Mt)]
        else:
            nx2 = ops.layernorm_mod(x, scale=mod(kx, 4), shift=mod(kx, 3), rows_per_batch=Ni)
            nc2 = ops.layernorm_mod(c, scale=mod(kc, 4), shift=mod(kc, 3), rows_per_batch=Nt)
            hm = linears([(nx2, "ff1", dict(act="gelu_tanh", **aux(pre))), (nc2, "cff1", dict(act="gelu_tanh", **aux(cpre)))])
        if save is not None:
            return
        linears([(hm[0], "ff2", dict(gate=mod(kx, 5), gate_rows=Ni, residual=x, out=x)),
                 (hm[1], "cff2", dict(gate=mod(kc, 5), gate_rows=Nt, residual=c, out=c))])

    # ------------------------------------------------------------------ forward keeping one checkpoint per block
    @torch.no_grad()
    def forward_train(self, hidden_states, timestep, encoder_hidden_states, pooled_projections=None):
        """Same arithmetic as __call__ (bf16 or fp8 Linears).  Returns (v [B,16,h,w] bf16, ctx)."""
        cfg, w = self.cfg, self.w
        D = cfg.dim
        B, C, h, wd = hidden_states.shape
        hh, ww = h // cfg.patch_size, wd // cfg.patch_size
        Ni, Nt = hh * ww, encoder_hidden_states.shape[1]
        x = ops.gemm(ops.patchify(hidden_states.contiguous()), w["img_in.w"], bias=w["img_in.b"])
        t_rows = timestep[:1] if (timestep.dim() == 0 or timestep.numel() == 1 or timestep.stride(0) == 0) else timestep
        mods = self._mods(self._temb(t_rows.reshape(-1)))
        mods = mods.expand(B, -1) if mods.shape[0] == 1 else mods
        c = self.embed_context(encoder_hidden_states)
        rope = self._rope(hh, ww, Nt)
        ctx = {"B": B, "Ni": Ni, "Nt": Nt, "h": h, "w": wd, "mods": mods, "rope": rope, "x_in": [], "c_in": [], "att": []}
        for i in range(cfg.num_layers):
            ctx["x_in"].append(x.clone())
            ctx["c_in"].append(c.clone())
            self._block(i, x, c, mods, rope, B, Ni, Nt, keep_att=ctx["att"] if self.keep_attention else None)
        ctx["x_final"] = x
        o = self.mod_off[("out",)]
        nx = ops.layernorm_mod(x, scale=mods[:, o:o + D], shift=mods[:, o + D:o + 2 * D], rows_per_batch=Ni)
        tok = ops.gemm(nx, w["proj_out.w"], bias=w["proj_out.b"])
        return ops.unpatchify(tok, B, cfg.out_channels, h, wd, torch.bfloat16), ctx

    # ------------------------------------------------------------------ explicit backward, one recomputed block at a time
    @torch.no_grad()
    def backward(self, ctx, dv):
        """dv: gradient w.r.t. the model output [B,16,h,w] (bf16).  Accumulates LoRA gradients into self.grads; returns the
        gradients w.r.t. the two embedded streams."""
        cfg, w = self.cfg, self.w
        D, H, hd = cfg.dim, cfg.num_heads, cfg.head_dim
        B, Ni, Nt = ctx["B"], ctx["Ni"], ctx["Nt"]
        S = Ni + Nt
        mods, rope = ctx["mods"], ctx["rope"]
        dev, bf16 = dv.device, torch.bfloat16

        def mod(key, j):
            o = self.mod_off[key] + j * D
            return mods[:, o:o + D]
        side = self._wgrad_stream if self.overlap_wgrad else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        dtok = SD3TransformerLoRA._patch_rows_of_output_grad(self, dv)           # [B*Ni, 64], column (py*2+px)*C + c
        dnx = ops.gemm(dtok, w["proj_out.wT"])
        dx = ops.layernorm_mod_bwd(ctx["x_final"], dnx, scale0=mod(("out",), 0), rows_per_batch=Ni)
        dc = torch.zeros(B * Nt, D, dtype=bf16, device=dev)       # nothing reads the text stream after the last block
        for i in reversed(range(cfg.num_layers)):
            b, s = self.blocks[i], {}
            kx, kc = ("x", i), ("c", i)
            x_in, c_in = ctx["x_in"][i], ctx["c_in"][i]
            self._block(i, x_in.clone(), c_in.clone(), mods, rope, B, Ni, Nt, save=s, att_kept=ctx["att"][i] if ctx["att"] else None)
            # ---- MLPs (the text-stream GEMMs ride in the launches of their image-stream twins, as in the forward)
            dyg, dcyg = ops.gate_mul(dx, mod(kx, 5), Ni), ops.gate_mul(dc, mod(kc, 5), Nt)
            dpres = ops.gemm_grouped([ops.gemm_desc(dyg, b["ff2.wT"], act="dgelu_tanh", aux_in=s["pre"]),
                                      ops.gemm_desc(dcyg, b["cff2.wT"], act="dgelu_tanh", aux_in=s["cpre"])])
            dmid = ops.gemm_grouped([ops.gemm_desc(dpres[0], b["ff1.wT"]), ops.gemm_desc(dpres[1], b["cff1.wT"])])
            dx1 = ops.layernorm_mod_bwd(s["x_mid"], dmid[0], scale0=mod(kx, 4), dres=dx, rows_per_batch=Ni)
            dc1 = ops.layernorm_mod_bwd(s["c_mid"], dmid[1], scale0=mod(kc, 4), dres=dc, rows_per_batch=Nt)
            del dpres, dmid, dyg, dcyg
            # ---- joint attention
            datt = torch.empty(B * S, D, dtype=bf16, device=dev)
            dyo, dyc = ops.gate_mul(dx1, mod(kx, 2), Ni), ops.gate_mul(dc1, mod(kc, 2), Nt)
            ops.gemm_grouped([ops.gemm_desc(dyo, b["out.wT"], out=datt, seg=(Ni, S, 0)),
                              ops.gemm_desc(dyc, b["cout.wT"], out=datt, seg=(Nt, S, Ni))])
            att2d = s["att"].view(B * S, D)
            self._lora_wgrad_group([((i, "out"), att2d, B * Ni, (Ni, S, 0), dyo, None), ((i, "cout"), att2d, B * Nt, (Nt, S, Ni), dyc, None)])
            q3 = s["qkv"].view(B, S, 3 * D)
            dqkv = torch.empty(B * S, 3 * D, dtype=bf16, device=dev)
            d3 = dqkv.view(B, S, 3 * D)
            ops.attention_bwd(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], s["att"], datt.view(B, S, D), s["lse"], H,
                              d3[:, :, :D], d3[:, :, D:2 * D], d3[:, :, 2 * D:])
            ops.qk_norm_rope_bwd(dqkv, s["qkv"], s["rs"], S, Ni, 2 * H, hd, b["rms_x"], b["rms_c"], H, rope=rope)
            dnx, dnc = ops.gemm_grouped([ops.gemm_desc(dqkv, b["qkv.wT"], a_seg=(Ni, S, 0), M=B * Ni),
                                         ops.gemm_desc(dqkv, b["cqkv.wT"], a_seg=(Nt, S, Ni), M=B * Nt)])
            self._lora_wgrad_group([((i, "qkv"), s["nx"], B * Ni, None, dqkv, (Ni, S, 0)), ((i, "cqkv"), s["nc"], B * Nt, None, dqkv, (Nt, S, Ni))])
            # ---- first norms
            dx = ops.layernorm_mod_bwd(x_in, dnx, scale0=mod(kx, 1), dres=dx1, rows_per_batch=Ni)
            dc = ops.layernorm_mod_bwd(c_in, dnc, scale0=mod(kc, 1), dres=dc1, rows_per_batch=Nt)
            ctx["x_in"][i] = ctx["c_in"][i] = None       # the checkpoint is spent
            if ctx["att"]:
                ctx["att"][i] = None
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        return dx, dc
