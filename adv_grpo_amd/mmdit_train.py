"""LoRA training of the MMDiT on the gfx950 kernels: forward with saved activations, explicit backward, AdamW, EMA.

Replaces, for the G-step (scripts/train_sd3_fast_pickscore.py:1077-1187): peft's LoRA wrapping of the attention
projections (TP:490-511: r=32, alpha=64, gaussian init, targets attn.{to_q,to_k,to_v,to_out.0,add_q_proj,
add_k_proj,add_v_proj,to_add_out}), torch autograd through diffusers' SD3Transformer2DModel, DeepSpeed/accelerate
gradient accumulation + clip_grad_norm_ + AdamW (TP:1165-1171, TP:554-561) and EMAModuleWrapper (adv_grpo/ema.py).

MI355X-first choices:
  * lora_mode="merged" (default): LoRA is MERGED into the bf16 weights (W_eff = W + (alpha/r) B A, and the transposed
    copy used by the data-gradient GEMMs); forward and dgrad therefore run at full-GEMM efficiency with no rank-32 side
    path.  lora_mode="side": PEFT's arithmetic -- y = x W^T + s (x A^T) B^T with W untouched -- as a K-EXTENSION of the
    same GEMM: the adapted Linear's input lives in a buffer with E extra columns that a skinny GEMM fills with
    u = x A^T, and the Linear contracts [x | u] with [W | s B] over K + E (E = 192 for the fused QKV, 64 for the output
    projections).  An update of B then reaches the log-probs at full bf16 resolution of s B instead of being rounded
    into W (deviation 2).  The backward still uses the merged transposes for dX (gradient-side rounding only).
    The LoRA weight gradients are recovered from (X, dY) of each adapted Linear:
        dB = s * dY^T (X A^T),   dA = s * (dY B)^T X
    as split-K GEMMs over the token axis accumulating atomically into the flat f32 gradient vector.
  * No activation recomputation: with 288 GB of HBM every tensor the backward needs is kept (~0.6 GB per block per
    micro-step at batch 16).
  * Ranks are padded 32 -> 64 (zero rows/columns) so every contraction is a multiple of the 64-deep MFMA k tile;
    the padding provably stays zero under AdamW.
  * All LoRA parameters / gradients / Adam moments live in ONE flat f32 vector each: one fused AdamW launch,
    one sum-of-squares launch for clip_grad_norm_, one RCCL all-reduce of the gradient vector for data parallelism.
"""
import math

import torch

from . import _lib
from . import ops
from .ema import EMAModuleWrapper
from .mmdit import SD3Transformer2DModel

RANK, RPAD = 32, 64


class _Adapter:
    __slots__ = ("name", "N", "K", "offA", "offB")

    def __init__(self, name, N, K, offA, offB):
        self.name, self.N, self.K, self.offA, self.offB = name, N, K, offA, offB


class SD3TransformerLoRA(SD3Transformer2DModel):
    def __init__(self, state_dict, cfg, device="cuda", lora_alpha=64, seed=0, lora_state=None, lora_mode="merged"):
        if lora_mode not in ("merged", "side"):
            raise ValueError(f"lora_mode must be 'merged' or 'side', got {lora_mode!r}")
        self.lora_mode = lora_mode
        self._base_sd = {k: v for k, v in state_dict.items()}
        super().__init__(state_dict, cfg, device)
        if lora_mode == "side":
            self.lora_ext = (3 * RPAD, RPAD)
        self.scale = lora_alpha / RANK
        D = cfg.dim
        # ---- flat parameter vector: per adapter A_pad [64, K] then B_pad [N, 64]
        self.adapters = {}
        off = 0
        for i in range(cfg.num_layers):
            names = ["to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj"]
            if i != cfg.num_layers - 1:
                names.append("to_add_out")
            for n in names:
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
        # EMA of the trainable parameters from their INITIAL values (EMAModuleWrapper built at construction, TP:528);
        # self.ema aliases its one flat tensor (eval swap, EMA checkpoints)
        self.ema_wrapper = EMAModuleWrapper([self.params], decay=0.9, update_step_interval=8, device=dev)
        self.ema = self.ema_wrapper.ema_parameters[0]
        self._base_T = {}
        # adapter-gradient launches run on this stream beside the main chain (one stream for the model: the token-contracted
        # GEMMs share a workspace, and micro-steps issued from two host threads must not each make their own)
        self.overlap_wgrad = True
        # the adapter-gradient side stream: one that is measured to run beside the current stream (ops.concurrent_stream: with
        # HIP's few hardware queues a plain torch.cuda.Stream() may share the main stream's queue and overlap nothing)
        self._wgrad_stream = ops.concurrent_stream(dev)
        self._prepare_transposes()
        self.refresh()

    # ------------------------------------------------------------------ parameter views
    def A_view(self, ad, src=None):
        return (self.params if src is None else src)[ad.offA:ad.offA + RPAD * ad.K].view(RPAD, ad.K)

    def B_view(self, ad, src=None):
        return (self.params if src is None else src)[ad.offB:ad.offB + ad.N * RPAD].view(ad.N, RPAD)

    def lora_state_dict(self):
        """peft-style names -> [r,K] / [N,r] f32 tensors (unpadded)."""
        out = {}
        for key, ad in self.adapters.items():
            out[key + ".lora_A.weight"] = self.A_view(ad)[:RANK].clone()
            out[key + ".lora_B.weight"] = self.B_view(ad)[:, :RANK].clone()
        return out

    def load_lora_state(self, lora_state):
        """PeftModel.from_pretrained (TP:506-509): overwrite the adapters and re-merge.  Optimizer moments are kept."""
        for key, ad in self.adapters.items():
            A, Bm = lora_state[key + ".lora_A.weight"], lora_state[key + ".lora_B.weight"]
            if tuple(A.shape) != (RANK, ad.K) or tuple(Bm.shape) != (ad.N, RANK):
                raise ValueError(f"{key}: adapter shapes {tuple(A.shape)} / {tuple(Bm.shape)} do not match r={RANK}")
            self.A_view(ad).zero_(); self.B_view(ad).zero_()
            self.A_view(ad)[:RANK] = A.to(self.device, torch.float32)
            self.B_view(ad)[:, :RANK] = Bm.to(self.device, torch.float32)
        self.params_bf16 = self.params.to(torch.bfloat16)
        self.ema.copy_(self.params)          # the reference rebuilds the EMA after PeftModel.from_pretrained (TP:506-528)
        self.refresh()

    def save_pretrained(self, path, use_ema=False):
        """save_ckpt (TP:389-398): PEFT layout; with use_ema the EMA weights are what is written (copy_ema_to /
        copy_temp_to around save_pretrained upstream: the live parameters are left untouched here)."""
        from . import checkpoint
        if use_ema and self.ema is not None:
            live, self.params = self.params, self.ema
            try:
                state = self.lora_state_dict()
            finally:
                self.params = live
        else:
            state = self.lora_state_dict()
        checkpoint.save_lora(path, state, r=RANK, lora_alpha=int(round(self.scale * RANK)))

    def lora_grads(self):
        out = {}
        for key, ad in self.adapters.items():
            out[key + ".lora_A.weight"] = self.A_view(ad, self.grads)[:RANK].clone()
            out[key + ".lora_B.weight"] = self.B_view(ad, self.grads)[:, :RANK].clone()
        return out

    # ------------------------------------------------------------------ frozen transposes for the data-gradient GEMMs
    def _prepare_transposes(self):
        dev = self.device
        T = lambda w: w.t().contiguous()
        for i, b in enumerate(self.blocks):
            for k in ("ff1", "ff2", "cff1", "cff2", "qkv2", "out2"):
                if k + ".w" in b:
                    b[k + ".wT"] = T(b[k + ".w"])
        self.w["proj_out.wT"] = T(self.w["proj_out.w"])
        # base (un-merged) copies of the adapted weights and their transposes
        sd = self._base_sd
        bf = lambda t: t.to(device=dev, dtype=torch.bfloat16).contiguous()
        for i in range(self.cfg.num_layers):
            p = f"transformer_blocks.{i}.attn"
            last = i == self.cfg.num_layers - 1
            base = {"qkv": bf(torch.cat([sd[f"{p}.to_q.weight"], sd[f"{p}.to_k.weight"], sd[f"{p}.to_v.weight"]])),
                    "cqkv": bf(torch.cat([sd[f"{p}.add_q_proj.weight"], sd[f"{p}.add_k_proj.weight"],
                                          sd[f"{p}.add_v_proj.weight"]])),
                    "out": bf(sd[f"{p}.to_out.0.weight"])}
            if not last:
                base["cout"] = bf(sd[f"{p}.to_add_out.weight"])
            self._base_T[i] = {k: (v, T(v)) for k, v in base.items()}
        del self._base_sd

    # ------------------------------------------------------------------ merge LoRA into the working weights
    merge_one_launch = True    # refresh() through advgrpo_lora_merge (one launch for the model); False: the per-adapter GEMMs of rounds 3 - 4

    def _groups(self, b):
        groups = {"qkv": ["to_q", "to_k", "to_v"], "cqkv": ["add_q_proj", "add_k_proj", "add_v_proj"], "out": ["to_out.0"]}
        if not b["last"]:
            groups["cout"] = ["to_add_out"]
        return groups

    @torch.no_grad()
    def refresh(self):
        """bf16 copies of A/B and W_eff / W_eff^T of every adapted projection (after construction, a state load, every optimizer step):
        W_eff[n,k] = W + s * B A, plus the stacked A and the block-diagonal B^T of every Linear group (the adapter-gradient GEMMs' operands)."""
        self.params_bf16.copy_(self.params)
        if not hasattr(self, "_Bbd"):
            self._Bbd, self._Acat = {}, {}
        if self.merge_one_launch:
            self._refresh_one_launch()
        else:
            self._refresh_per_adapter()
        if self.lora_mode == "side":
            # forward weight [W | s B]: the base weight is left as loaded, each adapter's s * B (bf16; s = 2 is exact)
            # sits in its own 64 side columns of its output rows
            D = self.cfg.dim
            for i, b in enumerate(self.blocks):
                p = f"transformer_blocks.{i}.attn"
                for gk, names in self._groups(b).items():
                    base = self._base_T[i][gk][0]
                    K = base.shape[1]
                    E = RPAD * len(names)
                    wx = b.get(gk + ".wx")
                    if wx is None:
                        wx = b[gk + ".wx"] = torch.zeros(base.shape[0], K + E, dtype=torch.bfloat16, device=self.device)
                        wx[:, :K] = base
                    for j, n in enumerate(names):
                        B16 = self.B_view(self.adapters[f"{p}.{n}"], self.params_bf16)
                        wx[j * D:(j + 1) * D, K + j * RPAD:K + (j + 1) * RPAD] = (self.scale * B16.float()).to(torch.bfloat16)
                    b[gk + ".w"] = wx
                    b[gk + ".A"] = self._lora[(i, gk)][0]
        if self.fp8 is not None:                 # fp8 Linears (enable_fp8): the merged weights have just changed
            self.requantize()

    def _group_buffers(self, i, b, gk, names):
        """The persistent outputs of one Linear group: W_eff^T, the stacked A [64 n, K], the block-diagonal B^T [64 n, n D]."""
        base = self._base_T[i][gk][0]
        wT = b.get(gk + ".wT")
        if wT is None:
            wT = b[gk + ".wT"] = torch.empty(base.shape[1], base.shape[0], dtype=torch.bfloat16, device=self.device)
        n_ad, D = len(names), self.cfg.dim
        A_cat = self._Acat.get((i, gk))
        if A_cat is None:
            A_cat = self._Acat[(i, gk)] = torch.empty(RPAD * n_ad, base.shape[1], dtype=torch.bfloat16, device=self.device)
        # [B_0^T; B_1^T; ...] as ONE block-diagonal right operand [64 n, n D]: u = [dY_0 B_0 | dY_1 B_1 | ...] is then a single
        # GEMM over the group's whole output gradient (K = n D; the off-diagonal zeros cost flops on a pass that is bound by
        # reading dY) instead of n launches (kept across refreshes: only the diagonal blocks are rewritten)
        Bbd = self._Bbd.get((i, gk))
        if Bbd is None:
            Bbd = self._Bbd[(i, gk)] = torch.zeros(RPAD * n_ad, n_ad * D, dtype=torch.bfloat16, device=self.device)
        return base, wT, A_cat, Bbd

    def _refresh_one_launch(self):
        """advgrpo_lora_merge over a device table of every adapter (csrc/lora_merge.hip): ~950 launches of ~8 us per optimizer step became one."""
        D = self.cfg.dim
        items, self._lora = [], {}
        pb = self.params_bf16.data_ptr()
        for i, b in enumerate(self.blocks):
            p = f"transformer_blocks.{i}.attn"
            for gk, names in self._groups(b).items():
                base, wT, A_cat, Bbd = self._group_buffers(i, b, gk, names)
                w = b[gk + ".w"] if self.lora_mode == "merged" else None
                K = base.shape[1]
                ads = [self.adapters[f"{p}.{n}"] for n in names]
                for j, ad in enumerate(ads):
                    assert ad.N == D and ad.K == K and ad.N % 64 == 0 and K % 64 == 0
                    it = _lib.LoraMergeItem()
                    it.A, it.B = pb + 2 * ad.offA, pb + 2 * ad.offB
                    it.base, it.ld_base = base.data_ptr() + 2 * j * D * base.stride(0), base.stride(0)
                    if w is not None:
                        it.w, it.ld_w = w.data_ptr() + 2 * j * D * w.stride(0), w.stride(0)
                    it.wT, it.ld_wT = wT.data_ptr() + 2 * j * D, wT.stride(0)
                    it.a_cat = A_cat.data_ptr() + 2 * j * RPAD * K
                    it.b_bd, it.ld_bd = Bbd.data_ptr() + 2 * (j * RPAD * Bbd.stride(0) + j * D), Bbd.stride(0)
                    it.N, it.K = ad.N, ad.K
                    items.append(it)
                self._lora[(i, gk)] = (A_cat, None, ads, Bbd)
        raw = b"".join(bytes(it) for it in items)
        cached = getattr(self, "_merge_table", None)
        if cached is None or cached[0] != raw:      # (the pointers move when a state load replaces a tensor)
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
            self._merge_table = cached = (raw, table)
        _lib.check(_lib.load().advgrpo_lora_merge(cached[1].data_ptr(), len(items), max(it.N for it in items), max(it.K for it in items),
                                                  RPAD, float(self.scale), _lib.stream_ptr()))

    def _refresh_per_adapter(self):
        """Rounds 3 - 4: per adapter two 64-deep GEMMs (W_eff and, separately, W_eff^T) and two transposes; kept for A/Bs and as the
        comparison of tests/test_gpu_train.py (the one-launch kernel sums in f32 FMA order, these in the MFMA's: they agree to a bf16 ulp)."""
        D = self.cfg.dim
        self._lora = {}
        for i, b in enumerate(self.blocks):
            p = f"transformer_blocks.{i}.attn"
            for gk, names in self._groups(b).items():
                base, wT, A_cat, Bbd = self._group_buffers(i, b, gk, names)
                baseT = self._base_T[i][gk][1]
                w = b[gk + ".w"] if self.lora_mode == "merged" else None
                for j, n in enumerate(names):
                    ad = self.adapters[f"{p}.{n}"]
                    A16 = self.A_view(ad, self.params_bf16)                      # [64, K]
                    B16 = self.B_view(ad, self.params_bf16)                      # [N, 64]
                    AT = ops.transpose(A16)                                      # [K, 64]
                    sl = slice(j * D, (j + 1) * D)
                    # W_eff[n,k] = W + s * B A ;  W_eff^T[k,n] = W^T + s * A^T B^T
                    if w is not None:
                        ops.gemm(B16, AT, alpha=self.scale, residual=base[sl], out=w[sl])
                    ops.gemm(AT, B16, alpha=self.scale, residual=baseT[:, sl], out=wT[:, sl])
                    A_cat[j * RPAD:(j + 1) * RPAD] = A16
                    Bbd[j * RPAD:(j + 1) * RPAD, j * D:(j + 1) * D] = ops.transpose(B16)
                self._lora[(i, gk)] = (A_cat, None, [self.adapters[f"{p}.{n}"] for n in names], Bbd)

    # ------------------------------------------------------------------ the adapter-free transformer (KL reference)
    _fp8_base = None           # {(block, Linear): Fp8Rows of the BASE weight}, filled by the first fp8 reference forward

    def _base_forward(self, *a, **kw):
        """The rollout forward of the model class underneath the adapters (QwenImageTransformerLoRA binds its own)."""
        return SD3Transformer2DModel.__call__(self, *a, **kw)

    @torch.no_grad()
    def forward_reference(self, hidden_states, timestep, encoder_hidden_states, pooled_projections):
        """The forward under PEFT's `disable_adapter()` (TP:1105-1108: the reference policy of the KL term): the rollout
        forward on the base weights of the adapted projections.  Returns v [B,16,h,w] bf16."""
        saved = []
        for i, b in enumerate(self.blocks):
            for gk, (base, _) in self._base_T[i].items():
                keys = [gk + ".w"] + ([gk + ".A"] if gk + ".A" in b else [])
                saved.append((b, {k: b[k] for k in keys}))
                b[gk + ".w"] = base
                b.pop(gk + ".A", None)          # side mode: no side columns -> the plain [rows, D] input
        ext, self.lora_ext = getattr(self, "lora_ext", (0, 0)), (0, 0)
        # fp8 Linears (enable_fp8): the forward reads self.fp8[(block, Linear)], the quantised MERGED weights -- the reference
        # policy needs the quantised BASE weights of the adapted projections there (quantised once: they never change);
        # without the swap the "reference" equals the policy and the KL term vanishes silently
        f8_saved = {}
        if self.fp8 is not None:
            if self._fp8_base is None:
                self._fp8_base = {}
            for i, b in enumerate(self.blocks):
                for gk, (base, _) in self._base_T[i].items():
                    if (i, gk) not in self._fp8_base:
                        self._fp8_base[(i, gk)] = ops.quant_fp8_rows(base)
                    f8_saved[(i, gk)] = self.fp8[(i, gk)]
                    self.fp8[(i, gk)] = self._fp8_base[(i, gk)]
        try:
            (v,) = self._base_forward(hidden_states, timestep, encoder_hidden_states, pooled_projections)
        finally:
            self.lora_ext = ext
            for b, kv in saved:
                b.update(kv)
            if f8_saved:
                self.fp8.update(f8_saved)
        return v

    # ------------------------------------------------------------------ forward with saved activations
    @torch.no_grad()
    def forward_train(self, hidden_states, timestep, encoder_hidden_states, pooled_projections):
        """Same arithmetic as __call__, keeping what backward() needs.  Returns (v [B,16,h,w] bf16, ctx)."""
        cfg, w = self.cfg, self.w
        D, H = cfg.dim, cfg.num_heads
        B, C, h, wd = hidden_states.shape
        hh, ww = h // cfg.patch_size, wd // cfg.patch_size
        Ni, Nt = hh * ww, encoder_hidden_states.shape[1]
        S = Ni + Nt
        dev = hidden_states.device
        bf16 = torch.bfloat16
        x = ops.gemm(ops.patchify(hidden_states.contiguous()), w["patch.w"], bias=w["patch.b"], residual=self._pos(B, hh, ww))
        t1 = "time_text_embed.timestep_embedder.linear_1"; t2 = "time_text_embed.timestep_embedder.linear_2"
        p1 = "time_text_embed.text_embedder.linear_1"; p2 = "time_text_embed.text_embedder.linear_2"
        te = ops.gemm(ops.gemm(ops.timestep_embedding(timestep), w[t1 + ".w"], bias=w[t1 + ".b"], act="silu"),
                      w[t2 + ".w"], bias=w[t2 + ".b"])
        temb = ops.gemm(ops.gemm(pooled_projections.to(bf16).contiguous(), w[p1 + ".w"], bias=w[p1 + ".b"], act="silu"),
                        w[p2 + ".w"], bias=w[p2 + ".b"], residual=te)
        mods = ops.gemm(ops.unary(temb, "silu"), w["mod.w"], bias=w["mod.b"])
        c = ops.gemm(encoder_hidden_states.to(bf16).reshape(B * Nt, -1).contiguous(), w["context_embedder.w"],
                     bias=w["context_embedder.b"])

        def mod(key, j):
            o = self.mod_off[key] + j * D
            return mods[:, o:o + D]
        ctx = {"B": B, "Ni": Ni, "Nt": Nt, "h": h, "w": wd, "mods": mods, "blocks": []}
        # fp8 Linears (enable_fp8, mmdit.py): the replay runs the SAME arithmetic as the rollout -- same per-row quantisation of the
        # same inputs, same kernels -- so the importance ratio starts at 1 as in bf16 mode; the backward differentiates the
        # bf16 Linear (straight-through: quantisation is treated as the identity), from the bf16 activations saved here
        f8 = self.fp8
        Mi, Mt = B * Ni, B * Nt

        def linears(i, b, items):
            if f8 is not None:
                return ops.gemm_grouped_fp8([ops.gemm_desc_fp8(a, f8[(i, key)], bias=b[key + ".b"], **kw) for a, key, kw in items])
            return ops.gemm_grouped([ops.gemm_desc(a, b[key + ".w"], bias=b[key + ".b"], **kw) for a, key, kw in items])

        def quant(t, **kw):
            return ops.quant_fp8_rows(t, **kw)

        for i, b in enumerate(self.blocks):
            kx, kc = ("x", i), ("c", i)
            # the residual streams are written OUT of place by the gated projections below: the buffers a block received stay what
            # the norms' backward needs (x_in / c_in, x_mid / c_mid) -- no copies (round 5: 96 copy launches, 1.7 ms per micro-step)
            s = {"x_in": x, "c_in": c}
            x1 = torch.empty_like(x)
            c1 = c if b["last"] else torch.empty_like(c)
            Eq, Eo = self.lora_ext
            nx_buf = torch.empty(B * Ni, D + Eq, dtype=bf16, device=dev)          # (Eq = Eo = 0 in fp8 mode)
            nc_buf = torch.empty(B * Nt, D + Eq, dtype=bf16, device=dev)
            cs, ch = (0, 1) if b["last"] else (1, 0)
            if f8 is not None:      # the norms write the bf16 rows the backward keeps AND the e4m3 rows the Linears read
                q_n = ops.Fp8Rows(torch.empty(Mi + Mt, D, dtype=torch.uint8, device=dev), torch.empty(Mi + Mt, dtype=torch.float32, device=dev))
                nx_in, nc_in = q_n.rows(0, Mi), q_n.rows(Mi, Mi + Mt)
                nx2 = ops.Fp8Rows(torch.empty(Mi, D, dtype=torch.uint8, device=dev), torch.empty(Mi, dtype=torch.float32, device=dev)) \
                    if b["dual"] else None
                if b["dual"]:
                    ops.layernorm_mod_fp8(x, nx_in, q2=nx2, out=nx_buf, scale=mod(kx, 1), shift=mod(kx, 0), scale2=mod(kx, 7),
                                          shift2=mod(kx, 6), rows_per_batch=Ni)
                else:
                    ops.layernorm_mod_fp8(x, nx_in, out=nx_buf, scale=mod(kx, 1), shift=mod(kx, 0), rows_per_batch=Ni)
                ops.layernorm_mod_fp8(c, nc_in, out=nc_buf, scale=mod(kc, cs), shift=mod(kc, ch), rows_per_batch=Nt)
            else:
                if b["dual"]:
                    _, nx2 = ops.layernorm_mod(x, out=nx_buf[:, :D], scale=mod(kx, 1), shift=mod(kx, 0), scale2=mod(kx, 7),
                                               shift2=mod(kx, 6), rows_per_batch=Ni)
                else:
                    ops.layernorm_mod(x, out=nx_buf[:, :D], scale=mod(kx, 1), shift=mod(kx, 0), rows_per_batch=Ni)
                ops.layernorm_mod(c, out=nc_buf[:, :D], scale=mod(kc, cs), shift=mod(kc, ch), rows_per_batch=Nt)
                nx_in, nc_in = self._lora_side_pair(b, [("qkv", nx_buf, None, None), ("cqkv", nc_buf, None, None)], D)
            nx, nc = nx_buf[:, :D], nc_buf[:, :D]                   # what the backward keeps: the Linear's input proper
            qkv = torch.empty(B * S, 3 * D, dtype=bf16, device=dev)
            qkv3 = qkv.view(B, S, 3 * D)
            rs = torch.empty(B * S, 2 * H, dtype=torch.float32, device=dev)
            linears(i, b, [(nx_in, "qkv", dict(out=qkv, seg=(Ni, S, 0), rms=(b["rms_x"], 2 * H, H, 1e-6, rs))),
                           (nc_in, "cqkv", dict(out=qkv, seg=(Nt, S, Ni), rms=(b["rms_c"], 2 * H, H, 1e-6, rs)))])
            att_ext = torch.empty(B, S, D + Eo, dtype=bf16, device=dev)
            att = att_ext[:, :, :D]
            lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
            ops.attention(qkv3[:, :, :D], qkv3[:, :, D:2 * D], qkv3[:, :, 2 * D:], H, out=att, lse=lse)
            att_in = att_ext.view(B * S, D + Eo)
            if f8 is not None:
                q_a = quant(att_in, split=(Ni, S))
                outs = [(q_a.rows(0, Mi), "out", dict(gate=mod(kx, 2), gate_rows=Ni, residual=x, out=x1))]
                if not b["last"]:
                    outs.append((q_a.rows(Mi, Mi + Mt), "cout", dict(gate=mod(kc, 2), gate_rows=Nt, residual=c, out=c1)))
            else:
                if Eo:
                    self._lora_side_pair(b, [("out", att_in, (Ni, S, 0), B * Ni)] +
                                         ([] if b["last"] else [("cout", att_in, (Nt, S, Ni), B * Nt)]), D)
                outs = [(att_in, "out", dict(gate=mod(kx, 2), gate_rows=Ni, residual=x, out=x1, a_seg=(Ni, S, 0), M=B * Ni))]
                if not b["last"]:
                    outs.append((att_in, "cout", dict(gate=mod(kc, 2), gate_rows=Nt, residual=c, out=c1, a_seg=(Nt, S, Ni),
                                                      M=B * Nt)))
            linears(i, b, outs)
            x, c = x1, c1
            s.update(nx=nx, nc=nc, qkv=qkv, rs=rs, att=att, lse=lse)
            if b["dual"]:
                rs2 = torch.empty(B * Ni, 2 * H, dtype=torch.float32, device=dev)
                (qkv2,) = linears(i, b, [(nx2, "qkv2", dict(rms=(b["rms_2"], 2 * H, H, 1e-6, rs2)))])
                q3 = qkv2.view(B, Ni, 3 * D)
                lse2 = torch.empty(B, H, Ni, dtype=torch.float32, device=dev)
                o2 = ops.attention(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], H, lse=lse2)
                o2d = o2.view(B * Ni, D)
                linears(i, b, [(quant(o2d) if f8 is not None else o2d, "out2", dict(gate=mod(kx, 8), gate_rows=Ni, residual=x, out=x))])
                s.update(qkv2=qkv2, rs2=rs2, att2=o2, lse2=lse2)
            s["x_mid"] = x
            if not b["last"]:
                s["c_mid"] = c
            x2 = torch.empty_like(x)
            c2 = c if b["last"] else torch.empty_like(c)
            pre = torch.empty(B * Ni, 4 * D, dtype=bf16, device=dev)
            s.update(pre=pre)
            cpre = None
            if not b["last"]:
                cpre = torch.empty(B * Nt, 4 * D, dtype=bf16, device=dev)
                s.update(cpre=cpre)
            if f8 is not None:
                q_m = ops.Fp8Rows(torch.empty(Mi + Mt, D, dtype=torch.uint8, device=dev), torch.empty(Mi + Mt, dtype=torch.float32, device=dev))
                ops.layernorm_mod_fp8(x, q_m.rows(0, Mi), scale=mod(kx, 4), shift=mod(kx, 3), rows_per_batch=Ni)
                if not b["last"]:
                    ops.layernorm_mod_fp8(c, q_m.rows(Mi, Mi + Mt), scale=mod(kc, 4), shift=mod(kc, 3), rows_per_batch=Nt)
                h_all = torch.empty(Mi + Mt, 4 * D, dtype=bf16, device=dev)
                ff1 = [(q_m.rows(0, Mi), "ff1", dict(act="gelu_tanh", aux_out=pre, out=h_all[:Mi]))]
                if not b["last"]:
                    ff1.append((q_m.rows(Mi, Mi + Mt), "cff1", dict(act="gelu_tanh", aux_out=cpre, out=h_all[Mi:])))
                linears(i, b, ff1)
                q_h = quant(h_all)
                hm = [q_h.rows(0, Mi), q_h.rows(Mi, Mi + Mt)]
            else:
                nxm = ops.layernorm_mod(x, scale=mod(kx, 4), shift=mod(kx, 3), rows_per_batch=Ni)
                ff1 = [(nxm, "ff1", dict(act="gelu_tanh", aux_out=pre))]
                if not b["last"]:
                    ncm = ops.layernorm_mod(c, scale=mod(kc, 4), shift=mod(kc, 3), rows_per_batch=Nt)
                    ff1.append((ncm, "cff1", dict(act="gelu_tanh", aux_out=cpre)))
                hm = linears(i, b, ff1)
            ff2 = [(hm[0], "ff2", dict(gate=mod(kx, 5), gate_rows=Ni, residual=x, out=x2))]
            if not b["last"]:
                ff2.append((hm[1], "cff2", dict(gate=mod(kc, 5), gate_rows=Nt, residual=c, out=c2)))
            linears(i, b, ff2)
            x, c = x2, c2
            ctx["blocks"].append(s)
        ctx["x_final"] = x
        nx = ops.layernorm_mod(x, scale=mod(("out",), 0), shift=mod(("out",), 1), rows_per_batch=Ni)
        tok = ops.gemm(nx, w["proj_out.w"], bias=w["proj_out.b"])
        out = ops.unpatchify(tok, B, cfg.out_channels, h, wd, bf16)
        return out, ctx

    # ------------------------------------------------------------------ LoRA weight gradients of one Linear group
    def _lora_wgrad(self, key, X, x_rows, x_seg, dY, dy_seg):
        """The adapter gradients of one Linear group, on the side stream: they only read (X, dY) and add into the flat gradient
        vector, so their ~20 small launches per block run beside the data-gradient chain of the next layers instead of in it.
        X: activations (rows via x_seg), dY: [.., n_adapters*D] output gradient (rows via dy_seg)."""
        if self._wgrad_stream is None or not self.overlap_wgrad:
            return self._lora_wgrad_now(key, X, x_rows, x_seg, dY, dy_seg)
        ready = torch.cuda.Event()
        ready.record()                                   # X and dY are complete on the calling stream
        self._wgrad_stream.wait_event(ready)
        for t in (X, dY):                                # read on the side stream after their Python names are gone
            t.record_stream(self._wgrad_stream)
        with torch.cuda.stream(self._wgrad_stream):
            self._lora_wgrad_now(key, X, x_rows, x_seg, dY, dy_seg)

    def _lora_wgrad_now(self, key, X, x_rows, x_seg, dY, dy_seg):
        self._lora_wgrad_group_now([(key, X, x_rows, x_seg, dY, dy_seg)])

    def _lora_wgrad_group(self, items):
        """_lora_wgrad for the image- and text-stream twins of one Linear group at once (items: (key, X, x_rows, x_seg, dY, dy_seg))."""
        if self._wgrad_stream is None or not self.overlap_wgrad:
            return self._lora_wgrad_group_now(items)
        ready = torch.cuda.Event()
        ready.record()
        self._wgrad_stream.wait_event(ready)
        for it in items:
            for t in (it[1], it[4]):
                t.record_stream(self._wgrad_stream)
        with torch.cuda.stream(self._wgrad_stream):
            self._lora_wgrad_group_now(items)

    def _lora_wgrad_group_now(self, items):
        """Adapter gradients of one Linear group, both streams: 2 skinny GEMM launches (t = X A^T, u = dY . blockdiag(B)) and ONE grouped
        token-contracted launch (csrc/gemm_tn.hip, grouped form) instead of 5 n + 1 launches per stream (round 4: 44 per block, now 6):
          dB_j += s dY_j^T t_j      and      dA_j += s u_j^T X            two problems per adapter, all in the one launch
        (the adapters of a fused q | k | v projection share X: their dA workgroups run side by side and X comes from HBM once)."""
        D = self.cfg.dim
        ts = ops.gemm_grouped([ops.gemm_desc(X, self._lora[key][0], a_seg=x_seg, M=M) for key, X, M, x_seg, dY, dy_seg in items])
        us = ops.gemm_grouped([ops.gemm_desc(dY, self._lora[key][3], a_seg=dy_seg, M=M) for key, X, M, x_seg, dY, dy_seg in items])
        descs = []
        for (key, X, M, x_seg, dY, dy_seg), t, u in zip(items, ts, us):
            for j, ad in enumerate(self._lora[key][2]):
                cols = slice(j * RPAD, (j + 1) * RPAD)
                descs.append(ops.tn_desc(dY[:, j * D:(j + 1) * D], t[:, cols], self.B_view(ad, self.grads), alpha=self.scale, M=M, p_seg=dy_seg))
                descs.append(ops.tn_desc(X, u[:, cols], self.A_view(ad, self.grads), alpha=self.scale, M=M, p_seg=x_seg, transpose_out=True))
        ops.gemm_tn_grouped(descs)                       # (2 problems per adapter: 12 for the q | k | v groups of both streams, 4 for the output projections)

    c_block_bwd = True         # the data-gradient chain of a block through advgrpo_mmdit_block_backward (one C-ABI call); False: launch by launch

    def _block_backward_c(self, i, b, s, dx, dc, dyg, dcyg, mods, B, Ni, Nt, ws):
        """Block i's data-gradient chain through the C entry, then its adapter-gradient launches (side stream) -> (dx, dc, dyg, dcyg) for block i - 1."""
        import ctypes
        lib, D, H = _lib.load(), self.cfg.dim, self.cfg.num_heads
        S, dev, bf16 = Ni + Nt, mods.device, torch.bfloat16
        p = lambda t: t.data_ptr() if t is not None else None
        new = lambda rows, cols: torch.empty(rows, cols, dtype=bf16, device=dev)
        first, last = i == 0, bool(b["last"])
        dx_out, dc_out = new(B * Ni, D), new(B * Nt, D)
        dyg_prev, dcyg_prev = (None, None) if first else (new(B * Ni, D), new(B * Nt, D))
        dyo, dyc, dqkv = new(B * Ni, D), (None if last else new(B * Nt, D)), new(B * S, 3 * D)
        att = s["att"]
        assert att.stride(2) == 1 and att.stride(0) == S * att.stride(1) and mods.stride(1) == 1
        d = _lib.MMDiTBlockBwdDesc()
        d.B, d.Ni, d.Nt, d.D, d.H, d.dual, d.last, d.first = B, Ni, Nt, D, H, int(b["dual"]), int(last), int(first)
        d.mods, d.mod_stride, d.mod_x, d.mod_c = mods.data_ptr(), mods.stride(0), self.mod_off[("x", i)], self.mod_off[("c", i)]
        if not first:
            d.mod_x_prev, d.mod_c_prev = self.mod_off[("x", i - 1)], self.mod_off[("c", i - 1)]
        d.ld_att = att.stride(1)
        for k in ("ff2", "ff1", "cff2", "cff1", "out", "cout", "qkv", "cqkv", "out2", "qkv2"):
            setattr(d, k + "_wT", p(b.get(k + ".wT")))
        d.rms_x, d.rms_c, d.rms_2 = p(b.get("rms_x")), p(b.get("rms_c")), p(b.get("rms_2"))
        for k in ("x_in", "c_in", "x_mid", "c_mid", "pre", "cpre", "qkv", "rs", "att", "lse", "qkv2", "rs2", "att2", "lse2"):
            t = s.get(k)
            assert t is None or t.is_contiguous() or k == "att"
            setattr(d, k, p(t))
        d.dx, d.dc, d.dyg, d.dcyg = p(dx), p(dc), p(dyg), p(dcyg)
        d.dx_out, d.dc_out, d.dyg_prev, d.dcyg_prev, d.dyo, d.dyc, d.dqkv = p(dx_out), p(dc_out), p(dyg_prev), p(dcyg_prev), p(dyo), p(dyc), p(dqkv)
        _lib.check(lib.advgrpo_mmdit_block_backward(ctypes.byref(d), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        att2d = att.reshape(B * S, D) if att.is_contiguous() else att.view(B * S, att.stride(1))[:, :D]
        self._lora_wgrad_group([((i, "out"), att2d, B * Ni, (Ni, S, 0), dyo, None)] +
                               ([] if last else [((i, "cout"), att2d, B * Nt, (Nt, S, Ni), dyc, None)]))
        self._lora_wgrad_group([((i, "qkv"), s["nx"], B * Ni, None, dqkv, (Ni, S, 0)),
                                ((i, "cqkv"), s["nc"], B * Nt, None, dqkv, (Nt, S, Ni))])
        return dx_out, dc_out, dyg_prev, dcyg_prev

    fuse_gates = True          # the norms' backward writes the gated copies of its result as well (backward; False: ops.gate_mul, for A/Bs)

    # ------------------------------------------------------------------ explicit backward
    @torch.no_grad()
    def backward(self, ctx, dv):
        """dv: gradient w.r.t. the model output [B,16,h,w] (bf16).  Accumulates LoRA gradients into self.grads."""
        cfg, w = self.cfg, self.w
        D, H = cfg.dim, cfg.num_heads
        B, Ni, Nt = ctx["B"], ctx["Ni"], ctx["Nt"]
        S = Ni + Nt
        mods = ctx["mods"]
        dev = dv.device
        bf16 = torch.bfloat16

        def mod(key, j):
            o = self.mod_off[key] + j * D
            return mods[:, o:o + D]
        side = self._wgrad_stream if self.overlap_wgrad else None
        if side is not None:                             # earlier work on the gradient vector (zeroing, previous micro-step)
            side.wait_stream(torch.cuda.current_stream())
        # final layer: v = unpatchify(LNmod(x) Wp^T + b)
        dtok = self._patch_rows_of_output_grad(dv)                              # [B*Ni, 64]
        dnx = ops.gemm(dtok, w["proj_out.wT"])                                  # [B*Ni, D]
        # Every gradient of a residual stream is consumed next by the gated projections hanging off that stream (dy = gate * dx): the
        # norm backward that produces dx writes those gated copies in the same pass (fuse_gates; the same bits as ops.gate_mul on the
        # stored dx, 5 launches and one read of dx per block less)
        fuse = self.fuse_gates
        L = cfg.num_layers

        def ln_bwd(x_saved, dy0, gates, rows, **kw):
            """-> (dx, gated copies in the order of `gates`)"""
            if fuse and gates:
                return ops.layernorm_mod_bwd(x_saved, dy0, rows_per_batch=rows, gates=gates, **kw)
            dxx = ops.layernorm_mod_bwd(x_saved, dy0, rows_per_batch=rows, **kw)
            return dxx, [ops.gate_mul(dxx, g, rows) for g in gates]
        dx, (dyg,) = ln_bwd(ctx["x_final"], dnx, [mod(("x", L - 1), 5)], Ni, scale0=mod(("out",), 0))
        dc = dcyg = None
        # one C-ABI call per block for the data-gradient chain (csrc/mmdit_block_bwd.cpp; the launches below, in the same order: bit-identical);
        # the side-path LoRA mode keeps its K-extended buffers out of that entry's dense-row contract and stays on the Python sequencing
        use_c = self.c_block_bwd and self.lora_mode == "merged" and fuse
        if use_c:
            lib = _lib.load()
            need = max(int(lib.advgrpo_mmdit_block_backward_workspace_bytes(B, Ni, Nt, D, H, int(d))) for d in {bool(b["dual"]) for b in self.blocks})
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
        for i in reversed(range(L)):
            b, s = self.blocks[i], ctx["blocks"][i]
            kx, kc = ("x", i), ("c", i)
            if use_c:
                dx, dc, dyg, dcyg = self._block_backward_c(i, b, s, dx, dc, dyg, dcyg, mods, B, Ni, Nt, ws)
                continue
            # ---- image MLP
            # (the text-stream data-gradient GEMMs ride in the launches of their image-stream twins, as in the forward)
            d2 = [ops.gemm_desc(dyg, b["ff2.wT"], act="dgelu_tanh", aux_in=s["pre"])]
            if not b["last"]:
                d2.append(ops.gemm_desc(dcyg, b["cff2.wT"], act="dgelu_tanh", aux_in=s["cpre"]))
            dpres = ops.gemm_grouped(d2)
            d1 = [ops.gemm_desc(dpres[0], b["ff1.wT"])]
            if not b["last"]:
                d1.append(ops.gemm_desc(dpres[1], b["cff1.wT"]))
            dmid = ops.gemm_grouped(d1)
            dx1, gx = ln_bwd(s["x_mid"], dmid[0], [mod(kx, 2)] + ([mod(kx, 8)] if b["dual"] else []), Ni, scale0=mod(kx, 4), dres=dx)
            dyo = gx[0]                                                         # grad of to_out.0 output
            # ---- text MLP
            if not b["last"]:
                dc1, (dyc,) = ln_bwd(s["c_mid"], dmid[1], [mod(kc, 2)], Nt, scale0=mod(kc, 4), dres=dc)
            # ---- second (image-only) attention of the dual blocks
            dnx2 = None
            if b["dual"]:
                dy2 = gx[1]
                datt2 = ops.gemm(dy2, b["out2.wT"]).view(B, Ni, D)
                q3 = s["qkv2"].view(B, Ni, 3 * D)
                dqkv2 = torch.empty(B * Ni, 3 * D, dtype=bf16, device=dev)
                d3 = dqkv2.view(B, Ni, 3 * D)
                ops.attention_bwd(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], s["att2"], datt2, s["lse2"], H,
                                  d3[:, :, :D], d3[:, :, D:2 * D], d3[:, :, 2 * D:])
                ops.rmsnorm_heads_bwd(dqkv2, s["qkv2"], s["rs2"], 0, 2 * H, b["rms_2"], H)
                dnx2 = ops.gemm(dqkv2, b["qkv2.wT"])
            # ---- joint attention
            datt = torch.zeros(B * S, D, dtype=bf16, device=dev) if b["last"] else torch.empty(B * S, D, dtype=bf16, device=dev)
            douts = [ops.gemm_desc(dyo, b["out.wT"], out=datt, seg=(Ni, S, 0))]
            att2d = s["att"].view(B * S, D)
            if not b["last"]:
                douts.append(ops.gemm_desc(dyc, b["cout.wT"], out=datt, seg=(Nt, S, Ni)))
            ops.gemm_grouped(douts)
            self._lora_wgrad_group([((i, "out"), att2d, B * Ni, (Ni, S, 0), dyo, None)] +
                                   ([] if b["last"] else [((i, "cout"), att2d, B * Nt, (Nt, S, Ni), dyc, None)]))
            q3 = s["qkv"].view(B, S, 3 * D)
            dqkv = torch.empty(B * S, 3 * D, dtype=bf16, device=dev)
            d3 = dqkv.view(B, S, 3 * D)
            ops.attention_bwd(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], s["att"], datt.view(B, S, D), s["lse"], H,
                              d3[:, :, :D], d3[:, :, D:2 * D], d3[:, :, 2 * D:])
            ops.rmsnorm_heads_bwd(dqkv, s["qkv"], s["rs"], 0, 2 * H, b["rms_x"], H, seg=(Ni, S, 0), M=B * Ni)
            ops.rmsnorm_heads_bwd(dqkv, s["qkv"], s["rs"], 0, 2 * H, b["rms_c"], H, seg=(Nt, S, Ni), M=B * Nt)
            dnx, dnc = ops.gemm_grouped([ops.gemm_desc(dqkv, b["qkv.wT"], a_seg=(Ni, S, 0), M=B * Ni),
                                         ops.gemm_desc(dqkv, b["cqkv.wT"], a_seg=(Nt, S, Ni), M=B * Nt)])
            self._lora_wgrad_group([((i, "qkv"), s["nx"], B * Ni, None, dqkv, (Ni, S, 0)),
                                    ((i, "cqkv"), s["nc"], B * Nt, None, dqkv, (Nt, S, Ni))])
            # ---- first norms
            # (the gated copies are for block i - 1's feed-forward output projections)
            gxn = [mod(("x", i - 1), 5)] if i else []
            gcn = [mod(("c", i - 1), 5)] if i else []
            dx, gx = ln_bwd(s["x_in"], dnx, gxn, Ni, scale0=mod(kx, 1), dy1=dnx2, scale1=mod(kx, 7) if b["dual"] else None, dres=dx1)
            if b["last"]:
                dc, gc = ln_bwd(s["c_in"], dnc, gcn, Nt, scale0=mod(kc, 0))
            else:
                dc, gc = ln_bwd(s["c_in"], dnc, gcn, Nt, scale0=mod(kc, 1), dres=dc1)
            dyg, dcyg = (gx[0], gc[0]) if i else (None, None)
        if side is not None:                             # the gradient vector is complete for whoever reads it next
            torch.cuda.current_stream().wait_stream(side)
        return dx, dc

    def _patch_rows_of_output_grad(self, dv):
        """inverse of unpatchify: [B,C,H,W] -> token rows [B*(H/2)*(W/2), 4*C] with column (py*2+px)*C + c."""
        B, C, Hh, Ww = dv.shape
        x = dv.view(B, C, Hh // 2, 2, Ww // 2, 2).permute(0, 2, 4, 3, 5, 1)     # index plumbing (one strided copy)
        return x.reshape(B * (Hh // 2) * (Ww // 2), 4 * C).contiguous()

    # ------------------------------------------------------------------ optimiser
    @torch.no_grad()
    def optimizer_step(self, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, max_grad_norm=1.0, grad_scale=1.0):
        """clip_grad_norm_(max_grad_norm) + AdamW.step() + zero_grad() (TP:1166-1171), then re-merge."""
        lib = _lib.load()
        self.opt_step += 1
        sumsq = torch.zeros(1, dtype=torch.float32, device=self.device)
        ws = torch.empty(lib.advgrpo_sumsq_workspace_bytes() // 4, dtype=torch.float32, device=self.device)
        _lib.check(lib.advgrpo_sumsq_f32(self.grads.data_ptr(), self.n_params, sumsq.data_ptr(), ws.data_ptr(), _lib.stream_ptr()))
        _lib.check(lib.advgrpo_adamw_step(self.params.data_ptr(), self.params_bf16.data_ptr(), self.grads.data_ptr(),
                                          self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.n_params, lr, betas[0],
                                          betas[1], eps, weight_decay, self.opt_step, sumsq.data_ptr(), max_grad_norm,
                                          grad_scale, _lib.stream_ptr()))
        self.refresh()
        return sumsq

    @torch.no_grad()
    def ema_step(self, optimization_step, decay=0.9, update_step_interval=8):
        """EMAModuleWrapper.step (adv_grpo/ema.py:39-52)."""
        self.ema_wrapper.decay, self.ema_wrapper.update_step_interval = decay, update_step_interval
        self.ema_wrapper.step([self.params], optimization_step)
